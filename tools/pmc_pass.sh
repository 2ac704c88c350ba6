#!/bin/bash
# One rocprofv3 counter pass over tools/traffic_driver.py (TRAFFIC_B frames per launch) -> gpurun_out/pmc_<tag>.json
# usage: pmc_pass.sh <tag> COUNTER [COUNTER ...]
set -e
tag=$1; shift
R=$PWD
cd /tmp && export TMPDIR=/tmp && cd "$R"
rm -rf gpurun_out/pmc_$tag
rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/pmc_$tag -- python3 tools/traffic_driver.py > gpurun_out/pmc_$tag.log 2>&1
python3 tools/collect_sq.py ${TRAFFIC_B:-64} gpurun_out/pmc_$tag.json gpurun_out/pmc_$tag > /dev/null
python3 - gpurun_out/pmc_$tag.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for k,v in d.items():
    if isinstance(v,dict) and ("binarise" in k):
        print(k, {a:b for a,b in v.items()})
PY
