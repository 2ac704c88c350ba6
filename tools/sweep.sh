#!/bin/bash
# usage: tools/sweep.sh TAG "ENV1=a ENV2=b -- bench args" ...   (one bench.py run per argument; prints value + isolated stage ms)
tag=$1; shift
i=0
for spec in "$@"; do
  envs="${spec%%--*}"; args="${spec#*--}"
  out=gpurun_out/${tag}_$i
  env $envs timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-latency $args > $out.json 2> $out.err || { echo "FAIL [$spec]"; tail -3 $out.err; i=$((i+1)); continue; }
  python - <<PY
import json
d = json.loads(open("$out.json").read().strip().splitlines()[-1])
print("[$spec]", d["value"], "ms/step", d["ms_per_step"], "iso", {k: round(v,2) for k,v in d["isolated_launch_ms"].items()})
PY
  i=$((i+1))
done
