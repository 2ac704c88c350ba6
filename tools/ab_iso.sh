#!/bin/bash
# A/B of builds of libocvar_hip.so on one box: isolated kernel times (one context, 2048 frames per launch) and the default bench
# usage: ab_iso.sh name1 name2 ...   (opencv-ar_amd/lib/libocvar_hip_<name>.so; "new" = the product library)
for rep in 1 2; do
for lib in "$@"; do
  if [ $lib = new ]; then unset OCVAR_HIP_LIB; else export OCVAR_HIP_LIB=$PWD/opencv-ar_amd/lib/libocvar_hip_$lib.so; fi
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency --no-check > gpurun_out/ab_$lib.json 2>/dev/null || exit 1
  python - gpurun_out/ab_$lib.json $lib <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
i=d["isolated_launch_ms"]; s=d["stage_ms"]
print("%-8s %7.0f f/s  iso binF %.2f binC %.2f f2C %.2f total %.2f | in-region binF %.2f binC %.2f" % (sys.argv[2], d["value"], i["binarise_frames"], i["binarise_crops"], i["follow2_crops"], i["batch_total"], s["binarise_frames"], s["binarise_crops"]))
PY
done
done
