#!/bin/bash
# A/B of two builds of libocvar_hip.so on one box: alternate runs
for rep in 1 2; do
for a in "--batch 64" "--batch 256" "--batch 1024" ""; do
  for lib in old new; do
    if [ $lib = old ]; then export OCVAR_HIP_LIB=$PWD/opencv-ar_amd/lib/libocvar_hip_old.so; else unset OCVAR_HIP_LIB; fi
    timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency --no-check $a > gpurun_out/ab.json 2>/dev/null || exit 1
    echo "$lib $a: $(python tools/show_bench.py gpurun_out/ab.json | head -1)"
  done
done
done
