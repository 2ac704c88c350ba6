#!/bin/bash
# A/B of builds of libocvar_hip.so on one box with the default bench command, alternating, REPS rounds (default 4)
# usage: ab_bench.sh name1 name2 ...   (opencv-ar_amd/lib/libocvar_hip_<name>.so; "new" = the product library)
for rep in $(seq 1 ${REPS:-4}); do
for lib in "$@"; do
  if [ $lib = new ]; then unset OCVAR_HIP_LIB; else export OCVAR_HIP_LIB=$PWD/opencv-ar_amd/lib/libocvar_hip_$lib.so; fi
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency --no-check > gpurun_out/ab_$lib.json 2>/dev/null || exit 1
  python -c "
import json,sys
d=json.loads(open('gpurun_out/ab_$lib.json').read().strip().splitlines()[-1]); s=d['stage_ms']
print('%-8s %7.0f f/s | in-region binF %.2f binC %.2f f2C %.2f chain %.2f' % ('$lib', d['value'], s['binarise_frames'], s['binarise_crops'], s['follow2_crops'], s['batch_total']))"
done
done
