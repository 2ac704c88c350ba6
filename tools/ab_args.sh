#!/bin/bash
# builds x argument strings: ab_args.sh "lib1 lib2" "args1" "args2" ...   ("new" = the product library)
libs=$1; shift
for a in "$@"; do
for lib in $libs; do
  if [ $lib = new ]; then unset OCVAR_HIP_LIB; else export OCVAR_HIP_LIB=$PWD/opencv-ar_amd/lib/libocvar_hip_$lib.so; fi
  timeout -k 10 200 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-latency --no-check $a > gpurun_out/aa_$lib.json 2>/dev/null || { echo "$lib $a: failed"; continue; }
  python -c "
import json
d=json.loads(open('gpurun_out/aa_$lib.json').read().strip().splitlines()[-1]); s=d['stage_ms']
print('%-6s %-28s %7.0f f/s | binF %.2f binC %.2f f2F %.2f f2C %.2f chain %.2f' % ('$lib', '$a', d['value'], s['binarise_frames'], s['binarise_crops'], s['follow2_frames'], s['follow2_crops'], s['batch_total']))"
done
done
