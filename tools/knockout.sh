#!/bin/bash
# What each group of kernels costs with four contexts sharing the GPU: the default bench command on the instrumented build
# (make prof) with kernels knocked out (results invalid: timing experiment).  Run through gpurun.
export OCVAR_HIP_LIB=$PWD/opencv-ar_amd/lib/libocvar_hip_prof.so
run() { timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-latency --no-check > gpurun_out/ko.json 2>/dev/null; echo "$1: $(python tools/show_bench.py gpurun_out/ko.json | head -2 | tr '\n' ' ' | cut -c1-230)"; }
run "everything"
OCVAR_SKIP_CROP_KERNELS=4 run "no crop tier 2"
OCVAR_SKIP_CROP_KERNELS=2 run "no crop tier 1(+what follows)"
OCVAR_SKIP_CROP_KERNELS=8 run "no crop tier 3"
OCVAR_SKIP_CROP_KERNELS=14 run "no crop followers"
OCVAR_SKIP_CROP_KERNELS=15 run "no crop pass"
OCVAR_SKIP_CROP_KERNELS=1 run "no crop binarise"
OCVAR_ONLY_BINARISE=1 run "frame binarise only"
run "everything"
