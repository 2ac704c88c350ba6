#!/bin/bash
# The round's kept bench lines (run through gpurun from the repo root; copy gpurun_out/r03_* into profiles/ afterwards).
R=${1:-r03}
python bench.py > gpurun_out/${R}_bench_default.json 2> gpurun_out/${R}_bench_default.err || exit 1
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/${R}_bench_driverlike.json 2>/dev/null || exit 1
for b in 64 256 1024; do python bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline --no-latency > gpurun_out/${R}_bench_batch$b.json 2>/dev/null || exit 1; done
python bench.py --config 2 --batch 32768 --no-cpu-baseline --no-latency > gpurun_out/${R}_bench_config2_640x480.json 2>/dev/null || exit 1
python bench.py --config 5 --batch 2048 --no-cpu-baseline --no-latency > gpurun_out/${R}_bench_config5_4k.json 2>/dev/null || exit 1
python tools/show_bench.py gpurun_out/${R}_bench_*.json | grep "frames/s"
(python tools/pipe_rate.py 16384 2048 4 2; python tools/pipe_rate.py 16384 1638 5 2) > gpurun_out/${R}_pipe_rate.txt 2>&1; grep -E "best|streaming" gpurun_out/${R}_pipe_rate.txt | tail -6
python tools/host_transport_rate.py > gpurun_out/${R}_host_transport.txt 2>&1; tail -4 gpurun_out/${R}_host_transport.txt
