#!/bin/bash
# Collects the round's profile evidence on the GPU box (run through gpurun from the repo root):
#   rocprofv3 --kernel-trace --stats of the default bench command and of a one-stream run (isolated kernel durations),
#   SQ counter passes and the HBM traffic passes (FETCH_SIZE / WRITE_SIZE, separate passes) of tools/traffic_driver.py.
# Outputs land in gpurun_out/prof_*; tools/collect_*.py turn them into the JSON files kept under profiles/.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
R=${1:-r03}
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_default -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-latency --no-check > gpurun_out/prof_default.json 2> gpurun_out/prof_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_1stream -- python3 bench.py --steps 5 --warmup 2 --streams 1 --batch 2048 --no-cpu-baseline --no-latency --no-check > gpurun_out/prof_1stream.json 2> gpurun_out/prof_1stream.err
export TRAFFIC_B=1024
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d gpurun_out/prof_sq1 -- python3 tools/traffic_driver.py > gpurun_out/prof_sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM --kernel-trace --output-format csv -d gpurun_out/prof_sq2 -- python3 tools/traffic_driver.py > gpurun_out/prof_sq2.log 2>&1
python3 tools/collect_sq.py 1024 gpurun_out/${R}_sq_counters.json gpurun_out/prof_sq1 gpurun_out/prof_sq2 > /dev/null
# HBM traffic at the benchmark's own launch size and geometry (8192 frames over 5 contexts: 1638 frames per launch; traffic_driver's context has a gate like bench.py's)
export TRAFFIC_B=${TRAFFIC_LAUNCH:-1638}
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_fetch -- python3 tools/traffic_driver.py > gpurun_out/prof_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_write -- python3 tools/traffic_driver.py > gpurun_out/prof_write.log 2>&1
python3 tools/collect_traffic.py gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/${R}_traffic.json > gpurun_out/${R}_traffic.log
find gpurun_out/prof_default gpurun_out/prof_1stream -name "*kernel_stats.csv" | head
echo profile_round done
