# throughput vs. (batch, streams, GPU_MAX_HW_QUEUES); prints value and ms/step
for cfg in "4096 4 4" "8192 8 8" "8192 4 4" "6144 6 8" "4096 8 8"; do
  set -- $cfg
  GPU_MAX_HW_QUEUES=$3 timeout -k 10 200 python bench.py --batch $1 --streams $2 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/ss_$1_$2.json 2>gpurun_out/ss_$1_$2.err || { echo "fail $cfg"; tail -3 gpurun_out/ss_$1_$2.err; continue; }
  python - <<PY
import json
d = json.loads(open("gpurun_out/ss_$1_$2.json").read().strip().splitlines()[-1])
print("$1 $2 q$3", d["value"], d["ms_per_step"], d["roofline"]["launch_ms"])
PY
done
