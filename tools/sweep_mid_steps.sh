# throughput of the default bench vs. the tier-2 step budget (what exceeds it goes to the wave tier)
for ms in 1536 768 384 192 96; do
  OCVAR_MID_STEPS=$ms timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/sweep_$ms.json 2>gpurun_out/sweep_$ms.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/sweep_$ms.json").read().strip().splitlines()[-1])
print($ms, d["value"], {k: round(v,2) for k,v in d["isolated_launch_ms"].items()})
PY
done
