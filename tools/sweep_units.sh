for mu in 16384 32768 65536 1000000000; do
  OCVAR_MIN_UNITS=$mu timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/mu_$mu.json 2>gpurun_out/mu_$mu.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/mu_$mu.json").read().strip().splitlines()[-1])
print($mu, d["value"], d["isolated_launch_ms"]["binarise_frames"], {k: round(v,2) for k,v in d["stage_ms"].items()})
PY
done
