for mb in 512 1024; do
  OCVAR_MID_BLOCKS=$mb timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/mb_$mb.json 2>gpurun_out/mb_$mb.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/mb_$mb.json").read().strip().splitlines()[-1])
print($mb, d["value"], {k: round(v,2) for k,v in d["isolated_launch_ms"].items()})
PY
done
