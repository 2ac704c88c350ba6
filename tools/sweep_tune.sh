#!/bin/bash
# bench.py under a list of --tune settings (result-invariant launch parameters), one line per setting.  Run through gpurun.
# usage: tools/sweep_tune.sh name "mid_blocks=256" "mid_blocks=512 long_blocks=256" ...
name=$1; shift
i=0
for t in "$@"; do
  args=""
  for kv in $t; do args="$args --tune $kv"; done
  timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-latency --no-check $args > gpurun_out/${name}_$i.json 2> gpurun_out/${name}_$i.err || exit 1
  echo "$t: $(python tools/show_bench.py gpurun_out/${name}_$i.json | head -1)"
  i=$((i+1))
done
