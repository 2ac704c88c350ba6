#!/bin/bash
# bench.py under a list of argument strings, one line per setting.  Run through gpurun.
# usage: tools/sweep_args.sh name "--gate 3" "--streams 6 --gate 2" ...
name=$1; shift
i=0
for a in "$@"; do
  timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-latency --no-check $a > gpurun_out/${name}_$i.json 2> gpurun_out/${name}_$i.err || exit 1
  echo "$a: $(python tools/show_bench.py gpurun_out/${name}_$i.json | head -1)"
  i=$((i+1))
done
