#!/bin/bash
# bench.py under a list of argument strings, one line per setting.  Run through gpurun.
# usage: tools/sweep_args.sh name "--streams 5" "--streams 6 --gate 3" ...
name=$1; shift
i=0
for t in "$@"; do
  timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-latency --no-check $t > gpurun_out/${name}_$i.json 2> gpurun_out/${name}_$i.err || { echo "$t: failed"; tail -2 gpurun_out/${name}_$i.err; }
  echo "$t: $(python tools/show_bench.py gpurun_out/${name}_$i.json | head -2 | tr '\n' ' ' | cut -c1-260)"
  i=$((i+1))
done
