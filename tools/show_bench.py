#!/usr/bin/env python3
"""Prints the headline and the per-kernel times of bench.py JSON lines (files given on the command line)."""
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "ERR", e); continue
    print(f"{f}: {d['value']:.0f} frames/s, {d['ms_per_step']:.2f} ms/step")
    for k in ("stage_ms", "isolated_launch_ms"):
        if k in d:
            print("  %-18s" % k + " ".join(f"{n.replace('follow','f').replace('binarise','bin').replace('_frames','F').replace('_crops','C')}={v:.2f}" for n, v in d[k].items()))
    if "latency_ms" in d: print("  latency_ms", d["latency_ms"])
    if "cpu_baseline" in d: print("  cpu", d["cpu_baseline"])
