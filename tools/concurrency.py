#!/usr/bin/env python3
"""How the contexts of bench.py share the GPU: from a rocprofv3 --kernel-trace CSV of the default command, the share of
the steady state (middle of the run) during which k binarise kernels / k kernels of any kind were running at once.

    python3 tools/concurrency.py gpurun_out/prof_default/runc/<pid>_kernel_trace.csv
"""
import collections
import csv
import sys


def kind(n):
    if "binarise_frames" in n or "binarise_crops" in n:
        return "bin"
    if "follow_" in n:
        return "fol"
    return "tail"


rows = [r for r in csv.DictReader(open(sys.argv[1])) if "ocvar::" in r["Kernel_Name"]]
t0 = min(int(r["Start_Timestamp"]) for r in rows)
t1 = max(int(r["End_Timestamp"]) for r in rows)
a, b = t0 + (t1 - t0) * 0.25, t0 + (t1 - t0) * 0.85
ev = []
for r in rows:
    k = kind(r["Kernel_Name"])
    ev.append((int(r["Start_Timestamp"]), 1, k))
    ev.append((int(r["End_Timestamp"]), -1, k))
ev.sort()
act, last, hist, tot = collections.Counter(), None, collections.Counter(), 0
for t, d, k in ev:
    if last is not None and t > last:
        lo, hi = max(last, a), min(t, b)
        if hi > lo:
            hist[(act["bin"], act["fol"], act["tail"])] += hi - lo
            tot += hi - lo
    act[k] += d
    last = t
nb, nk = collections.Counter(), collections.Counter()
for (bb, ff, tt), v in hist.items():
    nb[bb] += v
    nk[bb + ff + tt] += v
print("binarise kernels running at once:", {k: "%.1f %%" % (100 * v / tot) for k, v in sorted(nb.items())})
print("kernels of any kind running at once:", {k: "%.1f %%" % (100 * v / tot) for k, v in sorted(nk.items())})
print("most frequent (binarise, followers, tail) states:", [(k, "%.1f %%" % (100 * v / tot)) for k, v in sorted(hist.items(), key=lambda kv: -kv[1])[:6]])

# the figure bench.py reports as roofline.launch_ms: time at least one binarise_frames_kernel was running, per launch
for name in ("binarise_frames_kernel", "binarise_crops_kernel"):
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if name in r["Kernel_Name"] and a <= int(r["Start_Timestamp"]) <= b)
    # full-size launches only (the pipeline's first and last launches are partial): within 25 % of the median duration... by grid size if present
    if not iv:
        continue
    tot, ca, cb = 0, None, None
    for x, y in iv:
        if cb is None or x > cb:
            if cb is not None:
                tot += cb - ca
            ca, cb = x, y
        else:
            cb = max(cb, y)
    tot += cb - ca
    mean = sum(y - x for x, y in iv) / len(iv)
    print(f"{name}: {len(iv)} launches in the window, mean start-to-end {mean / 1e6:.3f} ms, time at least one was running per launch {tot / len(iv) / 1e6:.3f} ms")
