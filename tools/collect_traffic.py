#!/usr/bin/env python3
"""Parses two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE -- they do not fit one pass on gfx950) of
tools/traffic_driver.py into profiles/r03_traffic.json (or the path given as third argument): HBM bytes per frame of the streaming kernels.

Counter units are KiB (bytes = value * 1024).  On gfx950 the counters are only calibrated for 16-byte-per-lane streams
(MI355X_MICROARCH.md, HBM section), so both are calibrated on `calib_copy_dword_kernel`, a copy of a known byte count
with this path's access width (one dword per lane); the correction factors are written to the JSON.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 tools/traffic_driver.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 tools/traffic_driver.py
    TRAFFIC_B=2048 python3 tools/collect_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write [out.json]   (TRAFFIC_B as in the passes)
"""
import csv, glob, json, os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B, CAL, W, H = int(os.environ.get("TRAFFIC_B", "64")), 1 << 28, 1920, 1080


def mean_counter(d, counter):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def pick(d, name):
    for k, v in d.items():
        if name in k:
            return v
    raise KeyError(name)


fetch, write = mean_counter(sys.argv[1], "FETCH_SIZE"), mean_counter(sys.argv[2], "WRITE_SIZE")
cf = CAL / (pick(fetch, "calib_copy_dword_kernel") * 1024)
cw = CAL / (pick(write, "calib_copy_dword_kernel") * 1024)
out = {"_calibration": {"bytes": CAL, "fetch_factor": cf, "write_factor": cw,
                        "note": "bytes = counter * 1024 * factor; factors from calib_copy_dword_kernel (dword per lane)"}}
for name in ("binarise_frames_kernel", "binarise_crops_kernel"):
    fb = pick(fetch, name) * 1024 * cf / B
    wb = pick(write, name) * 1024 * cw / B
    out[name] = {"width": W, "height": H, "fetch_bytes_per_frame": fb, "write_bytes_per_frame": wb,
                 "hbm_bytes_per_frame": fb + wb, "raw_fetch_kib_per_launch": pick(fetch, name),
                 "raw_write_kib_per_launch": pick(write, name), "frames_per_launch": B}
dest = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "profiles", "r03_traffic.json")
json.dump(out, open(dest, "w"), indent=1)
print(json.dumps(out, indent=1))
