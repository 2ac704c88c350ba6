#!/usr/bin/env python3
"""Parses rocprofv3 PMC passes of tools/traffic_driver.py (counter_collection.csv + kernel_trace.csv) into one JSON:
per kernel the mean counter values per launch (sums over the device) and the mean launch duration under the profiler.

    rocprofv3 --pmc <counters> --kernel-trace --output-format csv -d gpurun_out/pmc_X -- python3 tools/traffic_driver.py
    python3 tools/collect_sq.py <frames_per_launch> <out.json> gpurun_out/pmc_X [gpurun_out/pmc_Y ...]
"""
import csv, glob, json, os, sys, collections

frames, out_path, dirs = int(sys.argv[1]), sys.argv[2], sys.argv[3:]
res = collections.defaultdict(dict)
for d in dirs:
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if "ocvar::" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        for c, v in cs.items():
            res[k][c] = round(sum(v) / len(v))
    kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if kt:
        dur = collections.defaultdict(list)
        for r in csv.DictReader(open(kt[0])):
            if "ocvar::" in r["Kernel_Name"]:
                dur[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        for k, v in dur.items():
            res[k].setdefault("duration_us_profiled", round(sum(v) / len(v), 1))
for k, v in res.items():
    if "SQ_INSTS_VALU" in v:
        v["valu_wave_instructions_per_frame"] = round(v["SQ_INSTS_VALU"] / frames)
out = {"_note": "rocprofv3 --pmc ... --kernel-trace -- python3 tools/traffic_driver.py with TRAFFIC_B=%d (one context, %d frames 1920x1080 per "
                "launch, mean over the launches); counters are sums over the device; SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* in "
                "quad-cycles; durations under the profiler" % (frames, frames), "frames_per_launch": frames}
out.update(res)
json.dump(out, open(out_path, "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if "follow_mid" in k or "binarise" in k}, indent=1))
