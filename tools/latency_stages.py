#!/usr/bin/env python3
"""Per-kernel times of ONE 1080p frame per call (the reference's calling pattern, samples/ARTest.cpp:57): median of 20 calls of
ocvar_hip_detect_host on a one-frame context, HIP-event stage times + wall time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import helpers as H
import opencv_ar_amd as oa
cid = int(sys.argv[1]) if len(sys.argv) > 1 else 3
cfg = H.synth_config(cid)
names = ["2x2-01"] if cid in (1, 2) else None
frame = H.synth_frame(cfg, 0, names)[0][None].copy()
det = oa.Detector(cfg.width, cfg.height, max_batch=1)
det.set_templates(oa.load_templates([os.path.join(oa.TEMPLATE_DIR, n + ".png") for n in (names or H.TEMPLATE_ORDER)]))
det.set_camera(oa.default_camera(cfg.width, cfg.height))
st, wall = [], []
for i in range(25):
    work = frame.copy()
    t0 = time.perf_counter()
    det.detect_host(work, grey_in_place=True)
    wall.append(1e3 * (time.perf_counter() - t0))
    st.append(det.stage_ms())
st = np.median(np.array(st[5:]), axis=0)
print("work counters of the last call (starts, crops, crop units, crop starts, crop pixels, pool ints, tier-2 starts F/C, tier-3 borders F/C):", det.counters().tolist())
print(f"{cfg.width}x{cfg.height}: wall median {np.median(wall[5:]):.3f} ms; stages", {k: round(float(v), 3) for k, v in zip(oa.STAGE_NAMES, st)})
