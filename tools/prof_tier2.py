#!/usr/bin/env python3
"""Where tier 2 (follow_mid_kernel) spends its time: runs one context on the benchmark frames with a library built with
-DOCVAR_PROF (make -C opencv-ar_amd prof; OCVAR_HIP_LIB points at it) and prints the profiling slots per 2048-frame launch:
cycles (s_memtime, summed over waves) in start hand-out / stepping / point flush / finishing, wave-steps, lane-steps,
borders finished by the wave, borders that reached the polygon approximation."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("OCVAR_HIP_LIB", os.path.join(ROOT, "opencv-ar_amd", "lib", "libocvar_hip_prof.so"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import helpers as H
import opencv_ar_amd as oa

B = int(os.environ.get("PROF_B", "2048"))
uniq = 64
cfg = H.synth_config(3)
base = np.stack([H.synth_frame(cfg, i, None)[0] for i in range(uniq)])
W, Hh = cfg.width, cfg.height
det = oa.Detector(W, Hh, max_batch=B)
det.set_templates(oa.load_templates([os.path.join(oa.TEMPLATE_DIR, n + ".png") for n in H.TEMPLATE_ORDER]))
det.set_camera(oa.default_camera(W, Hh))
gate = oa.Gate(2, 0) if os.environ.get("PROF_GATE") == "1" else None   # PROF_GATE=1: the launch geometry of a context that shares the GPU (a quarter of tier 2's grid)
if gate is not None:
    det.set_gate(gate)
d = torch.from_numpy(base).cuda().repeat(B // uniq, 1, 1, 1).contiguous()
torch.cuda.synchronize()
for _ in range(2):
    det.detect_device(d.data_ptr(), W, Hh, B)
c = det.counters(42)
ms = det.stage_ms()
print("stage ms", dict(zip(oa.STAGE_NAMES, ms.round(3).tolist())))
print("counters", c[:10].tolist())
names = ["cyc_handout", "cyc_step", "cyc_flush", "cyc_finish", "wave_steps", "lane_steps", "finished", "reached_dp",
         "steps_not_first", "n_not_first", "steps_closed", "n_closed", "steps_overrun", "n_overrun", "steps_closed_after_best", "n_closed_lt4pts"]
for label, off in (("frames", 10), ("crops", 26)):
    v = c[off:off + 16].astype(float)
    tot = v[:4].sum()
    print(label, {n: int(x) for n, x in zip(names, v)})
    if tot > 0:
        print("   cycle split  handout %.1f%%  step %.1f%%  flush %.1f%%  finish %.1f%%;  active lanes per wave-step %.1f;  cycles per wave-step %.0f;  cycles per finish %.0f" % (
            100 * v[0] / tot, 100 * v[1] / tot, 100 * v[2] / tot, 100 * v[3] / tot, v[5] / max(v[4], 1), v[1] / max(v[4], 1), v[3] / max(v[6], 1)))
