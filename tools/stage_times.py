#!/usr/bin/env python3
"""Stand-alone kernel durations (one context, nothing else on the GPU) for the libraries given on the command line
(names under opencv-ar_amd/lib, e.g. libocvar_hip.so libocvar_hip_old.so), alternating between them: median and minimum of
the HIP-event stage times over REPS launches of PROF_B frames.  For A/B comparisons of kernel changes on one box."""
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] != "--child":
    for rep in range(int(os.environ.get("ROUNDS", "2"))):
        for lib in sys.argv[1:]:
            env = dict(os.environ, OCVAR_HIP_LIB=os.path.join(ROOT, "opencv-ar_amd", "lib", lib))
            subprocess.run([sys.executable, __file__, "--child", lib], env=env)
    sys.exit(0)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import helpers as H
import opencv_ar_amd as oa
B, uniq = int(os.environ.get("PROF_B", "2048")), 32
cfg = H.synth_config(int(os.environ.get("PROF_CONFIG", "3")))
base = np.stack([H.synth_frame(cfg, i, None)[0] for i in range(uniq)])
W, Hh = cfg.width, cfg.height
det = oa.Detector(W, Hh, max_batch=B)
det.set_templates(oa.load_templates([os.path.join(oa.TEMPLATE_DIR, n + ".png") for n in H.TEMPLATE_ORDER]))
det.set_camera(oa.default_camera(W, Hh))
d = torch.from_numpy(base).cuda().repeat(B // uniq, 1, 1, 1).contiguous()
torch.cuda.synchronize()
rows = []
for _ in range(int(os.environ.get("REPS", "10"))):
    det.enqueue_device(d.data_ptr(), W, Hh, B)
    det.collect(8)
    rows.append(np.array(det.stage_ms(), float))
a = np.array(rows[2:])
med, mn = np.median(a, 0), a.min(0)
names = [n.replace("follow", "f").replace("binarise", "bin").replace("_frames", "F").replace("_crops", "C") for n in oa.STAGE_NAMES]
print("%-26s " % sys.argv[2] + " ".join("%s=%.2f/%.2f" % (n, m, x) for n, m, x in zip(names, med, mn) if m >= 0.2) + "  sum=%.2f" % med.sum())
