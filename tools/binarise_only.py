#!/usr/bin/env python3
"""Times binarise_frames_kernel alone (OCVAR_ONLY_BINARISE=1: the batch stops after the first kernel) for every library
given on the command line -- experiment builds whose outputs may be wrong.  Prints ms per 2048-frame launch."""
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] != "--child":
    for lib in sys.argv[1:]:
        env = dict(os.environ, OCVAR_ONLY_BINARISE="1", OCVAR_HIP_LIB=os.path.join(ROOT, "opencv-ar_amd", "lib", lib))
        subprocess.run([sys.executable, __file__, "--child", lib], env=env)
    sys.exit(0)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import helpers as H
import opencv_ar_amd as oa
B, uniq = int(os.environ.get("PROF_B", "2048")), 64
cfg = H.synth_config(int(os.environ.get("PROF_CONFIG", "3")))
base = np.stack([H.synth_frame(cfg, i, None)[0] for i in range(uniq)])
W, Hh = cfg.width, cfg.height
det = oa.Detector(W, Hh, max_batch=B)
det.set_templates(oa.load_templates([os.path.join(oa.TEMPLATE_DIR, n + ".png") for n in H.TEMPLATE_ORDER]))
det.set_camera(oa.default_camera(W, Hh))
d = torch.from_numpy(base).cuda().repeat(B // uniq, 1, 1, 1).contiguous()
torch.cuda.synchronize()
ms = []
for _ in range(6):
    det.enqueue_device(d.data_ptr(), W, Hh, B)
    try:
        det.collect(8)
    except Exception as e:   # experiment builds may trip capacity flags
        print("collect:", e)
    ms.append(float(det.stage_ms()[0]))
print("%-34s binarise_frames ms %s  -> %.3f" % (sys.argv[2], [round(m, 3) for m in ms], min(ms[1:])))
