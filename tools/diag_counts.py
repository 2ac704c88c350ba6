"""Work counters of one 1080p batch (diagnostic)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import helpers as H
import opencv_ar_amd as oa
B = int(os.environ.get("DIAG_B", "64"))
cfg = H.synth_config(3)
frames = np.stack([H.synth_frame(cfg, i)[0] for i in range(B)])
tpls = oa.load_templates([os.path.join(oa.TEMPLATE_DIR, n + ".png") for n in H.TEMPLATE_ORDER])
cam = oa.default_camera(cfg.width, cfg.height)
det = oa.Detector(cfg.width, cfg.height, max_batch=B)
det.set_templates(tpls); det.set_camera(cam)
d = torch.from_numpy(frames).cuda()
for _ in range(2):
    det.detect_device(d.data_ptr(), cfg.width, cfg.height, B)
c = det.counters()
names = ["frame_cands", "crop_rois", "crop_tiles", "crop_cands", "crop_pixels", "pool_ints", "mid_f", "mid_c", "long_f", "long_c"]
print({n: int(v) / B for n, v in zip(names, c)})
print("stage ms", dict(zip(oa.STAGE_NAMES, det.stage_ms().round(3).tolist())))
