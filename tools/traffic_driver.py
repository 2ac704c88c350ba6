"""Driver profiled by tools/collect_traffic.py: one calibration copy + a few 1080p batches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import helpers as H
import opencv_ar_amd as oa
B, CAL = int(os.environ.get("TRAFFIC_B", "64")), 1 << 28
cfg = H.synth_config(3)
base = np.stack([H.synth_frame(cfg, i)[0] for i in range(min(16, B))])
tpls = oa.load_templates([os.path.join(oa.TEMPLATE_DIR, n + ".png") for n in H.TEMPLATE_ORDER])
cam = oa.default_camera(cfg.width, cfg.height)
det = oa.Detector(cfg.width, cfg.height, max_batch=B)
det.set_templates(tpls); det.set_camera(cam)
gate = oa.Gate(2, 0) if os.environ.get("TRAFFIC_GATE", "1") == "1" else None   # a context with a gate takes the launch geometry bench.py's contexts take
if gate is not None:
    det.set_gate(gate)
assert oa.hip_lib().ocvar_hip_debug_calibrate(det._ctx, CAL) == 0
d = torch.from_numpy(base).cuda().repeat((B + len(base) - 1) // len(base), 1, 1, 1)[:B].contiguous()   # tiled on the device
torch.cuda.synchronize()
for _ in range(4):
    det.detect_device(d.data_ptr(), cfg.width, cfg.height, B)
print("traffic_driver: batch", B, "calibration bytes", CAL, "crop pixels/frame", det.counters()[4] / B)
