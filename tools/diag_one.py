"""Diagnostic: one 1080p frame through the detector with progress prints (used under `timeout`)."""
import faulthandler, os, sys, time
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import helpers as H
import opencv_ar_amd as oa
cfg = H.synth_config(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
names = None if cfg.width > 640 else ["2x2-01"]
frame, _ = H.synth_frame(cfg, 0, names)
tpls = H.oracle_templates(names); cam = H.oracle_camera(cfg.width, cfg.height)
print("creating", flush=True)
det = oa.Detector(cfg.width, cfg.height, max_batch=1)
det.set_templates([oa.Template.from_buffer_copy(bytes(t)) for t in tpls]); det.set_camera(oa.Camera.from_buffer_copy(bytes(cam)))
print("find_squares (frame pass only)", flush=True)
q, nq = det.find_squares(np.ascontiguousarray(frame[..., 0]))
print("frame pass ok", nq, det.stage_ms(), flush=True)
print("detecting", flush=True)
t0 = time.time()
m, c = det.detect_host(frame[None].copy())
print("done %.3fs" % (time.time() - t0), c, det.stage_ms(), det.counters(), flush=True)
