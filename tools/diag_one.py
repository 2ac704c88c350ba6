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
print("detecting", flush=True)
t0 = time.time()
m, c = det.detect_host(frame[None].copy())
print("done %.3fs" % (time.time() - t0), c, det.stage_ms(), det.counters(), flush=True)
if os.environ.get("OCVAR_DBG_FOLLOW_STOP") == "8":
    import ctypes as C, torch
    B = int(os.environ.get("DIAG_BATCH", "64"))
    frames = np.stack([H.synth_frame(cfg, i, names)[0] for i in range(16)] * (B // 16))
    det2 = oa.Detector(cfg.width, cfg.height, max_batch=B)
    det2.set_templates([oa.Template.from_buffer_copy(bytes(t)) for t in tpls]); det2.set_camera(oa.Camera.from_buffer_copy(bytes(cam)))
    d = torch.from_numpy(frames).cuda()
    for _ in range(3):
        det2.detect_device(d.data_ptr(), cfg.width, cfg.height, B)
    print(det2.stage_ms())
    buf = np.zeros((8192, 4), np.int64)
    oa.hip_lib().ocvar_hip_debug_waves(det2._ctx, C.c_void_p(buf.ctypes.data))
    print("tier2 max cycles: trace %d stats %d approx %d maxpts %d" % tuple(buf[8000]))
    for name, sl in (("frame", buf[:4096]), ("crop", buf[4096:8000])):
        act = sl[sl[:, 1] > 0]
        order = np.argsort(-act[:, 0])[:8]
        print(name, "waves with tickets", len(act), "n_mid", act[0, 2] if len(act) else 0)
        for i in order:
            print("   cycles %d (%.3f ms @100MHz ticks?) tickets %d" % (act[i, 0], act[i, 0] / 1e5, act[i, 1]))
        t0 = sl[sl[:, 3] > 0][:, 3]
        print("   start spread", (t0.max() - t0.min()) if len(t0) else 0)
