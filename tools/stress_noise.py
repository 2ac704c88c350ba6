import os, sys
ROOT = os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import helpers as H
import opencv_ar_amd as oa
import test_gpu_parity as T
rng = np.random.default_rng(0)
w, h = 1920, 1080
cfg = H.synth_config(3)
tpls = oa.load_templates([os.path.join(oa.TEMPLATE_DIR, n + ".png") for n in H.TEMPLATE_ORDER]); cam = oa.default_camera(w, h)
for B in (1, 4):
    det = oa.Detector(w, h, max_batch=B); det.set_templates(tpls); det.set_camera(cam)
    for name, gen in (("uniform noise", lambda: np.repeat(rng.integers(0, 256, (h, w, 1), np.uint8), 3, axis=2)),
                      ("blocky noise 4px", lambda: np.kron(rng.integers(0, 2, (h // 4, w // 4, 1), np.uint8) * 255, np.ones((4, 4, 3), np.uint8))),
                      ("blocky noise 8px", lambda: np.kron(rng.integers(0, 2, (h // 8, w // 8, 1), np.uint8) * 255, np.ones((8, 8, 3), np.uint8))),
                      ("checker 16px", lambda: np.kron(np.indices((h // 16 + 1, w // 16 + 1)).sum(0)[..., None] % 2 * 255, np.ones((16, 16, 3))).astype(np.uint8)[:h, :w])):
        frames = np.ascontiguousarray(np.stack([gen() for _ in range(B)]))
        try:
            m, c = det.detect_host(frames.copy())
            print(B, name, "ok", c.tolist(), det.counters().tolist(), [round(float(x), 2) for x in det.stage_ms()][-1])
        except Exception as e:
            print(B, name, "ERR", str(e)[:120], det.counters().tolist())

# exactness on pathological content (one frame each, compared with the oracle like the parity tests)
for (w, h, cell) in ((1920, 1080, 4), (1280, 720, 4), (1920, 1080, 8), (1000, 700, 2), (3840, 2160, 4)):
    cfg = H.synth_config(3, width=w, height=h)
    det, tp, cm = T.make_detector(oa, cfg, None, 1)
    frame = np.ascontiguousarray(np.kron(rng.integers(0, 2, ((h + cell - 1) // cell, (w + cell - 1) // cell, 1), np.uint8) * 255, np.ones((cell, cell, 3), np.uint8))[:h, :w])
    m, c = det.detect_host(frame[None].copy())
    T.check_frame(det, 0, frame, tp, cm, m, c)
    print(f"blocky noise {cell}px {w}x{h}: equal to the oracle; counters", det.counters().tolist())
