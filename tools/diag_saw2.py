import faulthandler, os, sys
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import helpers as H
import opencv_ar_amd as oa
src = open(os.path.join(ROOT, "tests", "test_gpu_parity.py")).read()
ns = {"np": np, "H": H}
exec("def _sawtooth_frame" + src.split("def _sawtooth_frame")[1].split("def test_borders")[0], ns)
w, h = 1280, 960
frames = np.stack([ns["_sawtooth_frame"](w, h, 6, [(100, 100, 700, 500), (800, 150, 1200, 900), (150, 600, 650, 880)]),
                   ns["_sawtooth_frame"](w, h, 4, [(60, 60, 1220, 900)])])
tpls = H.oracle_templates(["2x2-01"]); cam = H.oracle_camera(w, h)
det = oa.Detector(w, h, max_batch=2)
det.set_templates([oa.Template.from_buffer_copy(bytes(t)) for t in tpls]); det.set_camera(oa.Camera.from_buffer_copy(bytes(cam)))
print("detect batch of 2", flush=True)
m, c = det.detect_host(frames.copy())
print("done", c, flush=True)
for f in range(2):
    print("gray", f, flush=True); det.debug_gray(f, w, h)
    print("binary", f, flush=True); det.debug_binary(f, w, h)
    print("quads", f, flush=True); det.debug_frame_quads(f)
    print("cands", f, flush=True); det.debug_candidates(f)
crop = np.ascontiguousarray(frames[0][80:540, 80:740, 0])
print("find_squares crop", crop.shape, flush=True)
got, n = det.find_squares(crop)
print("find_squares done", n, flush=True)
