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
w, h = (3840, 2160) if (len(sys.argv) > 1 and sys.argv[1] == '2') else (1280, 960)
which = int(sys.argv[1]) if len(sys.argv) > 1 else 1
frame = ns["_sawtooth_frame"](w, h, 6, [(100, 100, 700, 500), (800, 150, 1200, 900), (150, 600, 650, 880)]) if which == 0 else (ns["_sawtooth_frame"](w, h, 4, [(60, 60, 1220, 900)]) if which == 1 else ns["_sawtooth_frame"](w, h, 4, [(40, 40, 3800, 2120)]))
cfg = H.synth_config(2, width=w, height=h)
tpls = H.oracle_templates(["2x2-01"]); cam = H.oracle_camera(w, h)
det = oa.Detector(w, h, max_batch=1)
det.set_templates([oa.Template.from_buffer_copy(bytes(t)) for t in tpls]); det.set_camera(oa.Camera.from_buffer_copy(bytes(cam)))
print("detect", which, flush=True)
m, c = det.detect_host(frame[None].copy())
print("done", c, det.stage_ms(), det.counters(), flush=True)
