#!/usr/bin/env python3
"""Where does the device's grey plane / binary image differ from the oracle's?  (debugging aid for binarise.hip; GPU box)
usage: diag_binarise.py [config_id] [frame_index]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H
import torch
import opencv_ar_amd as oa

cid = int(sys.argv[1]) if len(sys.argv) > 1 else 2
fi = int(sys.argv[2]) if len(sys.argv) > 2 else 0
cfg = H.synth_config(cid)
names = ["2x2-01"]
frame = H.synth_frame(cfg, fi, names)[0]
tpls = H.oracle_templates(names); cam = H.oracle_camera(cfg.width, cfg.height)
det = oa.Detector(cfg.width, cfg.height, max_batch=1)
det.set_templates([oa.Template.from_buffer_copy(bytes(t)) for t in tpls])
det.set_camera(oa.Camera.from_buffer_copy(bytes(cam)))
d = torch.from_numpy(frame[None]).to("cuda:0")
try:
    det.detect_device(d.data_ptr(), cfg.width, cfg.height, 1)
except Exception as e:
    print("detect failed:", e)
_, _, grey = H.oracle_registration(frame, tpls, cam)
g_ref = np.ascontiguousarray(grey[..., 0])
g = det.debug_gray(0, cfg.width, cfg.height)
dg = g != g_ref
print("grey diffs:", int(dg.sum()))
if dg.any():
    ys, xs = np.nonzero(dg); print("  rows", ys.min(), ys.max(), "cols", xs.min(), xs.max())
b_ref = H.oracle_binarise(g_ref)
b = det.debug_binary(0, cfg.width, cfg.height)
db = (b != b_ref); db[0, :] = db[-1, :] = False; db[:, 0] = db[:, -1] = False
print("binary diffs:", int(db.sum()), "of", db.size)
if db.any():
    ys, xs = np.nonzero(db)
    print("  rows", ys.min(), ys.max(), "cols", xs.min(), xs.max())
    print("  by column mod 4:", np.bincount(xs % 4, minlength=4).tolist())
    print("  by row mod 4:", np.bincount(ys % 4, minlength=4).tolist())
    print("  by strip (240 cols):", np.bincount(xs // 240).tolist())
    print("  first 10:", list(zip(ys[:10].tolist(), xs[:10].tolist())))
    print("  rows hist (per 32):", np.bincount(ys // 32).tolist())

# all eight neighbour bits, from the (zero-framed) binary image
bz = (b_ref > 0).astype(np.uint8); bz[0, :] = bz[-1, :] = 0; bz[:, 0] = bz[:, -1] = 0
pad = np.pad(bz, 1)
dx = [1, 1, 0, -1, -1, -1, 0, 1]; dy = [0, -1, -1, -1, 0, 1, 1, 1]
hh, ww = bz.shape
exp = np.zeros_like(bz)
for k in range(8):
    exp |= pad[1 + dy[k]:1 + dy[k] + hh, 1 + dx[k]:1 + dx[k] + ww] << k
m = det.debug_masks(0, cfg.width, cfg.height)
dm = m ^ exp
print("mask diffs (pixels):", int((dm != 0).sum()))
if dm.any():
    ys, xs = np.nonzero(dm)
    print("  rows", ys.min(), ys.max(), "cols", xs.min(), xs.max())
    print("  by bit:", [int(((dm >> k) & 1).sum()) for k in range(8)])
    print("  by column mod 4:", np.bincount(xs % 4, minlength=4).tolist(), " by row mod 8:", np.bincount(ys % 8, minlength=8).tolist())
    print("  by strip:", np.bincount(xs // 240).tolist())
    print("  first:", [(int(y), int(x), hex(int(m[y, x])), hex(int(exp[y, x]))) for y, x in list(zip(ys, xs))[:12]])
