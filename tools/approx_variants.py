#!/usr/bin/env python3
"""How much does the assumed OpenCV release of cvApproxPoly matter?  (CPU only; the oracle is the instrument, not the product.)

The reference says "OpenCV 2.3 or newer" (/root/reference/README.md:18-19) and calls cvApproxPoly(CV_POLY_APPROX_DP, 2 % of the
perimeter) per contour (/root/reference/src/opencvar.cpp:190-192).  The oracle -- and the HIP path, which is bit-exact against
it -- restate the legacy C routine of OpenCV <= 2.4.3 (accuracy passed as float, integer distances).  From 2.4.4 on cvApproxPoly
wraps the templated approxPolyDP_<int> (accuracy double, distances double), and later releases add a clean-up condition.
oracle/orc_contours.cpp carries all three (orc_set_approx_variant); this tool runs whole registrations of synthetic frames
under each and counts the frames whose frame-pass quads, pre-dedupe candidates or final markers differ from variant 0.

    python tools/approx_variants.py [n_frames=10000] [workers=8]
"""
import ctypes as C
import os
import sys
from multiprocessing import Pool

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import helpers as H

SIZES = [(640, 480), (641, 479), (800, 600), (1001, 701), (1280, 720), (333, 517), (1920, 1080)]


def scene(i):
    rng = np.random.default_rng(1000 + i)
    w, h = SIZES[int(rng.integers(len(SIZES)))]
    side_min = int(rng.integers(40, 120))
    cell = max(side_min + 60, 140)
    gx, gy = max(1, min(6, w // (cell + 40))), max(1, min(4, h // (cell + 40)))
    cfg = H.synth_config(3, width=w, height=h, grid_x=gx, grid_y=gy, side_min=side_min, side_max=side_min + int(rng.integers(0, 60)),
                         rot_mode=int(rng.integers(0, 3)), corner_jitter_pct=int(rng.integers(0, 12)),
                         occlude_pct=int(rng.choice([0, 0, 20, 50])), textured=int(rng.integers(0, 2)))
    img = H.synth_frame(cfg, int(rng.integers(0, 1 << 20)))[0].astype(np.int32)
    mode = int(rng.integers(0, 4))
    if mode == 1:
        img += rng.integers(-9, 10, img.shape)
    elif mode == 2:
        img = img * int(rng.integers(30, 90)) // 100 + int(rng.integers(0, 80))
    elif mode == 3:
        img = (img + np.roll(img, 1, 1) + np.roll(img, -1, 1)) // 3
    return cfg, np.clip(img, 0, 255).astype(np.uint8)


def work(i):
    cfg, frame = scene(i)
    tpls = H.oracle_templates()
    cam = H.oracle_camera(cfg.width, cfg.height)
    o = H.oracle()
    res = []
    for v in (0, 1, 2):
        o.orc_set_approx_variant(v)
        m, c, grey = H.oracle_registration(frame, tpls, cam)
        q = H.oracle_find_squares(np.ascontiguousarray(grey[..., 0]))
        res.append((q.tobytes(),
                    [(a.markerId, a.templateId, a.orient, a.bit, tuple(a.square)) for a in c],
                    [(a.markerId, a.templateId, a.score, tuple(a.square)) for a in m], len(q), len(c), len(m)))
    o.orc_set_approx_variant(0)
    out = []
    for v in (1, 2):
        out.append((res[v][0] != res[0][0], res[v][1] != res[0][1], res[v][2] != res[0][2]))
    return out, res[0][3], res[0][4], res[0][5]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    workers = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    diff = np.zeros((2, 3), np.int64)
    quads = cands = markers = 0
    with Pool(workers) as pool:
        for k, (d, nq, nc, nm) in enumerate(pool.imap_unordered(work, range(n), chunksize=8)):
            diff += np.array(d, np.int64)
            quads += nq
            cands += nc
            markers += nm
            if (k + 1) % 1000 == 0:
                print(f"{k + 1} frames: differing frames vs variant 0 (quads, candidates, markers): v1 {diff[0].tolist()}, v2 {diff[1].tolist()}", flush=True)
    print(f"approx_variants: {n} frames, {quads} frame-pass quads, {cands} candidates, {markers} markers under variant 0 (<= 2.4.3, float accuracy)")
    for v, name in ((0, "variant 1 (2.4.4+ approxPolyDP_<int>, double accuracy)"), (1, "variant 2 (+ successive_inner_product clean-up condition)")):
        print(f"  {name}: frames with different quads {diff[v][0]}, different candidates {diff[v][1]}, different final markers {diff[v][2]}")


if __name__ == "__main__":
    main()
