#!/usr/bin/env python3
"""Randomised parity sweep on an MI355X: many synthetic scenes (sizes incl. odd ones, clean/textured backgrounds,
marker sizes, rotations, perspective jitter, occlusion, 1..4 templates (shipped 2x2/3x3/4x4, the pinned 5x5..8x8 grids, random
grids of any size up to 8x8), post-processing noise/blur/contrast) through
the HIP path and the oracle, compared with the same bars as tests/test_gpu_parity.py (grey plane, binary image, quads,
decoded candidates bit-exact; markers exact / pose 1e-4).  Not part of the default test run (minutes of CPU oracle time).

    python tools/fuzz_parity.py [n_scenes] [seed]      (FUZZ_RANDOM_SIZES=1: random frame sizes instead of the fixed list)
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import helpers as H
import opencv_ar_amd as oa
import test_gpu_parity as T

n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
sizes = [(640, 480), (641, 479), (800, 600), (1001, 701), (1280, 720), (333, 517), (1920, 1080), (96, 64), (250, 250)]
t0 = time.time()
frames_checked = markers_seen = cands_seen = 0
for s in range(n_scenes):
    w, h = sizes[int(rng.integers(len(sizes)))]
    if os.environ.get("FUZZ_RANDOM_SIZES") == "1":   # any size: widths around the grey plane's panel boundaries (240 k +- 9) are drawn more often
        w = int(rng.integers(200, 1500))
        if rng.integers(3) == 0:
            w = 240 * int(rng.integers(1, 7)) + int(rng.integers(-9, 10))
        # (aspect ratios up to ~2:1 either way: cvarCameraScale stretches the 4:3 calibration to the frame, and at 234x1050 -- focal
        # lengths 6:1 -- the planar pose fit is so ill-conditioned that oracle and device, two independent solvers, end in different
        # places far from any valley: 1 marker in 1300 of an unrestricted sweep; every discrete output still agreed)
        h = int(rng.integers(max(160, w // 2), min(1100, 2 * w) + 1))
    if os.environ.get("FUZZ_RANDOM_SIZES") == "3":   # large frames (the marker grid below grows with them: up to 6 x 4 markers)
        w, h = [(3840, 2160), (2560, 1440), (2048, 1536), (3001, 1701)][int(rng.integers(4))]
    if os.environ.get("FUZZ_RANDOM_SIZES") == "2":   # small frames: crops and frames a few strips or less wide, markers cut by the edge
        w = int(rng.integers(64, 320))
        h = int(rng.integers(max(64, w // 2), min(320, 2 * w) + 1))
    names = [H.TEMPLATE_ORDER[i] for i in sorted(rng.choice(3, size=int(rng.integers(1, 4)), replace=False))]
    if s % 2:   # every other scene: code grids up to 8x8 (opencvar.h:174-175) -- the pinned synthetic ones and freshly drawn grids
        pool = list(H.BIG_TEMPLATES)
        for k in range(2):
            n = int(rng.integers(2, 9))
            H.register_template(f"rnd{s}-{k}", rng.integers(0, 2, (n, n)))
            pool.append(f"rnd{s}-{k}")
        names = names[:1] + [pool[i] for i in rng.choice(len(pool), size=int(rng.integers(1, 4)), replace=False)]
    side_min = int(rng.integers(40, 120))
    cell = max(side_min + 60, 140)
    gx, gy = max(1, min(6, w // (cell + 40))), max(1, min(4, h // (cell + 40)))
    cfg = H.synth_config(3, width=w, height=h, grid_x=gx, grid_y=gy, side_min=side_min, side_max=side_min + int(rng.integers(0, 60)),
                         rot_mode=int(rng.integers(0, 3)), corner_jitter_pct=int(rng.integers(0, 12)),
                         occlude_pct=int(rng.choice([0, 0, 20, 50])), textured=int(rng.integers(0, 2)))
    nb = int(rng.integers(1, 4))
    frames = []
    for f in range(nb):
        img = H.synth_frame(cfg, int(rng.integers(0, 1 << 20)), names)[0].astype(np.int32)
        mode = int(rng.integers(0, 5))
        if mode == 1:      # sensor noise, per channel (exercises BGR2GRAY with unequal channels)
            img += rng.integers(-9, 10, img.shape)
        elif mode == 2:    # low contrast + offset
            img = img * int(rng.integers(30, 90)) // 100 + int(rng.integers(0, 80))
        elif mode == 3:    # colour cast
            img = img * np.array([rng.integers(60, 101), rng.integers(60, 101), rng.integers(60, 101)]) // 100
        elif mode == 4:    # 3-tap blur along x
            img = (img + np.roll(img, 1, 1) + np.roll(img, -1, 1)) // 3
        frames.append(np.clip(img, 0, 255).astype(np.uint8))
    frames = np.ascontiguousarray(np.stack(frames))
    det, tpls, cam = T.make_detector(oa, cfg, names, nb)
    if os.environ.get("FUZZ_VERBOSE") == "1":
        print(f"scene {s}: {w}x{h}, {nb} frames, templates {names}", flush=True)
    # the crop pass's two forms (follow.hip::follow_mid_kernel): the batch-size default, one launch, two launches with pruning
    det.set_tuning(crop_phases=s % 3)   # 0: the default for the batch size
    if s % 4 == 3:     # stateful: every lane is a video stream, the scene drifts a few pixels per step (opencvar.cpp:635-668)
        from opencv_ar_amd.tracking import StreamTracker
        tracker = StreamTracker(det, nb)
        prev = [None] * nb
        cur = frames
        for step in range(3):
            markers, counts = tracker.step(cur.copy())
            for f in range(nb):
                ref_m, n_c = T.check_frame(det, f, cur[f], tpls, cam, markers, counts, prev=prev[f])
                prev[f] = ref_m
                frames_checked += 1; markers_seen += len(ref_m); cands_seen += n_c
            cur = np.ascontiguousarray(np.roll(cur, (int(rng.integers(-6, 7)), int(rng.integers(-6, 7))), axis=(1, 2)))
    else:
        markers, counts = det.detect_host(frames.copy())
        for f in range(nb):
            ref_m, n_c = T.check_frame(det, f, frames[f], tpls, cam, markers, counts)
            frames_checked += 1; markers_seen += len(ref_m); cands_seen += n_c
    del det
    if (s + 1) % 10 == 0:
        print(f"scene {s + 1}/{n_scenes}: {frames_checked} frames equal so far ({cands_seen} decoded candidates, {markers_seen} markers), {time.time() - t0:.0f} s", flush=True)
print(f"fuzz_parity: {frames_checked} frames, {cands_seen} candidates, {markers_seen} markers: all equal")
