"""CPU tests of the device algorithm cores (opencv-ar_amd/csrc/*_core.h compiled for the host, tests/emul) against
the oracle.  This checks the MI355X formulation itself -- parallel border starts instead of the sequential scan,
neighbour-mask follower, early-out approximation, Cholesky/Newton pose -- without a GPU.  The host build is test-only."""
import ctypes as C

import numpy as np

import helpers as H
from helpers import P


def emul_contours(em, b):
    b = np.ascontiguousarray(b)
    h, w = b.shape
    maxp, maxc = 4 * b.size + 16, b.size + 16
    pts = np.zeros(2 * maxp, np.int32)
    offs = np.zeros(maxc + 1, np.int32)
    starts = np.zeros(maxc, np.int32)
    holes = np.zeros(maxc, np.int32)
    n = em.emul_find_contours(P(b), w, h, P(pts), maxp, P(offs), P(starts), P(holes), maxc, None)
    assert n >= 0
    return [pts[2 * offs[i]:2 * offs[i + 1]].reshape(-1, 2).copy() for i in range(n)], starts[:n].copy(), holes[:n].copy()


def same_contours(a, b):
    (c1, s1, h1), (c2, s2, h2) = a, b
    return len(c1) == len(c2) and (s1 == s2).all() and (h1 == h2).all() and all(
        x.shape == y.shape and (x == y).all() for x, y in zip(c1, c2))


def test_parallel_border_starts_equal_sequential_scan(emul):
    rng = np.random.default_rng(1)
    for trial in range(300):
        emul.emul_set_run(trial & 1)  # with and without straight-run skipping
        emul.emul_set_lean((trial >> 1) % 3)  # 1: the lean follower, 2: its straight-line form (tiers 1/2 on the GPU)
        h, w = int(rng.integers(3, 40)), int(rng.integers(3, 40))
        dens = rng.choice([0.1, 0.3, 0.5, 0.6, 0.7, 0.9])
        b = (rng.random((h, w)) < dens).astype(np.uint8) * 255
        if trial % 5 == 0:
            b = np.kron((rng.random((h // 3 + 1, w // 3 + 1)) < dens).astype(np.uint8), np.ones((3, 3), np.uint8))[:h, :w] * 255
        assert same_contours(H.oracle_contours(b), emul_contours(emul, b)), trial
    b = (rng.random((240, 320)) < 0.55).astype(np.uint8) * 255
    for lean in (1, 2):
        emul.emul_set_lean(lean)
        assert same_contours(H.oracle_contours(b), emul_contours(emul, b))
    emul.emul_set_lean(0)


def emul_squares(em, gray):
    b = H.oracle_binarise(gray)
    sh, sw = b.shape
    h, w = gray.shape
    q = np.zeros(8 * 4096, np.int32)
    st = np.zeros(5, np.int64)
    n = em.emul_find_squares_bin(P(b), sw, sh, w, h, P(q), 4096, P(st))
    assert n >= 0
    return q[:8 * n].reshape(n, 4, 2)


def test_follow_approx_filter_on_synthetic_frames_and_crops(emul):
    for lean in (0, 1):  # 1: the lean store-while-following flow of tiers 2/3
        emul.emul_set_lean(lean)
        _follow_approx_filter(emul)
    emul.emul_set_lean(0)


def _follow_approx_filter(emul):
    for cid, tex, names in [(2, 0, ["2x2-01"]), (3, 0, None), (3, 1, None)]:
        cfg = H.synth_config(cid, textured=tex)
        bgr, _ = H.synth_frame(cfg, 1, names)
        gray = np.ascontiguousarray(bgr[..., 0])
        q1 = H.oracle_find_squares(gray)
        q2 = emul_squares(emul, gray)
        assert q1.shape == q2.shape and (q1 == q2).all()
        assert len(q1) >= 2 * len(names or H.TEMPLATE_ORDER)
        for qd in q1[:12]:
            x0, y0 = max(qd[:, 0].min() - 5, 0), max(qd[:, 1].min() - 5, 0)
            x1, y1 = min(qd[:, 0].max() + 5, cfg.width), min(qd[:, 1].max() + 5, cfg.height)
            crop = np.ascontiguousarray(gray[y0:y1, x0:x1])
            a, b = H.oracle_find_squares(crop), emul_squares(emul, crop)
            assert a.shape == b.shape and (a == b).all()


def test_approx_poly_random_polygons(emul):
    rng = np.random.default_rng(3)
    o = H.oracle()
    for _ in range(300):
        n = int(rng.integers(1, 60))
        ang = np.sort(rng.uniform(0, 2 * np.pi, n))
        r = rng.uniform(20, 200) * (1 + 0.05 * rng.standard_normal(n))
        pts = np.stack([300 + r * np.cos(ang), 300 + r * np.sin(ang)], 1).astype(np.int32)
        pts = np.ascontiguousarray(pts)
        eps = float(o.orc_arc_length_closed(P(pts), n) * 0.02)
        d1 = np.zeros(2 * n + 4, np.int32)
        d2 = np.zeros(32, np.int32)
        m1 = o.orc_approx_poly(P(pts), n, eps, P(d1))
        m2 = emul.emul_approx_poly(P(pts), n, eps, P(d2))
        if m2 > 8:
            assert m1 >= 5  # early-out: more than 8 raw vertices, only "more than 4 after clean-up" is promised
        else:
            assert m1 == m2 and (d1[:2 * m1] == d2[:2 * m1]).all()


def test_decode_and_pose_cores(emul):
    o = H.oracle()
    nbits = 0
    negative = 0
    # (third case: code grids 5x5 .. 8x8 -- the widthStep-8 side of the stride quirk for widths 5-7, 64-cell codes whose first
    # cell is the long long's sign bit; opencvar.h:174-175, acmath.cpp:546-554)
    for cid, names in [(2, ["2x2-01"]), (3, None), (0, H.BIG_TEMPLATES)]:
        cfg = H.synth_config(cid) if cid else H.synth_config(3, width=1280, height=720, grid_x=4, grid_y=2, rot_mode=0)
        tp = H.oracle_templates(names)
        cam = H.oracle_camera(cfg.width, cfg.height)
        bgr, _ = H.synth_frame(cfg, 2, names)
        markers, cands, grey = H.oracle_registration(bgr, tp, cam)
        gray = np.ascontiguousarray(grey[..., 0])
        quads = H.oracle_find_squares(gray)
        for c in cands:
            qd = quads[c.markerId]
            x0, y0 = max(qd[:, 0].min() - 5, 0), max(qd[:, 1].min() - 5, 0)
            x1, y1 = min(qd[:, 0].max() + 5, cfg.width), min(qd[:, 1].max() + 5, cfg.height)
            crop = gray[y0:y1, x0:x1]
            t = tp[c.templateId]
            pp = np.array(c.patPoint, np.float32)
            bit = emul.emul_read_code(C.c_void_p(crop.ctypes.data), int(x1 - x0), int(y1 - y0), cfg.width, P(pp), t.width, t.height)
            assert bit == c.bit
            nbits += 1
            negative += bit < 0
            sq = np.array(c.square, np.float32)
            g1, g2 = np.zeros(16), np.zeros(16)
            o.orc_square_to_matrix(P(sq), C.byref(cam), 1.0, P(g1))
            emul.emul_square_to_glmatrix(P(sq), C.byref(cam), 1.0, P(g2))
            assert np.abs(g1 - g2).max() <= 1e-6 * max(1.0, np.abs(g1).max())
    assert nbits > 40 and negative >= 2


DIST = [-0.21, 0.09, 0.0015, -0.0008, -0.02]   # k1 k2 p1 p2 k3 of a moderately distorting lens


def _project_rect(ratio, rvec, tvec, K, k):
    """numpy statement of the camera model cvFindExtrinsicCameraParams2 fits (pinhole + 5-coefficient distortion)"""
    th = np.linalg.norm(rvec)
    kx = rvec / th
    Kx = np.array([[0, -kx[2], kx[1]], [kx[2], 0, -kx[0]], [-kx[1], kx[0], 0]])
    R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
    out = []
    for X, Y in [(-ratio, -1), (ratio, -1), (ratio, 1), (-ratio, 1)]:
        p = R @ np.array([X, Y, 0.0]) + tvec
        x, y = p[0] / p[2], p[1] / p[2]
        r2 = x * x + y * y
        cd = 1 + k[0] * r2 + k[1] * r2 ** 2 + k[4] * r2 ** 3
        xd = x * cd + 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x)
        yd = y * cd + k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y
        out.append([xd * K[0] + K[2], yd * K[4] + K[5]])
    return np.array(out), R


def test_pose_with_lens_distortion(emul):
    """CvarCamera::distCoeffs reach cvFindExtrinsicCameraParams2 in the reference (opencvar.cpp:261-272).  A known pose is
    projected through the full camera model; both the oracle and the device core must recover it (the quad is exact, so
    the reprojection minimum is the pose itself), and agree with each other far inside the 1e-4 bar."""
    o = H.oracle()
    o.orc_gl_matrix.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(5)
    cam = H.oracle_camera(1280, 720)
    for i in range(5):
        cam.distCoeffs[i] = DIST[i]
    K = np.array(list(cam.cameraMatrix))
    for trial in range(40):
        rvec = rng.normal(size=3) * 0.5
        rvec[2] += rng.uniform(-2, 2)
        tvec = np.array([rng.uniform(-3, 3), rng.uniform(-2, 2), rng.uniform(8, 25)])
        ratio = float(rng.choice([1.0, 0.75]))
        img, R = _project_rect(ratio, rvec, tvec, K, DIST)
        sq = np.ascontiguousarray(img.astype(np.float32).reshape(-1))
        # ground truth from the float32-rounded corners is the pose up to that rounding: compare loosely with it, tightly
        # between the two solvers
        want = np.zeros(16)
        o.orc_gl_matrix(P(np.ascontiguousarray(R.reshape(-1))), P(tvec), P(want))
        g1, g2 = np.zeros(16), np.zeros(16)
        o.orc_square_to_matrix(P(sq), C.byref(cam), C.c_double(ratio), P(g1))
        emul.emul_square_to_glmatrix(P(sq), C.byref(cam), C.c_double(ratio), P(g2))
        scale = max(1.0, np.abs(want).max())
        assert np.abs(g1 - want).max() <= 2e-3 * scale, (trial, g1, want)
        assert np.abs(g1 - g2).max() <= 1e-6 * scale, trial
    # and the coefficients matter: the pinhole solution for the same corners is a different pose
    pin = H.oracle_camera(1280, 720)
    g3 = np.zeros(16)
    o.orc_square_to_matrix(P(sq), C.byref(pin), C.c_double(ratio), P(g3))
    assert np.abs(g3 - g1).max() > 1e-3


def test_grey_plane_panels(emul):
    """hd.h::gray_col / gray_pitch: panel p of a row holds columns 240 p - 8 .. 240 p + 247 in 256 bytes (what one wave of the frame
    kernel converts).  The properties the readers rely on: home offsets are distinct and inside the row; from any column the next
    8 columns follow contiguously (a crop lane's 4-byte load, decode's two neighbouring samples); a column within 8 of a panel's
    start is the 8-column tail of the panel before it shifted by one panel (where the frame kernel's halo lanes -- and the
    odd-column kernel -- keep the second copy)."""
    import ctypes as C
    emul.emul_gray_col.restype = C.c_uint
    for W in (16, 239, 240, 241, 243, 480, 487, 640, 1001, 1920, 3840):
        pitch = emul.emul_gray_pitch(W)
        assert pitch % 256 == 0 and pitch >= ((W + 239) // 240) * 256
        offs = [emul.emul_gray_col(x) for x in range(W)]
        assert len(set(offs)) == W and max(offs) < pitch and min(offs) == 8
        for x in range(W):
            p = x // 240
            assert offs[x] == p * 256 + (x - 240 * p) + 8
            # the 256 bytes of panel p start at column 240 p - 8: a run that begins at x stays inside panel p for 9 columns
            assert offs[x] + 8 <= p * 256 + 255
            if x >= 240 and x % 240 < 8:   # second copy: same column, seen from the panel before
                assert emul.emul_gray_col(x - 8) + 8 == (p - 1) * 256 + 248 + x % 240


def test_pose_cores_agree_over_aspect_ratios(emul):
    """Device pose core (Cholesky LM, closed-form seed) against the oracle's (Jacobi pseudo-inverse) on random perturbed squares in
    frames from 1:2 to 4:1 -- cvarCameraScale stretches the 4:3 calibration to the frame, so the aspect ratio is the conditioning of
    the fit.  (Beyond 1:2.5 in portrait a few per cent of such quads end in different places: DESIGN.md section 5.)"""
    o = H.oracle()
    emul.emul_square_to_glmatrix.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]
    rng = np.random.default_rng(7)
    n = 0
    for asp in (0.5, 0.75, 1.0, 1.33, 1.78, 2.0, 3.0, 4.0):
        for _ in range(60):
            h = int(rng.integers(200, 1000))
            w = int(h * asp)
            s = rng.uniform(30, 160)
            cx, cy, a = rng.uniform(s, max(s + 1, w - s)), rng.uniform(s, max(s + 1, h - s)), rng.uniform(0, 2 * np.pi)
            base = np.array([[-1, -1], [1, -1], [1, 1], [-1, 1]], float) * s / 2
            rot = np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]])
            q = (base @ rot.T) * np.array([1.0, rng.uniform(0.5, 1.0)]) + rng.normal(0, s * 0.04, (4, 2)) + [cx, cy]
            sq = np.round(q).astype(np.float32).reshape(-1)
            cam = H.oracle_camera(w, h)
            g1, g2 = np.zeros(16), np.zeros(16)
            o.orc_square_to_matrix(P(sq), C.byref(cam), 1.0, P(g1))
            emul.emul_square_to_glmatrix(P(sq), C.byref(cam), 1.0, P(g2))
            assert np.abs(g1 - g2).max() <= 1e-4 * max(1.0, np.abs(g1).max()), (asp, w, h, sq.tolist())
            n += 1
    assert n == 480
