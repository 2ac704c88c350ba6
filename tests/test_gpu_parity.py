"""GPU parity tests: the HIP path, called through the C ABI, against the oracle on identical seeded frames.

Bars (BASELINE.json north_star): marker/template ids, code bits, binary image, quads: bit-exact; quad corners
within 0.5 px (they are integer-valued here, so also exact); 4x4 GL pose within 1e-4 relative."""
import ctypes as C

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu

POSE_RTOL = 1e-4   # north_star: glMatrix within 1e-4 relative
CORNER_TOL = 0.5   # north_star: quad corners within 0.5 px


@pytest.fixture(scope="module")
def oa():
    import torch
    assert torch.cuda.is_available(), "gpu tests need a device"
    import opencv_ar_amd
    return opencv_ar_amd


def make_detector(oa, cfg, names, batch):
    tpls = H.oracle_templates(names)
    cam = H.oracle_camera(cfg.width, cfg.height)
    det = oa.Detector(cfg.width, cfg.height, max_batch=batch)
    det.set_templates([oa.Template.from_buffer_copy(bytes(t)) for t in tpls])
    det.set_camera(oa.Camera.from_buffer_copy(bytes(cam)))
    return det, tpls, cam


def check_frame(det, f, frame_bgr, tpls, cam, markers, counts, prev=None):
    ref_m, ref_c, grey = H.oracle_registration(frame_bgr, tpls, cam, prev=prev)
    h, w = frame_bgr.shape[:2]
    # grey plane and binary image (interior; the HIP path stores it with cvFindContours' zeroed frame)
    assert np.array_equal(det.debug_gray(f, w, h), grey[..., 0])
    b_ref = H.oracle_binarise(np.ascontiguousarray(grey[..., 0]))
    b_gpu = det.debug_binary(f, w, h)
    assert np.array_equal(b_gpu[1:-1, 1:-1], b_ref[1:-1, 1:-1])
    # every bit of the neighbour-mask plane the border followers walk on
    assert np.array_equal(det.debug_masks(f, w, h), H.neighbour_masks(b_ref))
    if prev is None:
        q_ref = H.oracle_find_squares(np.ascontiguousarray(grey[..., 0]))
        q_gpu = det.debug_frame_quads(f)
        assert q_ref.shape == q_gpu.shape and np.array_equal(q_ref, q_gpu)
    cands = det.debug_candidates(f)
    assert len(cands) == len(ref_c)
    for a, b in zip(cands, ref_c):
        assert (a.markerId, a.templateId, a.orient, a.bit) == (b.markerId, b.templateId, b.orient, b.bit)
        assert np.array_equal(np.array(a.square), np.array(b.square))
        assert np.array_equal(np.array(a.patPoint), np.array(b.patPoint))
    assert counts[f] == len(ref_m)
    for k, r in enumerate(ref_m):
        m = markers[f, k]
        assert m["templateId"] == r.templateId and m["markerId"] == r.markerId and m["score"] == r.score
        assert np.abs(m["square"] - np.array(r.square)).max() <= CORNER_TOL
        assert m["aspectRatio"] == r.aspectRatio
        g = np.array(r.glMatrix)
        assert np.abs(m["glMatrix"] - g).max() <= POSE_RTOL * max(1.0, np.abs(g).max()), (
            "pose", f, k, float(np.abs(m["glMatrix"] - g).max()), float(np.abs(g).max()), np.array(r.square).tolist(), g.tolist(), m["glMatrix"].tolist())
    return ref_m, len(ref_c)


def test_config1_single_frame_host_entry(oa):
    cfg = H.synth_config(1)
    det, tpls, cam = make_detector(oa, cfg, ["2x2-01"], 1)
    frame, _ = H.synth_frame(cfg, 0, ["2x2-01"])
    frames = frame[None].copy()
    markers, counts = det.detect_host(frames, grey_in_place=True)
    ref_m, n_c = check_frame(det, 0, frame, tpls, cam, markers, counts)
    assert len(ref_m) == 1 and n_c >= 1
    # reference side effect: the caller's frame is greyed in place (opencvar.cpp:624-627)
    _, _, grey = H.oracle_registration(frame, tpls, cam)
    assert np.array_equal(frames[0], grey)


def test_config2_batch_of_small_frames(oa):
    import torch
    cfg = H.synth_config(2)
    n = 24
    det, tpls, cam = make_detector(oa, cfg, ["2x2-01"], n)
    frames = np.stack([H.synth_frame(cfg, f, ["2x2-01"])[0] for f in range(n)])
    d = torch.from_numpy(frames).cuda()
    markers, counts = det.detect_device(d.data_ptr(), cfg.width, cfg.height, n)
    total = 0
    for f in range(n):
        total += check_frame(det, f, frames[f], tpls, cam, markers, counts)[1]
    assert total >= 3 * n  # most of the 4 planted markers per frame decode
    assert np.array_equal(d.cpu().numpy(), frames)  # not greyed unless asked


def test_config3_full_hd_three_templates(oa):
    import torch
    cfg = H.synth_config(3)
    n = 4
    det, tpls, cam = make_detector(oa, cfg, None, n)
    frames = np.stack([H.synth_frame(cfg, f)[0] for f in range(n)])
    d = torch.from_numpy(frames).cuda()
    markers, counts = det.detect_device(d.data_ptr(), cfg.width, cfg.height, n)
    for f in range(n):
        ref_m, n_c = check_frame(det, f, frames[f], tpls, cam, markers, counts)
        assert n_c >= 36 and 1 <= len(ref_m) <= 3


def test_code_grids_5x5_to_8x8(oa):
    """Code grids above the shipped 4x4 (the reference's cvarLoadTag defaults to 8x8, opencvar.h:174-175; acArray2DToBit packs
    64 cells into a long long, acmath.cpp:546-554): 5x5, 6x6, 7x7 read through the widthStep-8 stride quirk, three 8x8 grids
    -- one ballot lane per cell, all 64 lanes, codes with the sign bit set -- on (tw+2) x (th+2) patches up to 10 x 10, planted
    in all four rotations.  Templates come from cvarLoadTemplateTag on the PNG fixtures (libopencv-ar.so) and must equal the
    oracle's; every candidate's bit / orient / corners are compared through check_frame."""
    import ctypes as C
    import os
    import torch
    names = H.BIG_TEMPLATES
    cfg = H.synth_config(3, width=1280, height=720, grid_x=4, grid_y=2, rot_mode=0)
    n = 12
    det, tpls, cam = make_detector(oa, cfg, names, n)
    loaded = oa.load_templates([os.path.join(H.ROOT, "tests", "golden", "png", nm + ".png") for nm in names])
    for a, b in zip(loaded, tpls):
        assert (a.width, a.height, list(a.code)) == (b.width, b.height, list(b.code))
    det.set_templates(loaded)
    frames = np.stack([H.synth_frame(cfg, 40 + f, names)[0] for f in range(n)])
    d = torch.from_numpy(frames).cuda()
    markers, counts = det.detect_device(d.data_ptr(), cfg.width, cfg.height, n)
    seen, negative = {}, 0
    for f in range(n):
        check_frame(det, f, frames[f], tpls, cam, markers, counts)
        for c in det.debug_candidates(f):
            if c.orient:
                seen.setdefault(names[c.templateId], set()).add(c.orient)
                negative += c.bit < 0
    for nm in ("8x8-s1", "8x8-neg", "8x8-corners"):
        assert len(seen.get(nm, ())) >= 3, (nm, seen.get(nm))
    assert set().union(*[seen.get(nm, set()) for nm in ("8x8-s1", "8x8-neg", "8x8-corners")]) == {1, 2, 3, 4}
    assert negative >= 4
    for nm in ("5x5-s1", "6x6-s1", "7x7-s1"):
        assert seen.get(nm) == {1}, (nm, seen.get(nm))
    # perspective + rotation anywhere + three templates of different sizes at once (mixed tw in one frame's decode loop)
    cfg2 = H.synth_config(3, width=1000, height=700, grid_x=3, grid_y=2, rot_mode=1, corner_jitter_pct=6)
    mix = ["8x8-neg", "3x3-01", "6x6-s1", "8x8-corners"]
    det2, tpls2, cam2 = make_detector(oa, cfg2, mix, 3)
    fr2 = np.stack([H.synth_frame(cfg2, 7 + f, mix)[0] for f in range(3)])
    m2, c2 = det2.detect_host(fr2.copy())
    for f in range(3):
        check_frame(det2, f, fr2[f], tpls2, cam2, m2, c2)


def test_host_entry_with_megabytes_of_frames(oa):
    """ocvar_hip_detect_host with a batch well above 8 MB (7 x 1280 x 720 x 3 = 19 MB) in an ordinary heap array whose base is
    not page-aligned, more frames than the context's batch (sub-batches of 3, the last one short), previous markers given
    (stateful, opencvar.cpp:635-668) and the reference's in-place grey (opencvar.cpp:624-627) coming back: the pipelined host
    transport (frames staged through the library's own page-locked buffers, copy of sub-batch k+1 overlapping the kernels of
    sub-batch k) against the oracle, frame by frame.  The caller's arrays are surrounded by guard bytes that must survive."""
    cfg = H.synth_config(3, width=1280, height=720, grid_x=3, grid_y=2)
    names = None
    det, tpls, cam = make_detector(oa, cfg, names, 3)
    n = 7
    w, h = cfg.width, cfg.height
    fb = w * h * 3
    guard = 64 + 3
    buf = np.full(2 * guard + n * fb, 0x5A, np.uint8)     # frames start 67 bytes into the block: unaligned in every sense
    frames = buf[guard:guard + n * fb].reshape(n, h, w, 3)
    orig = np.stack([H.synth_frame(cfg, 20 + f, names)[0] for f in range(n)])
    frames[:] = orig
    assert n * fb >= (8 << 20) and frames.ctypes.data % 4096 != 0
    # step 1 (stateless) gives the previous markers of step 2
    m1, c1 = det.detect_host(frames.copy())
    prev = [[m1[f, k] for k in range(c1[f])] for f in range(n)]
    ref_prev = [H.oracle_registration(orig[f], tpls, cam)[0] for f in range(n)]
    markers, counts = det.detect_host(frames, grey_in_place=True, prev=prev)
    assert (buf[:guard] == 0x5A).all() and (buf[guard + n * fb:] == 0x5A).all()
    last = n - 1   # the debug hooks see the last sub-batch (frame 6 = its lane 0)
    for f in range(n):
        ref_m, _, grey = H.oracle_registration(orig[f], tpls, cam, prev=ref_prev[f])
        assert np.array_equal(frames[f], grey), f
        assert counts[f] == len(ref_m), f
        for k, r in enumerate(ref_m):
            m = markers[f, k]
            assert m["templateId"] == r.templateId and m["markerId"] == r.markerId and m["score"] == r.score
            assert np.abs(m["square"] - np.array(r.square)).max() <= CORNER_TOL
            g = np.array(r.glMatrix)
            assert np.abs(m["glMatrix"] - g).max() <= POSE_RTOL * max(1.0, np.abs(g).max())
    check_frame(det, 0, orig[last], tpls, cam, markers[last:], counts[last:], prev=ref_prev[last])


def test_padded_rows_gapped_frames_and_unaligned_base(oa):
    """IplImage.widthStep may exceed 3*width and imageData need not be 4-byte aligned (opencvar.cpp:619 takes any
    8UC3 image): rows padded to 3*W+5 bytes, frames 77 bytes apart, base pointer at an odd address -- the kernel's
    byte-load path -- and grey written back in place into that layout."""
    import torch
    cfg = H.synth_config(2, width=644, height=482)
    n = 3
    det, tpls, cam = make_detector(oa, cfg, ["2x2-01"], n)
    frames = [H.synth_frame(cfg, f, ["2x2-01"])[0] for f in range(n)]
    w, h = cfg.width, cfg.height
    row_stride = 3 * w + 5
    frame_stride = row_stride * h + 77
    buf = np.full(1 + n * frame_stride, 0xA5, np.uint8)
    for f in range(n):
        rows = buf[1 + f * frame_stride: 1 + f * frame_stride + row_stride * h].reshape(h, row_stride)
        rows[:, :3 * w] = frames[f].reshape(h, 3 * w)
    d = torch.from_numpy(buf).cuda()
    assert (d.data_ptr() + 1) % 2 == 1
    markers, counts = det.detect_device(d.data_ptr() + 1, w, h, n, row_stride=row_stride, frame_stride=frame_stride,
                                        grey_in_place=True)
    out = d.cpu().numpy()
    for f in range(n):
        ref_m, n_c = check_frame(det, f, frames[f], tpls, cam, markers, counts)
        assert n_c >= 3
        _, _, grey = H.oracle_registration(frames[f], tpls, cam)
        rows = out[1 + f * frame_stride: 1 + f * frame_stride + row_stride * h].reshape(h, row_stride)
        assert np.array_equal(rows[:, :3 * w].reshape(h, w, 3), grey)
        assert (rows[:, 3 * w:] == 0xA5).all()          # padding untouched
    assert out[0] == 0xA5 and (out[1 + (n - 1) * frame_stride + row_stride * h:] == 0xA5).all()


def test_host_entry_pipelines_sub_batches(oa):
    """ocvar_hip_detect_host with more frames than the context's batch: sub-batches go through the library's page-locked
    staging buffers (the copy of the next one overlaps detection of the current one), grey comes back in place."""
    cfg = H.synth_config(2)
    det, tpls, cam = make_detector(oa, cfg, ["2x2-01"], 2)   # sub-batches of 2
    n = 5
    frames = np.stack([H.synth_frame(cfg, f, ["2x2-01"])[0] for f in range(n)])
    work = frames.copy()
    markers, counts = det.detect_host(work, grey_in_place=True)
    assert counts.shape == (n,)
    for f in range(n):
        ref_m, _, grey = H.oracle_registration(frames[f], tpls, cam)
        assert np.array_equal(work[f], grey)
        assert counts[f] == len(ref_m)
        for k, r in enumerate(ref_m):
            m = markers[f, k]
            assert m["templateId"] == r.templateId and m["markerId"] == r.markerId and m["score"] == r.score
            assert np.abs(m["square"] - np.array(r.square)).max() <= CORNER_TOL
            g = np.array(r.glMatrix)
            assert np.abs(m["glMatrix"] - g).max() <= POSE_RTOL * max(1.0, np.abs(g).max())


def test_textured_background_and_odd_size(oa):
    cfg = H.synth_config(3, textured=1, width=1001, height=701, grid_x=2, grid_y=2)
    det, tpls, cam = make_detector(oa, cfg, None, 2)
    frames = np.stack([H.synth_frame(cfg, f)[0] for f in range(2)])
    markers, counts = det.detect_host(frames.copy())
    for f in range(2):
        check_frame(det, f, frames[f], tpls, cam, markers, counts)


@pytest.mark.gpu
@pytest.mark.parametrize("width,height", [(487, 365), (729, 243), (961, 541)])
def test_odd_sizes_whose_last_column_or_row_begins_a_grey_panel(oa, width, height):
    """The grey plane is stored in panels of 240 columns that overlap by 16 (hd.h::gray_col): the odd last column -- greyed by its
    own kernel -- of these widths lies within 8 columns of a panel's start, so it is kept twice; crops reach over it."""
    cfg = H.synth_config(3, textured=1, width=width, height=height, grid_x=max(1, width // 240), grid_y=max(1, height // 240),
                         side_min=70, side_max=110)
    det, tpls, cam = make_detector(oa, cfg, None, 2)
    frames = np.stack([H.synth_frame(cfg, f)[0] for f in range(2)])
    markers, counts = det.detect_host(frames.copy())
    for f in range(2):
        check_frame(det, f, frames[f], tpls, cam, markers, counts)


def test_config5_4k_jitter_and_occlusion(oa):
    cfg = H.synth_config(5)
    det, tpls, cam = make_detector(oa, cfg, None, 1)
    frame, truth = H.synth_frame(cfg, 0)
    markers, counts = det.detect_host(frame[None].copy())
    ref_m, n_c = check_frame(det, 0, frame, tpls, cam, markers, counts)
    assert n_c >= 100


def test_empty_and_noise_frames(oa):
    rng = np.random.default_rng(0)
    w, h = 320, 240
    cfg = H.synth_config(2, width=w, height=h)
    det, tpls, cam = make_detector(oa, cfg, ["2x2-01"], 3)
    frames = np.zeros((3, h, w, 3), np.uint8)
    frames[0] = 220                                        # nothing to find
    frames[1] = rng.integers(0, 256, (h, w, 3), np.uint8)  # coloured noise: exercises BGR2GRAY + many tiny borders
    frames[2] = np.repeat(rng.integers(0, 256, (h, w, 1), np.uint8), 3, axis=2)
    markers, counts = det.detect_host(frames.copy())
    for f in range(3):
        check_frame(det, f, frames[f], tpls, cam, markers, counts)
    assert counts[0] == 0


def test_smallest_frames_and_markers_cut_by_the_frame_edge(oa):
    """Edge cases: the smallest frames the entry point accepts (16x16, 17x33: one strip, one chunk, every border rule
    at once) and views of a synthetic frame whose edges cut through markers (quads touching the zeroed 1-px frame,
    cvarFindSquares' first-vertex border rule opencvar.cpp:204-206, crops clipped by cvSetImageROI)."""
    rng = np.random.default_rng(5)
    for (w, h) in ((16, 16), (17, 33), (40, 18)):
        cfg = H.synth_config(2, width=w, height=h)
        det, tpls, cam = make_detector(oa, cfg, ["2x2-01"], 2)
        frames = np.stack([np.repeat(rng.integers(0, 256, (h, w, 1), np.uint8), 3, axis=2),
                           np.kron(rng.integers(0, 2, ((h + 3) // 4, (w + 3) // 4, 1), np.uint8) * 255, np.ones((4, 4, 3), np.uint8))[:h, :w]])
        frames = np.ascontiguousarray(frames)
        markers, counts = det.detect_host(frames.copy())
        for f in range(2):
            check_frame(det, f, frames[f], tpls, cam, markers, counts)
    cfg = H.synth_config(2)
    full, truth = H.synth_frame(cfg, 1, ["2x2-01"])
    c = truth[0]["corner"]
    cx, cy = int(c[:, 0].mean()), int(c[:, 1].mean())
    views = [full[:, cx:], full[cy:, :], full[: cy + 3, : cx + 5], full[max(0, cy - 150):, max(0, cx - 150): cx + 9]]
    for v in views:
        v = np.ascontiguousarray(v)
        h, w = v.shape[:2]
        if w < 16 or h < 16:
            continue
        cfgv = H.synth_config(2, width=w, height=h)
        det, tpls, cam = make_detector(oa, cfgv, ["2x2-01"], 1)
        markers, counts = det.detect_host(v[None].copy())
        check_frame(det, 0, v, tpls, cam, markers, counts)


def test_find_squares_entry_matches_oracle_on_crops(oa):
    cfg = H.synth_config(3)
    det, tpls, cam = make_detector(oa, cfg, None, 1)
    frame, _ = H.synth_frame(cfg, 5)
    gray = np.ascontiguousarray(frame[..., 0])
    quads = H.oracle_find_squares(gray)
    for qd in quads[:10]:
        x0, y0 = max(qd[:, 0].min() - 5, 0), max(qd[:, 1].min() - 5, 0)
        x1, y1 = min(qd[:, 0].max() + 5, cfg.width), min(qd[:, 1].max() + 5, cfg.height)
        crop = np.ascontiguousarray(gray[y0:y1, x0:x1])
        ref = H.oracle_find_squares(crop)
        got, n = det.find_squares(crop)
        assert n == len(ref) and np.array_equal(got, ref)


def test_stateful_tracking_second_call(oa):
    cfg = H.synth_config(2)
    det, tpls, cam = make_detector(oa, cfg, ["2x2-01"], 2)
    frames = np.stack([H.synth_frame(cfg, f, ["2x2-01"])[0] for f in (3, 4)])
    markers, counts = det.detect_host(frames.copy())
    prev = [[markers[f, k] for k in range(counts[f])] for f in range(2)]
    m2, c2 = det.detect_host(frames.copy(), prev=prev)
    for f in range(2):
        ref1, _, _ = H.oracle_registration(frames[f], tpls, cam)
        check_frame(det, f, frames[f], tpls, cam, m2, c2, prev=ref1)


def test_stream_tracker_three_steps_two_streams(oa):
    """Two video streams, three time steps: lane s of every batch is stream s and gets stream s's markers of the
    previous step (opencvar.cpp:635-668) -- against the oracle run sequentially per stream."""
    from opencv_ar_amd.tracking import StreamTracker
    cfg = H.synth_config(2)
    det, tpls, cam = make_detector(oa, cfg, ["2x2-01"], 2)
    tracker = StreamTracker(det, 2)
    seq = {0: [3, 4, 4], 1: [7, 7, 8]}   # synthetic frame index per step (a repeated frame tracks onto itself)
    ref_prev = {0: None, 1: None}
    for t in range(3):
        frames = np.stack([H.synth_frame(cfg, seq[s][t], ["2x2-01"])[0] for s in (0, 1)])
        markers, counts = tracker.step(frames.copy())
        for s in (0, 1):
            check_frame(det, s, frames[s], tpls, cam, markers, counts, prev=ref_prev[s])
            ref_prev[s], _, _ = H.oracle_registration(frames[s], tpls, cam, prev=ref_prev[s])


def test_pathological_noise_frames_are_handled_exactly(oa):
    """Full-frame blocky noise: tens of thousands of plausible starts, thousands of tier-1 survivors and percolating
    clusters whose borders have more corner points than a wave slab holds (tier 3 then follows again into the pool).
    The reference takes any frame; so does this path, with the same result."""
    rng = np.random.default_rng(12)
    w, h, cell = 1920, 1080, 4
    cfg = H.synth_config(3, width=w, height=h)
    det, tpls, cam = make_detector(oa, cfg, None, 1)
    frame = np.ascontiguousarray(np.kron(rng.integers(0, 2, (h // cell, w // cell, 1), np.uint8) * 255, np.ones((cell, cell, 3), np.uint8)))
    markers, counts = det.detect_host(frame[None].copy())
    pool_ints = det.counters()[5]
    check_frame(det, 0, frame, tpls, cam, markers, counts)
    assert pool_ints > 0    # the point pool was used: some border exceeded a slab


def test_results_to_device_block_equals_collect(oa):
    """ocvar_hip_results_to_device (the block a rank hands to the RCCL gather) holds exactly what ocvar_hip_collect
    returns, and sharding.unpack decodes it."""
    import torch
    from opencv_ar_amd import sharding as S
    cfg = H.synth_config(2)
    n = 5
    det, tpls, cam = make_detector(oa, cfg, ["2x2-01"], n)
    frames = np.stack([H.synth_frame(cfg, f, ["2x2-01"])[0] for f in range(n)])
    d = torch.from_numpy(frames).cuda()
    block = torch.zeros(S.block_bytes(n), dtype=torch.uint8, device="cuda")
    det.enqueue_device(d.data_ptr(), cfg.width, cfg.height, n)
    det.results_to_device(block.data_ptr(), block.data_ptr() + n * S.MAX_MARKERS * S.MARKER_BYTES)
    markers, counts = det.collect()
    torch.cuda.synchronize()
    res = S.unpack([block], n, oa.MARKER_DTYPE)
    assert sorted(res) == list(range(n))
    for f in range(n):
        c, m = res[f]
        assert c == counts[f] and c >= 1
        assert m.tobytes() == markers[f, :c].tobytes()
    # the narrow block (ocvar_hip_results_to_device_ex, what bench.py gathers): the first K records of every frame, full counts
    for K in (1, 3):
        narrow = torch.zeros(S.block_bytes(n, K), dtype=torch.uint8, device="cuda")
        det.enqueue_device(d.data_ptr(), cfg.width, cfg.height, n)
        det.results_to_device(narrow.data_ptr(), narrow.data_ptr() + n * K * S.MARKER_BYTES, per_frame=K)
        det.collect()
        torch.cuda.synchronize()
        res = S.unpack([narrow], n, oa.MARKER_DTYPE, K)
        for f in range(n):
            c, m = res[f]
            assert c == counts[f] and len(m) == min(c, K)
            assert m.tobytes() == markers[f, :min(c, K)].tobytes()


def test_crop_pass_one_and_two_phases_give_the_same_results(oa):
    """The crop pass walks a crop's borders in one launch (batches of <= 8 frames) or in two with exact pruning behind the
    crop's best quad (follow.hip::follow_mid_kernel): both forms, forced on the same textured frames, equal the oracle --
    and with it each other -- candidate for candidate."""
    import torch
    cfg = H.synth_config(3, width=800, height=600, grid_x=3, grid_y=2, textured=1, corner_jitter_pct=6, occlude_pct=20)
    n = 10
    det, tpls, cam = make_detector(oa, cfg, None, n)
    frames = np.stack([H.synth_frame(cfg, 100 + f)[0] for f in range(n)])
    d = torch.from_numpy(frames).cuda()
    got = {}
    for phases in (1, 2):
        det.set_tuning(crop_phases=phases)
        markers, counts = det.detect_device(d.data_ptr(), cfg.width, cfg.height, n)
        for f in range(n):
            check_frame(det, f, frames[f], tpls, cam, markers, counts)
        got[phases] = (markers.tobytes(), counts.tobytes())
    assert got[1] == got[2]
    # the other launch parameters of ocvar_hip_set_tuning are result-invariant as well: a short tier-2 budget (everything long
    # goes to the wave tier), tiny grids, taller binarise chunks
    det.set_tuning(crop_phases=0, mid_steps=64, mid_blocks=3, long_blocks=2, short_blocks=5, min_units=1)
    markers, counts = det.detect_device(d.data_ptr(), cfg.width, cfg.height, n)
    assert (markers.tobytes(), counts.tobytes()) == got[1]
    assert "product" in oa.build_info() or "OCVAR_PROF" in oa.build_info()


def test_round_trip_properties_full_size(oa):
    """Size-independent properties at the headline size: determinism across batch positions, and every decoded
    4x4 marker's quad lies on a planted marker (corner within 12 px of the truth: the decoded quad is the inner border)."""
    import torch
    cfg = H.synth_config(3)
    n = 8
    det, tpls, cam = make_detector(oa, cfg, None, n)
    frame, truth = H.synth_frame(cfg, 11)
    frames = np.stack([frame] * n)
    d = torch.from_numpy(frames).cuda()
    markers, counts = det.detect_device(d.data_ptr(), cfg.width, cfg.height, n)
    assert (counts == counts[0]).all()
    for f in range(1, n):
        assert markers[f].tobytes() == markers[0].tobytes()
    cands = det.debug_candidates(n - 1)
    tcorners = np.concatenate([t["corner"] for t in truth])
    hits = 0
    for c in cands:
        if c.orient and c.templateId == 2:
            sq = np.array(c.square).reshape(4, 2)
            dist = np.sqrt(((sq[:, None, :] - tcorners[None]) ** 2).sum(-1)).min(1)
            assert dist.max() < 12.0
            hits += 1
    assert hits >= 3


def _sawtooth_frame(w, h, tooth, shapes):
    """White frame with dark rectangles whose edges are sawtoothed: borders with thousands of CHAIN_APPROX_SIMPLE points."""
    img = np.full((h, w), 220, np.uint8)
    for (x0, y0, x1, y1) in shapes:
        img[y0:y1, x0:x1] = 20
        for x in range(x0, x1, 2 * tooth):          # teeth along the top and bottom edges
            img[y0 - tooth:y0, x:x + tooth] = 20
            img[y1:y1 + tooth, x + tooth:x + 2 * tooth] = 20
        for y in range(y0, y1, 2 * tooth):          # and along the left and right edges
            img[y:y + tooth, x0 - tooth:x0] = 20
            img[y + tooth:y + 2 * tooth, x1:x1 + tooth] = 20
    return np.repeat(img[:, :, None], 3, axis=2)


def test_borders_with_thousands_of_points(oa):
    """Long, point-rich borders go through the follower's upper tiers (lane slabs, wave slabs); results must not
    depend on which tier handled a border."""
    w, h = 1280, 960
    cfg = H.synth_config(2, width=w, height=h)
    det, tpls, cam = make_detector(oa, cfg, ["2x2-01"], 2)
    frames = np.stack([_sawtooth_frame(w, h, 6, [(100, 100, 700, 500), (800, 150, 1200, 900), (150, 600, 650, 880)]),
                       _sawtooth_frame(w, h, 4, [(60, 60, 1220, 900)])])
    markers, counts = det.detect_host(frames.copy())
    for f in range(2):
        check_frame(det, f, frames[f], tpls, cam, markers, counts)
    # the square finder alone, on a crop-sized image cut out of the first frame
    crop = np.ascontiguousarray(frames[0][80:540, 80:740, 0])
    ref = H.oracle_find_squares(crop)
    got, n = det.find_squares(crop)
    assert n == len(ref) and np.array_equal(got, ref)


def test_capacity_overflow_is_loud_not_wrong(oa):
    """A border with more points than any follower tier can hold must either be handled exactly or make the call
    fail with OCVAR_E_CAPACITY -- never a silently different result."""
    w, h = 3840, 2160
    cfg = H.synth_config(2, width=w, height=h)
    det, tpls, cam = make_detector(oa, cfg, ["2x2-01"], 1)
    frame = _sawtooth_frame(w, h, 4, [(40, 40, 3800, 2120)])
    try:
        markers, counts = det.detect_host(frame[None].copy())
    except oa.OcvarError as e:
        assert "-4" in str(e) or "overflow" in str(e)
        return
    check_frame(det, 0, frame, tpls, cam, markers, counts)


def test_camera_with_lens_distortion(oa):
    """A calibrated camera (cvarReadCamera(file): non-zero distCoeffs) reaches the pose solver as in the reference
    (opencvar.cpp:261-272): glMatrix of every marker against the oracle's full-camera-model solve."""
    cfg = H.synth_config(2)
    names = ["2x2-01"]
    tpls = H.oracle_templates(names)
    cam = H.oracle_camera(cfg.width, cfg.height)
    for i, v in enumerate([-0.21, 0.09, 0.0015, -0.0008, -0.02]):
        cam.distCoeffs[i] = v
    det = oa.Detector(cfg.width, cfg.height, max_batch=3)
    det.set_templates([oa.Template.from_buffer_copy(bytes(t)) for t in tpls])
    det.set_camera(oa.Camera.from_buffer_copy(bytes(cam)))
    frames = np.stack([H.synth_frame(cfg, f, names)[0] for f in (0, 5, 9)])
    markers, counts = det.detect_host(frames.copy())
    pin = H.oracle_camera(cfg.width, cfg.height)
    moved = 0
    for f in range(3):
        ref_m, _ = check_frame(det, f, frames[f], tpls, cam, markers, counts)
        ref_pin, _, _ = H.oracle_registration(frames[f], tpls, pin)
        moved += sum(np.abs(np.array(a.glMatrix) - np.array(b.glMatrix)).max() > 1e-3 for a, b in zip(ref_m, ref_pin))
    assert moved >= 1   # the coefficients change the pose: the test would notice a solver that ignores them


@pytest.mark.gpu
def test_planes_of_every_frame_of_a_busy_batch(oa):
    """Grey plane and neighbour masks of ALL frames of a batch that keeps the memory system busy (64 full-HD frames): a store
    whose data registers were overwritten too early corrupted mask words only from the 16th frame of a batch on -- every
    small-batch test was green (the buffer stores' scalar offset field)."""
    cfg = H.synth_config(3)
    uniq, B = 4, 64
    base = np.stack([H.synth_frame(cfg, i, None)[0] for i in range(uniq)])
    det, tpls, cam = make_detector(oa, cfg, None, B)
    import torch
    d = torch.from_numpy(base).cuda().repeat(B // uniq, 1, 1, 1).contiguous()
    torch.cuda.synchronize()
    w, h = cfg.width, cfg.height
    refs = []
    for u in range(uniq):
        _, _, grey = H.oracle_registration(base[u], tpls, cam)
        g_ref = np.ascontiguousarray(grey[..., 0])
        refs.append((g_ref, H.neighbour_masks(H.oracle_binarise(g_ref))))
    for rep in range(2):
        markers, counts = det.detect_device(d.data_ptr(), w, h, B)
        assert all(counts[f] == counts[f % uniq] for f in range(B))
    for f in range(B):
        assert np.array_equal(det.debug_gray(f, w, h), refs[f % uniq][0]), f
        assert np.array_equal(det.debug_masks(f, w, h), refs[f % uniq][1]), f
