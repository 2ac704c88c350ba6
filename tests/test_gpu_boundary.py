"""GPU tests of the boundaries around the hot path: the reference's debug square API through lib/libopencv-ar.so
(SURVEY 8(f)4), the per-call latency sample, the RCCL result gather through torch.distributed and through
the one-process C helper of include/ocvar_multi.h."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

import helpers as H
from helpers import P

pytestmark = pytest.mark.gpu

LIB = os.path.join(H.PKG, "lib", "libopencv-ar.so.1.0.0")
BIN = os.path.join(H.PKG, "bin")
TEMPLATES = os.path.join(H.ROOT, "assets", "templates")


class IplImage(C.Structure):   # OpenCV's layout (include/shim/opencv/cv.h): 144 bytes on LP64
    _fields_ = [("nSize", C.c_int), ("ID", C.c_int), ("nChannels", C.c_int), ("alphaChannel", C.c_int), ("depth", C.c_int),
                ("colorModel", C.c_char * 4), ("channelSeq", C.c_char * 4), ("dataOrder", C.c_int), ("origin", C.c_int),
                ("align", C.c_int), ("width", C.c_int), ("height", C.c_int), ("roi", C.c_void_p), ("maskROI", C.c_void_p),
                ("imageId", C.c_void_p), ("tileInfo", C.c_void_p), ("imageSize", C.c_int), ("imageData", C.c_void_p),
                ("widthStep", C.c_int), ("BorderMode", C.c_int * 4), ("BorderConst", C.c_int * 4), ("imageDataOrigin", C.c_void_p)]


class CvSeqBlock(C.Structure):
    pass


CvSeqBlock._fields_ = [("prev", C.POINTER(CvSeqBlock)), ("next", C.POINTER(CvSeqBlock)), ("start_index", C.c_int),
                       ("count", C.c_int), ("data", C.c_void_p)]


class CvSeq(C.Structure):   # OpenCV 2.x/3.x CvSeq: total at offset 40
    _fields_ = [("flags", C.c_int), ("header_size", C.c_int), ("h_prev", C.c_void_p), ("h_next", C.c_void_p), ("v_prev", C.c_void_p),
                ("v_next", C.c_void_p), ("total", C.c_int), ("elem_size", C.c_int), ("block_max", C.c_void_p), ("ptr", C.c_void_p),
                ("delta_elems", C.c_int), ("storage", C.c_void_p), ("free_blocks", C.c_void_p), ("first", C.POINTER(CvSeqBlock))]


class StdVector(C.Structure):   # libstdc++ std::vector: begin, end, end of storage
    _fields_ = [("begin", C.c_void_p), ("end", C.c_void_p), ("cap", C.c_void_p)]


def ipl(bgr):
    h, w = bgr.shape[:2]
    img = IplImage()
    img.nSize, img.nChannels, img.depth, img.width, img.height = C.sizeof(IplImage), 3, 8, w, h
    img.widthStep, img.imageSize = bgr.strides[0], bgr.strides[0] * h
    img.imageData = img.imageDataOrigin = bgr.ctypes.data
    return img


@pytest.fixture(scope="module")
def host():
    import torch
    assert torch.cuda.is_available(), "gpu tests need a device"
    import opencv_ar_amd as oa
    oa.hip_lib()
    assert C.sizeof(IplImage) == 144 and CvSeq.total.offset == 40 and C.sizeof(CvSeq) == 96
    lib = C.CDLL(LIB)
    lib.cvarFindSquares.restype = C.POINTER(CvSeq)
    lib.cvarFindSquares.argtypes = [C.c_void_p, C.c_void_p]
    return lib


def seq_points(seq):
    """walks the element blocks the way cvGetSeqElem does"""
    s = seq.contents
    assert s.elem_size == 8 and s.header_size == 96
    pts, blk = [], s.first
    while len(pts) < s.total:
        b = blk.contents
        pts.extend(np.ctypeslib.as_array(C.cast(b.data, C.POINTER(C.c_int)), shape=(b.count, 2)).copy().tolist())
        blk = b.next
    return np.array(pts[:s.total], np.int32).reshape(-1, 4, 2)


def test_debug_square_api_through_the_reference_boundary(host):
    """cvarFindSquares -> cvarGetAllSquares / cvarGetSquare -> cvarCompareSquare -> cvarDrawSquares
    (/root/reference/src/opencvar.cpp:156-223, 327-430, 564-590) on a greyed synthetic frame, against the oracle."""
    cfg = H.synth_config(2)
    frame, _ = H.synth_frame(cfg, 3, ["2x2-01"])
    tpls, cam = H.oracle_templates(["2x2-01"]), H.oracle_camera(cfg.width, cfg.height)
    _, _, grey = H.oracle_registration(frame, tpls, cam)   # the image the reference hands to cvarFindSquares (624-632)
    ref = H.oracle_find_squares(np.ascontiguousarray(grey[..., 0]))
    assert len(ref) >= 4
    img_arr = np.ascontiguousarray(grey.copy())
    img = ipl(img_arr)
    seq = host.cvarFindSquares(C.byref(img), None)
    assert seq.contents.total == 4 * len(ref)                # what a caller reads at OpenCV's offset of `total`
    assert np.array_equal(seq_points(seq), ref)              # sequence order included
    # cvarGetAllSquares appends CvPoint2D32f to the caller's vector (564-590)
    vec = StdVector()
    assert host.cvarGetAllSquares(seq, C.byref(vec)) == len(ref)
    n_pts = (vec.end - vec.begin) // 8
    got = np.ctypeslib.as_array(C.cast(vec.begin, C.POINTER(C.c_float)), shape=(n_pts, 2)).reshape(-1, 4, 2)
    assert np.array_equal(got, ref.astype(np.float32))
    # cvarGetSquare keeps the LAST square of the sequence (401-430)
    last = np.zeros((4, 2), np.float32)
    assert host.cvarGetSquare(seq, P(last)) == len(ref)
    assert np.array_equal(last, ref[-1].astype(np.float32))
    # cvarCompareSquare counts corner pairs closer than 10 px over every square of the image (327-367)
    probe = ref[0].astype(np.float32)
    want = sum(int(np.hypot(*(p - q.astype(np.float64))) < 10) for sq in ref for p in probe.astype(np.float64) for q in sq)
    assert host.cvarCompareSquare(C.byref(img), P(probe)) == want
    # cvarDrawSquares returns the number of squares and marks their corners green (369-399; aliased lines here)
    assert host.cvarDrawSquares(C.byref(img), seq) == len(ref)
    for sq in ref:
        for x, y in sq:
            assert img_arr[y, x].tolist() == [0, 255, 0]


def test_find_squares_sequences_live_as_long_as_their_storage(host):
    """In the reference the returned CvSeq lives in the caller's CvMemStorage (opencvar.cpp:168).  Here a storage pointer is a
    key: hundreds of later calls with OTHER storages (the old 256-sequence ring would have freed it) leave a sequence intact;
    cvarReleaseSquares(key) -- this library's stand-in for cvClearMemStorage -- ends its life."""
    cfg = H.synth_config(2)
    frame, _ = H.synth_frame(cfg, 3, ["2x2-01"])
    grey = np.ascontiguousarray(np.repeat(frame[..., :1], 3, axis=2))
    ref = H.oracle_find_squares(np.ascontiguousarray(grey[..., 0]))
    img = ipl(grey)
    key_a, key_b = C.c_void_p(0x1000), C.c_void_p(0x2000)   # opaque keys: never dereferenced by the library
    seq = host.cvarFindSquares(C.byref(img), key_a)
    assert seq.contents.storage == key_a.value
    small = ipl(np.ascontiguousarray(grey[:64, :64]))
    for _ in range(300):
        host.cvarFindSquares(C.byref(small), key_b)
    assert np.array_equal(seq_points(seq), ref)
    host.cvarReleaseSquares.argtypes = [C.c_void_p]
    host.cvarReleaseSquares.restype = None
    host.cvarReleaseSquares(key_b)
    assert np.array_equal(seq_points(seq), ref)
    host.cvarReleaseSquares(key_a)


def test_find_squares_on_a_colour_image_uses_bgr2gray_weights(host):
    cfg = H.synth_config(2)
    frame, _ = H.synth_frame(cfg, 1, ["2x2-01"])
    tinted = frame.copy()
    tinted[..., 2] = (tinted[..., 2].astype(np.int32) * 3 // 4).astype(np.uint8)   # R != G = B
    g = ((tinted[..., 0].astype(np.uint32) * 1868 + tinted[..., 1].astype(np.uint32) * 9617 + tinted[..., 2].astype(np.uint32) * 4899 + 8192) >> 14).astype(np.uint8)
    ref = H.oracle_find_squares(np.ascontiguousarray(g))
    img = ipl(tinted)
    seq = host.cvarFindSquares(C.byref(img), None)
    assert np.array_equal(seq_points(seq), ref)


def test_per_call_latency_sample_runs():
    out = subprocess.run([os.path.join(BIN, "artest_latency"), TEMPLATES, "2", "5"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    assert rec["width"] == 640 and rec["markers_out"] >= 1 and 0 < rec["min_ms"] <= rec["median_ms"] <= rec["p90_ms"]


def test_multi_gpu_helper_gathers_over_rccl():
    """include/ocvar_multi.h on the devices of this box (one here): ncclCommInitAll + one ncclGather of the result block
    per batch; the demo compares every CvarMarker record with the single-GPU entry point on the same frames."""
    out = subprocess.run([os.path.join(BIN, "multi_gpu_demo"), TEMPLATES, "0", "2", "6", "2"], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert out.returncode == 0, out.stdout + out.stderr
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    assert rec["devices"] >= 1 and rec["frames"] == 6 * rec["devices"] and rec["mismatches_vs_single_gpu"] == 0
    assert rec["markers_total"] >= rec["frames"]
    # stateful: streams sharded over the devices, their markers kept on their device between three steps
    # (ocvar_multi_track_host) == per-step single-GPU calls with the markers handed back by the host
    assert rec["tracked_steps"] == 3 and rec["tracked_mismatches"] == 0


def test_rccl_gather_of_result_blocks_world1():
    """The bench's N > 1 exchange on real hardware: a torch.distributed process group with backend "nccl" (= RCCL), the
    detector's device-resident result block pushed through sharding.gather_blocks and decoded with unpack."""
    import torch
    import torch.distributed as dist
    import opencv_ar_amd as oa
    from opencv_ar_amd import sharding as S
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", "29533"
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        cfg = H.synth_config(2)
        n = 5
        frames = np.stack([H.synth_frame(cfg, f, ["2x2-01"])[0] for f in range(n)])
        tpls, cam = H.oracle_templates(["2x2-01"]), H.oracle_camera(cfg.width, cfg.height)
        det = oa.Detector(cfg.width, cfg.height, max_batch=n)
        det.set_templates([oa.Template.from_buffer_copy(bytes(t)) for t in tpls])
        det.set_camera(oa.Camera.from_buffer_copy(bytes(cam)))
        d = torch.from_numpy(frames).cuda()
        block = torch.zeros(S.block_bytes(n), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        det.enqueue_device(d.data_ptr(), cfg.width, cfg.height, n)
        det.results_to_device(block.data_ptr(), block.data_ptr() + n * S.MAX_MARKERS * S.MARKER_BYTES)
        markers, counts = det.collect()
        blocks = S.gather_blocks(block, 0, 1, dist)       # dist.gather over RCCL, no short cut
        torch.cuda.synchronize()
        assert len(blocks) == 1 and blocks[0].data_ptr() != block.data_ptr()
        res = S.unpack(blocks, n, oa.MARKER_DTYPE)
        assert sorted(res) == list(range(n))
        for f in range(n):
            c, m = res[f]
            assert c == counts[f] >= 1
            assert m.tobytes() == markers[f, :c].tobytes()
    finally:
        dist.destroy_process_group()


def _grid_of_squares(w, h, pitch=44, side=30):
    g = np.full((h, w), 200, np.uint8)
    for y in range(20, h - side - 20, pitch):
        for x in range(20, w - side - 20, pitch):
            g[y:y + side, x:x + side] = 40
    return g


def test_more_squares_than_the_default_list_holds(host):
    """The reference's square list is unbounded (opencvar.cpp:187-214).  An image with more than OCVAR_MAX_QUADS (256)
    squares is run again on a context with a longer list by the host mirror: same sequence as the oracle, in order."""
    g = _grid_of_squares(1280, 960)
    ref = H.oracle_find_squares(g)
    assert len(ref) > 256
    img_arr = np.ascontiguousarray(np.repeat(g[:, :, None], 3, axis=2))
    img = ipl(img_arr)
    seq = host.cvarFindSquares(C.byref(img), None)
    assert seq.contents.total == 4 * len(ref)
    assert np.array_equal(seq_points(seq), ref)


def test_registration_of_a_frame_with_hundreds_of_squares():
    """cvarArMultRegistration through libopencv-ar.so on a frame whose frame pass yields > 256 squares (and > 256 crops):
    the C ABI reports OCVAR_E_CAPACITY on a default context; one created with ocvar_hip_create_ex equals the oracle."""
    import torch
    import opencv_ar_amd as oa
    g = _grid_of_squares(1280, 960)
    frame = np.ascontiguousarray(np.repeat(g[:, :, None], 3, axis=2))
    tpls, cam = H.oracle_templates(["2x2-01"]), H.oracle_camera(1280, 960)
    ref_m, ref_c, _ = H.oracle_registration(frame, tpls, cam)
    small = oa.Detector(1280, 960, max_batch=1)
    small.set_templates([oa.Template.from_buffer_copy(bytes(t)) for t in tpls])
    small.set_camera(oa.Camera.from_buffer_copy(bytes(cam)))
    with pytest.raises(oa.OcvarError):
        small.detect_host(frame[None].copy())
    assert oa.hip_lib().ocvar_hip_capacity_flags(small._ctx) & 4
    lib = oa.hip_lib()
    lib.ocvar_hip_create_ex.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    big = oa.Detector.__new__(oa.Detector)
    big._lib, big._ctx, big.max_batch, big.n_templates = lib, C.c_void_p(), 1, 0
    assert lib.ocvar_hip_create_ex(C.byref(big._ctx), 0, 1280, 960, 1, 1024) == 0
    big.set_templates([oa.Template.from_buffer_copy(bytes(t)) for t in tpls])
    big.set_camera(oa.Camera.from_buffer_copy(bytes(cam)))
    markers, counts = big.detect_host(frame[None].copy())
    assert counts[0] == len(ref_m)
    for k, r in enumerate(ref_m):
        assert markers[0, k]["templateId"] == r.templateId and markers[0, k]["markerId"] == r.markerId
        gm = np.array(r.glMatrix)
        assert np.abs(markers[0, k]["glMatrix"] - gm).max() <= 1e-4 * max(1.0, np.abs(gm).max())


def test_small_frames_in_heap_blocks_never_reach_the_device_directly():
    """Small batches of small frames live in malloc'ed heap blocks that share their pages with other data; a randomised
    sweep once faulted the GPU on such a host address while the block was page-locked in place (round 2).  The library no
    longer page-locks caller memory at all: frames of every size and the previous-marker arrays travel through the
    context's own page-locked buffers.  Many odd-sized stateful calls, results against the oracle."""
    import opencv_ar_amd as oa
    from opencv_ar_amd.tracking import StreamTracker
    rng = np.random.default_rng(9)
    cfg = H.synth_config(3, width=96, height=64, grid_x=1, grid_y=1, side_min=40, side_max=50)
    names = ["2x2-01"]
    tpls, cam = H.oracle_templates(names), H.oracle_camera(96, 64)
    keep = []
    for trial in range(24):
        nb = int(rng.integers(2, 5))
        keep.append(np.zeros(int(rng.integers(1, 40000)), np.uint8))   # shifts where the next arrays land in the heap
        det = oa.Detector(96, 64, max_batch=nb)
        det.set_templates([oa.Template.from_buffer_copy(bytes(t)) for t in tpls])
        det.set_camera(oa.Camera.from_buffer_copy(bytes(cam)))
        tracker = StreamTracker(det, nb)
        frames = np.ascontiguousarray(np.stack([H.synth_frame(cfg, int(rng.integers(0, 1000)), names)[0] for _ in range(nb)]))
        prev = [None] * nb
        for step in range(2):
            markers, counts = tracker.step(frames.copy())
            for f in range(nb):
                ref_m, _, _ = H.oracle_registration(frames[f], tpls, cam, prev=prev[f])
                assert counts[f] == len(ref_m)
                for k, r in enumerate(ref_m):
                    assert markers[f, k]["markerId"] == r.markerId and markers[f, k]["templateId"] == r.templateId
                prev[f] = ref_m
        det.close()


@pytest.mark.gpu
def test_contexts_sharing_a_gate_give_the_same_results():
    """Two contexts with batches in flight on their own streams, at most one binarise kernel at a time between them
    (ocvar_hip_gate_create / ocvar_hip_set_gate): same markers as the same batches without the gate, which the parity tests
    check against the oracle."""
    import torch
    import opencv_ar_amd as oa
    cfg = H.synth_config(2)
    names = ["2x2-01"]
    n = 12
    tpls = oa.load_templates([os.path.join(oa.TEMPLATE_DIR, x + ".png") for x in names])
    cam = oa.default_camera(cfg.width, cfg.height)
    frames = [np.stack([H.synth_frame(cfg, 40 * k + f, names)[0] for f in range(n)]) for k in range(2)]
    d = [torch.from_numpy(f).cuda() for f in frames]
    streams = [torch.cuda.Stream() for _ in range(2)]
    torch.cuda.synchronize()

    def run(gate):
        dets = []
        for k in range(2):
            det = oa.Detector(cfg.width, cfg.height, max_batch=n)
            det.set_templates(tpls)
            det.set_camera(cam)
            if gate is not None:
                det.set_gate(gate)
            dets.append(det)
        out = []
        for rep in range(3):   # several batches per context: the order of the gate's events is exercised
            for k in range(2):
                dets[k].enqueue_device(d[k].data_ptr(), cfg.width, cfg.height, n, stream=streams[k].cuda_stream)
            out = [dets[k].collect(8) for k in range(2)]
        return [(m.tobytes(), c.tobytes()) for m, c in out]

    plain = run(None)
    gated = run(oa.Gate(1))
    assert plain == gated
    assert all(np.frombuffer(c, np.int32).min() >= 1 for _, c in plain)


def test_pipe_of_contexts_equals_one_context():
    """ocvar_hip_pipe_*: the frames of one device array cut into chunks that go round three gated contexts (ragged first
    chunks, a last partial chunk) -- same markers and counts, frame by frame, as one context detecting them in one batch."""
    import torch
    import opencv_ar_amd as oa
    cfg = H.synth_config(2)
    names = ["2x2-01"]
    n = 29
    tpls = oa.load_templates([os.path.join(oa.TEMPLATE_DIR, x + ".png") for x in names])
    cam = oa.default_camera(cfg.width, cfg.height)
    frames = np.stack([H.synth_frame(cfg, 7 * f, names)[0] for f in range(n)])
    d = torch.from_numpy(frames).cuda()
    torch.cuda.synchronize()
    det = oa.Detector(cfg.width, cfg.height, max_batch=n)
    det.set_templates(tpls)
    det.set_camera(cam)
    m1, c1 = det.detect_device(d.data_ptr(), cfg.width, cfg.height, n, max_per_frame=8)
    pipe = oa.Pipe(cfg.width, cfg.height, chunk_frames=4, n_contexts=3, gate_width=1)
    pipe.set_templates(tpls)
    pipe.set_camera(cam)
    for _ in range(2):
        m2, c2 = pipe.detect_device(d.data_ptr(), cfg.width, cfg.height, n, max_per_frame=8)
        assert c1.tobytes() == c2.tobytes() and c1.min() >= 1
        assert m1.tobytes() == m2.tobytes()


def test_pipe_streaming_submit_collect():
    """ocvar_hip_pipe_submit / _collect: chunks handed to the contexts as the caller produces them, collected oldest first with
    their tags; OCVAR_E_BUSY when every context has one in flight; results equal one context's on the same frames."""
    import torch
    import opencv_ar_amd as oa
    cfg = H.synth_config(2)
    names = ["2x2-01"]
    n, chunk = 22, 4
    tpls = oa.load_templates([os.path.join(oa.TEMPLATE_DIR, x + ".png") for x in names])
    cam = oa.default_camera(cfg.width, cfg.height)
    frames = np.stack([H.synth_frame(cfg, 5 * f, names)[0] for f in range(n)])
    d = torch.from_numpy(frames).cuda()
    torch.cuda.synchronize()
    det = oa.Detector(cfg.width, cfg.height, max_batch=n)
    det.set_templates(tpls)
    det.set_camera(cam)
    m1, c1 = det.detect_device(d.data_ptr(), cfg.width, cfg.height, n, max_per_frame=8)
    pipe = oa.Pipe(cfg.width, cfg.height, chunk_frames=chunk, n_contexts=3, gate_width=1)
    pipe.set_templates(tpls)
    pipe.set_camera(cam)
    pipe.set_result_limit(8)
    assert pipe.collect(chunk, 8) is None and pipe.in_flight() == 0
    fb = cfg.width * cfg.height * 3
    starts = list(range(0, n, chunk))
    sub = 0
    got = {}
    while len(got) < len(starts):
        while sub < len(starts):
            cnt = min(chunk, n - starts[sub])
            if not pipe.submit(d.data_ptr() + starts[sub] * fb, cfg.width, cfg.height, cnt, tag=1000 + sub):
                assert pipe.in_flight() == 3      # busy: every context has a chunk
                break
            sub += 1
        tag, m, c = pipe.collect(chunk, 8)
        assert tag == 1000 + len(got)             # oldest first
        got[tag - 1000] = (m, c)
    assert pipe.in_flight() == 0
    for k, s0 in enumerate(starts):
        m, c = got[k]
        assert c.tobytes() == c1[s0:s0 + len(c)].tobytes() and m.tobytes() == m1[s0:s0 + len(c)].tobytes()


def test_pipe_tracks_streams_with_state_on_the_device():
    """ocvar_hip_pipe_track_device: 11 video streams, chunks of 4 going round 2 contexts, 3 time steps; every stream's markers
    of the previous step stay in device memory between the calls (ocvar_hip_enqueue_tracked).  Against the oracle run
    sequentially per stream (opencvar.cpp:635-668), and against one context that is handed the previous markers from the
    host, record for record."""
    import torch
    import opencv_ar_amd as oa
    cfg = H.synth_config(2)
    names = ["2x2-01"]
    ns = 11
    tpls_o, cam_o = H.oracle_templates(names), H.oracle_camera(cfg.width, cfg.height)
    tpls = [oa.Template.from_buffer_copy(bytes(t)) for t in tpls_o]
    cam = oa.Camera.from_buffer_copy(bytes(cam_o))
    pipe = oa.Pipe(cfg.width, cfg.height, chunk_frames=4, n_contexts=2, gate_width=1)
    pipe.set_templates(tpls)
    pipe.set_camera(cam)
    det = oa.Detector(cfg.width, cfg.height, max_batch=ns)
    det.set_templates(tpls)
    det.set_camera(cam)
    seq = [[3 + s, 4 + s, 4 + s] for s in range(ns)]   # synthetic frame index of stream s at step t (a repeated frame tracks onto itself)
    ref_prev = [None] * ns
    host_prev = None
    for t in range(3):
        frames = np.stack([H.synth_frame(cfg, seq[s][t], names)[0] for s in range(ns)])
        d = torch.from_numpy(frames).cuda()
        torch.cuda.synchronize()
        markers, counts = pipe.track_device(d.data_ptr(), cfg.width, cfg.height, ns, reset=(t == 0))
        m1, c1 = det.detect_device(d.data_ptr(), cfg.width, cfg.height, ns, prev=host_prev)
        assert counts.tobytes() == c1.tobytes()
        for s in range(ns):
            assert markers[s, :counts[s]].tobytes() == m1[s, :c1[s]].tobytes()
            ref_m, _, _ = H.oracle_registration(frames[s], tpls_o, cam_o, prev=ref_prev[s])
            assert counts[s] == len(ref_m)
            for k, r in enumerate(ref_m):
                assert markers[s, k]["markerId"] == r.markerId and markers[s, k]["templateId"] == r.templateId and markers[s, k]["score"] == r.score
                assert np.abs(markers[s, k]["square"] - np.array(r.square)).max() <= 0.5
                g = np.array(r.glMatrix)
                assert np.abs(markers[s, k]["glMatrix"] - g).max() <= 1e-4 * max(1.0, np.abs(g).max())
            ref_prev[s] = ref_m
        host_prev = [[m1[s, k] for k in range(c1[s])] for s in range(ns)]
    assert max(len(r) for r in ref_prev) >= 2   # tracked + new markers were carried
    # a reset forgets the streams: the next step equals a stateless call
    markers, counts = pipe.track_device(d.data_ptr(), cfg.width, cfg.height, ns, reset=True)
    m0, c0 = det.detect_device(d.data_ptr(), cfg.width, cfg.height, ns)
    assert counts.tobytes() == c0.tobytes() and all(markers[s, :c0[s]].tobytes() == m0[s, :c0[s]].tobytes() for s in range(ns))


def test_host_entry_uses_a_buffer_the_caller_page_locked_in_place():
    """A caller that wants PCIe copies straight out of its own buffer page-locks it itself (hipHostMalloc: torch's pinned
    memory); ocvar_hip_detect_host recognises such a buffer and skips its staging copies.  Same results and in-place grey as the
    staged path on ordinary memory."""
    import torch
    import opencv_ar_amd as oa
    cfg = H.synth_config(2)
    names = ["2x2-01"]
    n = 7
    tpls = oa.load_templates([os.path.join(oa.TEMPLATE_DIR, x + ".png") for x in names])
    cam = oa.default_camera(cfg.width, cfg.height)
    frames = np.stack([H.synth_frame(cfg, 50 + f, names)[0] for f in range(n)])
    frames[..., 2] = (frames[..., 2].astype(np.int32) * 7 // 8).astype(np.uint8)   # R != G = B: greying changes the bytes
    det = oa.Detector(cfg.width, cfg.height, max_batch=3)
    det.set_templates(tpls)
    det.set_camera(cam)
    plain = frames.copy()
    m1, c1 = det.detect_host(plain, grey_in_place=True)
    pinned_t = torch.empty(frames.shape, dtype=torch.uint8, pin_memory=True)
    pinned = pinned_t.numpy()
    pinned[:] = frames
    m2, c2 = det.detect_host(pinned, grey_in_place=True)
    assert c1.tobytes() == c2.tobytes() and m1.tobytes() == m2.tobytes() and c1.min() >= 1
    assert np.array_equal(plain, pinned) and not np.array_equal(plain, frames)   # both greyed in place


def test_stage_stamps_put_two_contexts_on_one_clock():
    """ocvar_hip_stage_stamps: the 13 stage boundaries of a batch in ms after a reference event of the caller's -- increasing
    within a batch, consistent with ocvar_hip_stage_ms, and comparable between contexts (bench.py derives from them how long a
    kernel was on the GPU while several contexts ran)."""
    import torch
    import opencv_ar_amd as oa
    cfg = H.synth_config(2)
    names = ["2x2-01"]
    n = 8
    tpls = oa.load_templates([os.path.join(oa.TEMPLATE_DIR, x + ".png") for x in names])
    cam = oa.default_camera(cfg.width, cfg.height)
    frames = np.stack([H.synth_frame(cfg, f, names)[0] for f in range(n)])
    d = torch.from_numpy(frames).cuda()
    dets = []
    for _ in range(2):
        det = oa.Detector(cfg.width, cfg.height, max_batch=n)
        det.set_templates(tpls)
        det.set_camera(cam)
        dets.append(det)
    ref = torch.cuda.Event(enable_timing=True)
    ref.record()
    ref.synchronize()
    for det in dets:
        det.enqueue_device(d.data_ptr(), cfg.width, cfg.height, n)
    for det in dets:
        det.collect(8)
    stamps = [det.stage_stamps(ref.cuda_event) for det in dets]
    for det, st in zip(dets, stamps):
        assert len(st) == 13 and st[0] > 0 and (np.diff(st) >= 0).all()
        ms = det.stage_ms()
        assert np.allclose(np.diff(st)[:11], ms[:11], atol=0.02) and abs((st[12] - st[0]) - ms[11]) < 0.02
    assert abs(stamps[0][0] - stamps[1][0]) < 1000.0   # the same clock: both batches started within a second of the reference


def test_ready_and_result_limit():
    """ocvar_hip_ready turns 1 once the batch is done (collect then returns at once); a result limit of 2 records per frame
    brings the first two markers of every frame to the host and leaves the counts alone."""
    import time
    import torch
    import opencv_ar_amd as oa
    cfg = H.synth_config(3, width=800, height=600, grid_x=3, grid_y=2)
    n = 6
    tpls = oa.load_templates([os.path.join(oa.TEMPLATE_DIR, x + ".png") for x in H.TEMPLATE_ORDER])
    cam = oa.default_camera(cfg.width, cfg.height)
    frames = np.stack([H.synth_frame(cfg, 300 + f)[0] for f in range(n)])
    d = torch.from_numpy(frames).cuda()
    torch.cuda.synchronize()
    det = oa.Detector(cfg.width, cfg.height, max_batch=n)
    det.set_templates(tpls)
    det.set_camera(cam)
    full_m, full_c = det.detect_device(d.data_ptr(), cfg.width, cfg.height, n)
    assert full_c.max() >= 2
    det.set_result_limit(2)
    det.enqueue_device(d.data_ptr(), cfg.width, cfg.height, n)
    t0 = time.time()
    while not det.ready():
        assert time.time() - t0 < 30
        time.sleep(1e-3)
    m, c = det.collect(8)
    assert c.tobytes() == full_c.tobytes()
    for f in range(n):
        k = min(int(c[f]), 2)
        assert m[f, :k].tobytes() == full_m[f, :k].tobytes()
        assert not m[f, k:].tobytes().strip(b"\0")   # nothing beyond the limit reaches the caller
