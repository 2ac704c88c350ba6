"""MI355X-native AR-marker detection path of youtalk/opencv-ar -- Python binding of the C ABI.

The product is `lib/libocvar_hip.so` (hand-written HIP kernels for gfx950 behind `include/ocvar_hip.h`) and
`lib/libopencv-ar.so` (the host C++ mirror of the reference's `include/opencvar` API).  This module only binds
the C ABI with ctypes so that tests, `bench.py` and Python callers can drive it; torch is used by callers for
device memory and `torch.distributed`, never for compute.  There is no CPU fallback: importing works without a
GPU (so the symbol table can be checked), creating a `Detector` does not.
"""
import ctypes as C
import os

import numpy as np

# Several contexts in flight need one hardware queue per stream ("Hardware queues"); the ROCm runtime reads this when
# it initialises, so it only helps when this module is imported before anything touched the GPU -- callers that care set it
# themselves, first thing (bench.py does).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")
HIP_LIB = os.environ.get("OCVAR_HIP_LIB") or os.path.join(LIB_DIR, "libocvar_hip.so")   # (override: instrumented builds of tools/prof_tier2.py)
SYNTH_LIB = os.path.join(LIB_DIR, "libocvar_synth.so")
HOST_LIB = os.path.join(LIB_DIR, "libopencv-ar.so.1.0.0")

MAX_TEMPLATES, MAX_QUADS, MAX_MARKERS = 16, 256, 64

# every symbol include/ocvar_hip.h declares
HIP_SYMBOLS = [
    "ocvar_hip_create", "ocvar_hip_create_ex", "ocvar_hip_capacity_flags", "ocvar_hip_gate_create", "ocvar_hip_gate_destroy", "ocvar_hip_set_gate", "ocvar_hip_ready", "ocvar_hip_set_result_limit",
    "ocvar_hip_pipe_create", "ocvar_hip_pipe_destroy", "ocvar_hip_pipe_last_error", "ocvar_hip_pipe_set_templates", "ocvar_hip_pipe_set_camera",
    "ocvar_hip_pipe_detect_device", "ocvar_hip_pipe_track_device", "ocvar_hip_pipe_submit", "ocvar_hip_pipe_collect", "ocvar_hip_pipe_in_flight", "ocvar_hip_pipe_set_result_limit", "ocvar_hip_enqueue_tracked", "ocvar_hip_build_info", "ocvar_hip_set_tuning", "ocvar_hip_destroy", "ocvar_hip_last_error", "ocvar_hip_set_templates", "ocvar_hip_set_camera",
    "ocvar_hip_detect_device", "ocvar_hip_enqueue", "ocvar_hip_collect", "ocvar_hip_detect_host", "ocvar_hip_find_squares",
    "ocvar_hip_debug_gray", "ocvar_hip_debug_binary", "ocvar_hip_debug_masks", "ocvar_hip_debug_frame_quads", "ocvar_hip_debug_candidates",
    "ocvar_hip_stage_ms", "ocvar_hip_stream", "ocvar_hip_stage_stamps", "ocvar_hip_counters", "ocvar_hip_results_to_device", "ocvar_hip_results_to_device_ex", "ocvar_hip_debug_calibrate",
]
STAGE_NAMES = ["binarise_frames", "follow1_frames", "follow2_frames", "follow3_frames", "order_crops", "binarise_crops",
               "follow1_crops", "follow2_crops", "follow3_crops", "decode", "dedupe_pose", "batch_total"]


class Camera(C.Structure):  # CvarCamera, reference include/opencvar/opencvar.h:54-60
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("cameraMatrix", C.c_double * 9),
                ("distCoeffs", C.c_double * 5), ("glProjection", C.c_double * 16)]


class Template(C.Structure):  # CvarTemplate, opencvar.h:65-70
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("scale", C.c_double), ("code", C.c_longlong * 4)]


class Marker(C.Structure):  # CvarMarker, opencvar.h:75-82
    _fields_ = [("glMatrix", C.c_double * 16), ("templateId", C.c_int), ("markerId", C.c_int),
                ("score", C.c_double), ("square", C.c_float * 8), ("aspectRatio", C.c_double)]


class Candidate(C.Structure):
    _fields_ = [("markerId", C.c_int), ("templateId", C.c_int), ("orient", C.c_int), ("valid", C.c_int),
                ("bit", C.c_longlong), ("square", C.c_float * 8), ("patPoint", C.c_float * 8)]


MARKER_DTYPE = np.dtype([("glMatrix", "<f8", (16,)), ("templateId", "<i4"), ("markerId", "<i4"), ("score", "<f8"),
                         ("square", "<f4", (8,)), ("aspectRatio", "<f8")], align=True)
assert C.sizeof(Camera) == 248 and C.sizeof(Template) == 48 and C.sizeof(Marker) == 184 and MARKER_DTYPE.itemsize == 184


class OcvarError(RuntimeError):
    pass


_hip = None


def hip_lib():
    """Loads the HIP library; raises if it has not been built (no silent fallback)."""
    global _hip
    if _hip is None:
        if not os.path.exists(HIP_LIB):
            raise OcvarError(f"{HIP_LIB} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                             "(there is no CPU fallback for the detection path)")
        try:
            # torch ships its own HIP runtime; load it first so this library binds to the same one instead of
            # bringing a second runtime into the process (two runtimes cannot both own the device)
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(HIP_LIB)
        vp, i, sz = C.c_void_p, C.c_int, C.c_size_t
        lib.ocvar_hip_create.argtypes = [C.POINTER(vp), i, i, i, i]
        lib.ocvar_hip_destroy.argtypes = [vp]
        lib.ocvar_hip_gate_create.argtypes = [C.POINTER(vp), i, i]
        lib.ocvar_hip_gate_destroy.argtypes = [vp]
        lib.ocvar_hip_gate_destroy.restype = None
        lib.ocvar_hip_set_gate.argtypes = [vp, vp]
        lib.ocvar_hip_ready.argtypes = [vp]
        lib.ocvar_hip_set_result_limit.argtypes = [vp, i]
        lib.ocvar_hip_pipe_create.argtypes = [C.POINTER(vp), i, i, i, i, i, i]
        lib.ocvar_hip_pipe_destroy.argtypes = [vp]
        lib.ocvar_hip_pipe_destroy.restype = None
        lib.ocvar_hip_pipe_last_error.argtypes = [vp]
        lib.ocvar_hip_pipe_last_error.restype = C.c_char_p
        lib.ocvar_hip_pipe_set_templates.argtypes = [vp, vp, i]
        lib.ocvar_hip_pipe_set_camera.argtypes = [vp, vp]
        lib.ocvar_hip_pipe_detect_device.argtypes = [vp, vp, i, i, i, sz, C.c_longlong, i, vp, vp, i]
        lib.ocvar_hip_pipe_track_device.argtypes = [vp, vp, i, i, i, sz, C.c_longlong, i, i, vp, vp, i]
        lib.ocvar_hip_pipe_submit.argtypes = [vp, vp, i, i, i, sz, i, i, C.c_longlong]
        lib.ocvar_hip_pipe_collect.argtypes = [vp, C.POINTER(C.c_longlong), vp, vp, i]
        lib.ocvar_hip_pipe_in_flight.argtypes = [vp]
        lib.ocvar_hip_pipe_set_result_limit.argtypes = [vp, i]
        lib.ocvar_hip_enqueue_tracked.argtypes = [vp, vp, i, i, i, sz, i, i, vp, vp, vp]
        lib.ocvar_hip_set_tuning.argtypes = [vp, i, i]
        lib.ocvar_hip_build_info.argtypes = []
        lib.ocvar_hip_build_info.restype = C.c_char_p
        lib.ocvar_hip_destroy.restype = None
        lib.ocvar_hip_last_error.argtypes = [vp]
        lib.ocvar_hip_last_error.restype = C.c_char_p
        lib.ocvar_hip_set_templates.argtypes = [vp, vp, i]
        lib.ocvar_hip_set_camera.argtypes = [vp, vp]
        lib.ocvar_hip_detect_device.argtypes = [vp, vp, i, i, i, sz, i, i, vp, vp, vp, vp, i]
        lib.ocvar_hip_enqueue.argtypes = [vp, vp, i, i, i, sz, i, i, vp, vp, vp]
        lib.ocvar_hip_collect.argtypes = [vp, vp, vp, i]
        lib.ocvar_hip_detect_host.argtypes = [vp, vp, i, i, i, sz, i, i, vp, vp, vp, vp, i]
        lib.ocvar_hip_find_squares.argtypes = [vp, vp, i, i, i, vp, i, vp]
        lib.ocvar_hip_debug_gray.argtypes = [vp, i, vp]
        lib.ocvar_hip_debug_binary.argtypes = [vp, i, vp]
        lib.ocvar_hip_debug_masks.argtypes = [vp, i, vp]
        lib.ocvar_hip_debug_frame_quads.argtypes = [vp, i, vp, vp]
        lib.ocvar_hip_debug_candidates.argtypes = [vp, i, vp, i, vp]
        lib.ocvar_hip_stage_ms.argtypes = [vp, vp, i]
        lib.ocvar_hip_counters.argtypes = [vp, vp, i]
        lib.ocvar_hip_stage_stamps.argtypes = [vp, vp, vp, i]
        lib.ocvar_hip_stream.argtypes = [vp]
        lib.ocvar_hip_stream.restype = vp
        lib.ocvar_hip_results_to_device.argtypes = [vp, vp, vp, vp]
        lib.ocvar_hip_results_to_device_ex.argtypes = [vp, vp, vp, i, vp]
        lib.ocvar_hip_debug_calibrate.argtypes = [vp, sz]
        _hip = lib
    return _hip


_host = None


def host_lib():
    """The host C++ mirror of the reference's public API (libopencv-ar.so: cvarLoadTemplateTag, cvarReadCamera, ...)."""
    global _host
    if _host is None:
        if not os.path.exists(HOST_LIB):
            raise OcvarError(f"{HOST_LIB} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        hip_lib()   # libopencv-ar.so links against the HIP library: same runtime-ordering rule
        lib = C.CDLL(HOST_LIB)
        lib.cvarLoadTemplateTag.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
        lib.cvarReadCamera.argtypes = [C.c_char_p, C.c_void_p]
        lib.cvarCameraScale.argtypes = [C.c_void_p, C.c_int, C.c_int]
        _host = lib
    return _host


TEMPLATE_DIR = os.path.join(os.path.dirname(_HERE), "assets", "templates")


def load_templates(paths, scale=0.01):
    """cvarLoadTemplateTag (reference opencvar.cpp:284-321) on each PNG -> list of Template."""
    out = []
    for path in paths:
        t = Template()
        if host_lib().cvarLoadTemplateTag(C.byref(t), os.fsencode(path), scale) != 1:
            raise OcvarError(f"cvarLoadTemplateTag failed for {path}")
        out.append(t)
    return out


def default_camera(width, height, filename=None):
    """cvarReadCamera(filename or NULL) then cvarCameraScale(width, height) (opencvar.cpp:37-102)."""
    cam = Camera()
    if host_lib().cvarReadCamera(os.fsencode(filename) if filename else None, C.byref(cam)) != 1:
        raise OcvarError(f"cvarReadCamera failed for {filename}")
    host_lib().cvarCameraScale(C.byref(cam), width, height)
    return cam


def _ptr(a):
    return C.c_void_p(a.ctypes.data) if a is not None else None


class Gate:
    """At most `width` binarise kernels of the detectors that share the gate run at once (include/ocvar_hip.h)."""

    def __init__(self, width=2, device=0):
        self._lib = hip_lib()
        self._g = C.c_void_p()
        rc = self._lib.ocvar_hip_gate_create(C.byref(self._g), device, width)
        if rc != 0:
            raise OcvarError(f"ocvar_hip_gate_create failed ({rc})")

    def __del__(self):
        if getattr(self, "_g", None):
            self._lib.ocvar_hip_gate_destroy(self._g)
            self._g = None


class Pipe:
    """Several contexts on one GPU as one detector (include/ocvar_hip.h: ocvar_hip_pipe_*): device-resident frames, any number
    of them, detected chunk by chunk with one chunk in flight per context."""

    def __init__(self, max_width, max_height, chunk_frames=2048, n_contexts=4, gate_width=2, device=0):
        self._lib = hip_lib()
        self._p = C.c_void_p()
        rc = self._lib.ocvar_hip_pipe_create(C.byref(self._p), device, max_width, max_height, chunk_frames, n_contexts, gate_width)
        if rc != 0:
            msg = self._lib.ocvar_hip_pipe_last_error(self._p).decode() if self._p else ""
            self.close()
            raise OcvarError(f"ocvar_hip_pipe_create failed ({rc}): {msg}")

    def close(self):
        if getattr(self, "_p", None):
            self._lib.ocvar_hip_pipe_destroy(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:   # interpreter shutdown: module globals may be gone
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise OcvarError(f"{what} failed ({rc}): {self._lib.ocvar_hip_pipe_last_error(self._p).decode()}")

    def set_templates(self, templates):
        arr = (Template * len(templates))(*templates)
        self._check(self._lib.ocvar_hip_pipe_set_templates(self._p, arr, len(templates)), "pipe_set_templates")

    def set_camera(self, camera):
        self._check(self._lib.ocvar_hip_pipe_set_camera(self._p, C.byref(camera)), "pipe_set_camera")

    def detect_device(self, d_ptr, width, height, n_frames, row_stride=None, frame_stride=None, grey_in_place=False, max_per_frame=MAX_MARKERS):
        row_stride = row_stride or 3 * width
        frame_stride = frame_stride or row_stride * height
        markers = np.zeros((n_frames, max_per_frame), MARKER_DTYPE)
        counts = np.zeros(n_frames, np.int32)
        self._check(self._lib.ocvar_hip_pipe_detect_device(self._p, d_ptr, width, height, row_stride, frame_stride, n_frames,
                                                           int(grey_in_place), _ptr(markers), _ptr(counts), max_per_frame), "pipe_detect_device")
        return markers, counts

    def submit(self, d_ptr, width, height, n_frames, tag=0, row_stride=None, frame_stride=None, grey_in_place=False):
        """streaming form: hand the next chunk to the next context; False when every context already has one in flight"""
        row_stride = row_stride or 3 * width
        frame_stride = frame_stride or row_stride * height
        rc = self._lib.ocvar_hip_pipe_submit(self._p, d_ptr, width, height, row_stride, frame_stride, n_frames, int(grey_in_place), tag)
        if rc == -6:
            return False
        self._check(rc, "pipe_submit")
        return True

    def collect(self, max_frames, max_per_frame=MAX_MARKERS):
        """the oldest chunk in flight: (tag, markers [n][max_per_frame], counts [n]) or None when nothing is in flight"""
        markers = np.zeros((max_frames, max_per_frame), MARKER_DTYPE)
        counts = np.zeros(max_frames, np.int32)
        tag = C.c_longlong(0)
        n = self._lib.ocvar_hip_pipe_collect(self._p, C.byref(tag), _ptr(markers), _ptr(counts), max_per_frame)
        if n < 0:
            self._check(n, "pipe_collect")
        if n == 0:
            return None
        return tag.value, markers[:n], counts[:n]

    def in_flight(self):
        return self._lib.ocvar_hip_pipe_in_flight(self._p)

    def set_result_limit(self, max_per_frame):
        self._check(self._lib.ocvar_hip_pipe_set_result_limit(self._p, max_per_frame), "pipe_set_result_limit")

    def track_device(self, d_ptr, width, height, n_streams, reset=False, row_stride=None, frame_stride=None, grey_in_place=False,
                     max_per_frame=MAX_MARKERS):
        """one time step of n_streams video streams (frame s = stream s); the streams' markers of the previous step stay on the device"""
        row_stride = row_stride or 3 * width
        frame_stride = frame_stride or row_stride * height
        markers = np.zeros((n_streams, max_per_frame), MARKER_DTYPE)
        counts = np.zeros(n_streams, np.int32)
        self._check(self._lib.ocvar_hip_pipe_track_device(self._p, d_ptr, width, height, row_stride, frame_stride, n_streams,
                                                          int(grey_in_place), int(reset), _ptr(markers), _ptr(counts), max_per_frame), "pipe_track_device")
        return markers, counts


def build_info():
    """how libocvar_hip.so was built (flags that change what the kernels do: profiling hooks, tuning switches)"""
    return hip_lib().ocvar_hip_build_info().decode()


class Detector:
    """A device context: batches of frames -> CvarMarker arrays (cvarArMultRegistration semantics per frame)."""

    def __init__(self, max_width, max_height, max_batch=1, device=0):
        self._lib = hip_lib()
        self._ctx = C.c_void_p()
        rc = self._lib.ocvar_hip_create(C.byref(self._ctx), device, max_width, max_height, max_batch)
        if rc != 0:
            msg = self._lib.ocvar_hip_last_error(self._ctx).decode() if self._ctx else "no gfx950 device"
            if self._ctx:
                self._lib.ocvar_hip_destroy(self._ctx)
                self._ctx = C.c_void_p()
            raise OcvarError(f"ocvar_hip_create failed ({rc}): {msg}")
        self.max_batch = max_batch
        self.n_templates = 0

    def close(self):
        if self._ctx:
            self._lib.ocvar_hip_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise OcvarError(f"{what} failed ({rc}): {self._lib.ocvar_hip_last_error(self._ctx).decode()}")

    def set_templates(self, templates):
        arr = (Template * len(templates))(*templates)
        self._check(self._lib.ocvar_hip_set_templates(self._ctx, arr, len(templates)), "set_templates")
        self.n_templates = len(templates)

    def set_camera(self, camera):
        self._check(self._lib.ocvar_hip_set_camera(self._ctx, C.byref(camera)), "set_camera")

    @staticmethod
    def _prev_arrays(prev, n_frames):
        if prev is None:
            return None, None
        pm = np.zeros((n_frames, MAX_MARKERS), MARKER_DTYPE)
        pc = np.zeros(n_frames, np.int32)
        for f, lst in enumerate(prev):
            pc[f] = min(len(lst), MAX_MARKERS)
            for k in range(pc[f]):
                pm[f, k] = lst[k]
        return pm, pc

    def enqueue_device(self, d_ptr, width, height, n_frames, row_stride=None, frame_stride=None, grey_in_place=False,
                       prev=None, stream=None):
        row_stride = row_stride or 3 * width
        frame_stride = frame_stride or row_stride * height
        pm, pc = self._prev_arrays(prev, n_frames)
        self._keep = (pm, pc)
        self._check(self._lib.ocvar_hip_enqueue(self._ctx, d_ptr, width, height, row_stride, frame_stride, n_frames,
                                                int(grey_in_place), _ptr(pm), _ptr(pc), stream), "enqueue")
        self._n = n_frames

    def collect(self, max_per_frame=MAX_MARKERS):
        n = self._n
        markers = np.zeros((n, max_per_frame), MARKER_DTYPE)
        counts = np.zeros(n, np.int32)
        self._check(self._lib.ocvar_hip_collect(self._ctx, _ptr(markers), _ptr(counts), max_per_frame), "collect")
        return markers, counts

    def ready(self):
        """True once the enqueued batch has finished (collect() will not wait)"""
        rc = self._lib.ocvar_hip_ready(self._ctx)
        if rc < 0:
            self._check(rc, "ready")
        return rc == 1

    def set_result_limit(self, max_per_frame):
        """marker records per frame a batch brings to the host (default MAX_MARKERS); counts stay the full counts"""
        self._check(self._lib.ocvar_hip_set_result_limit(self._ctx, max_per_frame), "set_result_limit")

    TUNE = {"crop_phases": 1, "mid_steps": 2, "mid_blocks": 3, "long_blocks": 4, "short_blocks": 5, "min_units": 6, "hp_mask": 7, "gate_mode": 8}

    def set_tuning(self, **kw):
        """result-invariant launch parameters (include/ocvar_hip.h: OCVAR_TUNE_*); 0 restores a default"""
        for k, v in kw.items():
            self._check(self._lib.ocvar_hip_set_tuning(self._ctx, self.TUNE[k], int(v)), f"set_tuning({k})")

    def set_gate(self, gate):
        """share a Gate with other detectors on the same GPU (None removes it); the detector keeps the gate alive"""
        self._check(self._lib.ocvar_hip_set_gate(self._ctx, gate._g if gate is not None else None), "set_gate")
        self._gate = gate

    def results_to_device(self, d_markers_ptr, d_counts_ptr, stream=None, per_frame=MAX_MARKERS):
        """Copies the enqueued batch's [n][per_frame] marker records (the first per_frame of every frame) and [n] full counts
        into caller-owned device buffers (stream-ordered), e.g. torch tensors handed to an RCCL gather."""
        self._check(self._lib.ocvar_hip_results_to_device_ex(self._ctx, d_markers_ptr, d_counts_ptr, per_frame, stream), "results_to_device")

    def detect_device(self, d_ptr, width, height, n_frames, **kw):
        max_per_frame = kw.pop("max_per_frame", MAX_MARKERS)
        self.enqueue_device(d_ptr, width, height, n_frames, **kw)
        return self.collect(max_per_frame)

    def detect_host(self, frames, grey_in_place=False, prev=None, max_per_frame=MAX_MARKERS):
        """frames: uint8 array [n, H, W, 3] (C-contiguous).  Greyed in place when asked (reference side effect)."""
        assert frames.dtype == np.uint8 and frames.ndim == 4 and frames.shape[3] == 3 and frames.flags.c_contiguous
        n, h, w, _ = frames.shape
        markers = np.zeros((n, max_per_frame), MARKER_DTYPE)
        counts = np.zeros(n, np.int32)
        pm, pc = self._prev_arrays(prev, n)
        self._check(self._lib.ocvar_hip_detect_host(self._ctx, _ptr(frames), w, h, 3 * w, 3 * w * h, n, int(grey_in_place),
                                                    _ptr(pm), _ptr(pc), _ptr(markers), _ptr(counts), max_per_frame),
                    "detect_host")
        self._n = n
        return markers, counts

    def find_squares(self, gray):
        g = np.ascontiguousarray(gray, dtype=np.uint8)
        h, w = g.shape
        quads = np.zeros((MAX_QUADS, 4, 2), np.int32)
        n = C.c_int(0)
        self._check(self._lib.ocvar_hip_find_squares(self._ctx, _ptr(g), w, h, w, _ptr(quads), MAX_QUADS, C.byref(n)),
                    "find_squares")
        return quads[:min(n.value, MAX_QUADS)].copy(), n.value

    # parity hooks
    def debug_gray(self, frame, width, height):
        out = np.zeros((height, width), np.uint8)
        self._check(self._lib.ocvar_hip_debug_gray(self._ctx, frame, _ptr(out)), "debug_gray")
        return out

    def debug_binary(self, frame, width, height):
        out = np.zeros((height & -2, width & -2), np.uint8)
        self._check(self._lib.ocvar_hip_debug_binary(self._ctx, frame, _ptr(out)), "debug_binary")
        return out

    def debug_masks(self, frame, width, height):
        """8-neighbour masks of the frame pass (bit s: the neighbour in direction s = E,NE,N,NW,W,SW,S,SE is set), untiled."""
        out = np.zeros((height & -2, width & -2), np.uint8)
        self._check(self._lib.ocvar_hip_debug_masks(self._ctx, frame, _ptr(out)), "debug_masks")
        return out

    def debug_frame_quads(self, frame):
        quads = np.zeros((MAX_QUADS, 4, 2), np.int32)
        n = C.c_int(0)
        self._check(self._lib.ocvar_hip_debug_frame_quads(self._ctx, frame, _ptr(quads), C.byref(n)), "debug_frame_quads")
        return quads[:min(n.value, MAX_QUADS)].copy()

    def debug_candidates(self, frame, max_cands=MAX_QUADS * MAX_TEMPLATES):
        arr = (Candidate * max_cands)()
        n = C.c_int(0)
        self._check(self._lib.ocvar_hip_debug_candidates(self._ctx, frame, arr, max_cands, C.byref(n)), "debug_candidates")
        return [arr[i] for i in range(min(n.value, max_cands))]

    def stage_ms(self):
        ms = np.zeros(12, np.float32)
        k = self._lib.ocvar_hip_stage_ms(self._ctx, _ptr(ms), 12)
        return ms[:max(k, 0)]

    def stream_ptr(self):
        """the context's own hipStream_t as an integer (torch.cuda.ExternalStream(ptr) wraps it)"""
        return self._lib.ocvar_hip_stream(self._ctx)

    def stage_stamps(self, ref_event):
        """the 13 stage boundaries of the last batch in ms after ref_event (a raw hipEvent_t, e.g. torch.cuda.Event(enable_timing=True).cuda_event)"""
        ms = np.zeros(13, np.float32)
        k = self._lib.ocvar_hip_stage_stamps(self._ctx, ref_event, _ptr(ms), 13)
        return ms[:max(k, 0)]

    def counters(self, n=10):
        """work counters of the last batch (n = 42 adds the profiling slots of a -DOCVAR_PROF build)"""
        out = np.zeros(n, np.int64)
        k = self._lib.ocvar_hip_counters(self._ctx, _ptr(out), n)
        return out[:max(k, 0)]
