"""ctypes bindings used by the tests: the oracle (checker), the synthetic generator, templates."""
import ctypes as C
import json
import os
import subprocess

import numpy as np


def P(a):
    """numpy array -> void* argument"""
    return C.c_void_p(a.ctypes.data)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "opencv-ar_amd")


class Camera(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("cameraMatrix", C.c_double * 9),
                ("distCoeffs", C.c_double * 5), ("glProjection", C.c_double * 16)]


class Template(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("scale", C.c_double), ("code", C.c_longlong * 4)]


class Marker(C.Structure):
    _fields_ = [("glMatrix", C.c_double * 16), ("templateId", C.c_int), ("markerId", C.c_int),
                ("score", C.c_double), ("square", C.c_float * 8), ("aspectRatio", C.c_double)]


class Candidate(C.Structure):
    _fields_ = [("markerId", C.c_int), ("templateId", C.c_int), ("orient", C.c_int), ("pad", C.c_int),
                ("bit", C.c_longlong), ("square", C.c_float * 8), ("patPoint", C.c_float * 8)]


class SynthConfig(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("width", "height", "grid_x", "grid_y", "side_min", "side_max", "rot_mode",
                                        "corner_jitter_pct", "occlude_pct", "textured")]


class SynthMarker(C.Structure):
    _fields_ = [("corner", C.c_double * 8), ("template_index", C.c_int), ("quadrant", C.c_int),
                ("occluded", C.c_int), ("pad", C.c_int)]


class SynthTemplate(C.Structure):
    _fields_ = [("pixels", C.POINTER(C.c_uint8)), ("w", C.c_int), ("h", C.c_int)]


assert C.sizeof(Camera) == 248 and C.sizeof(Template) == 48 and C.sizeof(Marker) == 184


def _build(path, target, cwd):
    if not os.path.exists(path):
        subprocess.check_call(["make", "-C", cwd, target], stdout=subprocess.DEVNULL)
    return path


_oracle = None


def oracle():
    """The CPU restatement (checker only)."""
    global _oracle
    if _oracle is None:
        p = _build(os.path.join(ROOT, "oracle", "liboracle.so"), "liboracle.so", os.path.join(ROOT, "oracle"))
        lib = C.CDLL(p)
        lib.orc_arc_length_closed.restype = C.c_double
        lib.orc_contour_area.restype = C.c_double
        lib.orc_acCalcLength.restype = C.c_double
        lib.orc_acCalcLength.argtypes = [C.c_double] * 4
        lib.orc_approx_poly.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_void_p]
        lib.orc_readout_bits.restype = C.c_longlong
        lib.orc_acBitToArray2D.argtypes = [C.c_longlong, C.c_void_p, C.c_int, C.c_int]
        lib.orc_load_tag.argtypes = [C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_double]
        lib.orc_load_template_pixels.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double]
        lib.orc_square_to_matrix.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]
        _oracle = lib
    return _oracle


_synth = None


def synth_lib():
    global _synth
    if _synth is None:
        p = _build(os.path.join(PKG, "lib", "libocvar_synth.so"), "lib/libocvar_synth.so", PKG)
        lib = C.CDLL(p)
        lib.ocvar_synth_frame.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                          C.c_void_p, C.c_int]
        _synth = lib
    return _synth


_extra_templates = {}


def register_template(name, inner):
    """A template made up at run time (tools/fuzz_parity.py draws random code grids): inner = h x w array of 0/1 cells; the
    1-cell black frame is added here.  No golden codes exist for it (None): the oracle computes them."""
    g = np.pad((np.asarray(inner) > 0).astype(np.uint8) * 255, 1)
    _extra_templates[name] = (np.ascontiguousarray(g), None)


def template_pixels():
    """name -> uint8 array (full image incl. the 1-px frame) and expected codes."""
    with open(os.path.join(ROOT, "tests", "golden", "templates.json")) as f:
        d = json.load(f)
    out = {k: (np.array(v["pixels"], dtype=np.uint8), v["codes"]) for k, v in d.items()}
    out.update(_extra_templates)
    return out


TEMPLATE_ORDER = ["2x2-01", "3x3-01", "4x4-01"]
# synthetic code grids above the shipped sizes (tools/make_templates.py; codes pinned on the reference's compiled acmath.cpp):
# 5x5, 6x6, 7x7 take the widthStep != width side of the stride quirk, the 8x8 ones fill the long long (two of them negative)
BIG_TEMPLATES = ["5x5-s1", "6x6-s1", "7x7-s1", "8x8-s1", "8x8-neg", "8x8-corners"]


def synth_config(config_id, **over):
    cfg = SynthConfig()
    synth_lib().ocvar_synth_config(config_id, C.byref(cfg))
    for k, v in over.items():
        setattr(cfg, k, v)
    return cfg


def synth_frame(cfg, frame_index, names=None):
    """Returns (bgr HxWx3 uint8, list of truth dicts)."""
    names = names or TEMPLATE_ORDER
    tp = template_pixels()
    arrs = [np.ascontiguousarray(tp[n][0]) for n in names]
    st = (SynthTemplate * len(arrs))()
    for i, a in enumerate(arrs):
        st[i].pixels = a.ctypes.data_as(C.POINTER(C.c_uint8))
        st[i].h, st[i].w = a.shape
    bgr = np.zeros((cfg.height, cfg.width, 3), np.uint8)
    truth = (SynthMarker * 256)()
    n = synth_lib().ocvar_synth_frame(C.byref(cfg), frame_index, st, len(arrs), P(bgr), cfg.width * 3, truth, 256)
    out = [dict(corner=np.array(truth[i].corner).reshape(4, 2), template=truth[i].template_index,
                quadrant=truth[i].quadrant, occluded=truth[i].occluded) for i in range(n)]
    return bgr, out


def oracle_templates(names=None, scale=0.01):
    names = names or TEMPLATE_ORDER
    tp = template_pixels()
    arr = (Template * len(names))()
    for i, n in enumerate(names):
        px = np.ascontiguousarray(tp[n][0])
        oracle().orc_load_template_pixels(C.byref(arr[i]), P(px), px.shape[1], px.shape[0], scale)
    return arr


def oracle_camera(w, h):
    cam = Camera()
    oracle().orc_camera_default(C.byref(cam))
    oracle().orc_camera_scale(C.byref(cam), w, h)
    return cam


def oracle_registration(bgr, templates, cam, prev=None, max_markers=512, max_cands=4096):
    """Runs the oracle on a copy of bgr.  Returns (markers list, candidates list, greyed bgr)."""
    img = np.ascontiguousarray(bgr.copy())
    h, w = img.shape[:2]
    markers = (Marker * max_markers)()
    n_in = 0
    if prev:
        for i, m in enumerate(prev):
            markers[i] = m
        n_in = len(prev)
    cands = (Candidate * max_cands)()
    nc = C.c_int(0)
    n = oracle().orc_registration(P(img), w, h, w * 3, markers, n_in, max_markers, templates, len(templates),
                                  C.byref(cam), cands, max_cands, C.byref(nc))
    return [markers[i] for i in range(min(n, max_markers))], [cands[i] for i in range(min(nc.value, max_cands))], img


def oracle_find_squares(gray):
    g = np.ascontiguousarray(gray)
    h, w = g.shape
    q = np.zeros(8 * 4096, np.int32)
    n = oracle().orc_find_squares(P(g), w, h, w, P(q), 4096)
    return q[:8 * n].reshape(n, 4, 2)


def oracle_binarise(gray):
    g = np.ascontiguousarray(gray)
    h, w = g.shape
    out = np.zeros((h & -2, w & -2), np.uint8)
    oracle().orc_binarise(P(g), w, h, w, P(out), None)
    return out


def oracle_contours(bin_img):
    b = np.ascontiguousarray(bin_img)
    h, w = b.shape
    maxp, maxc = 4 * b.size + 16, b.size + 16
    pts = np.zeros(2 * maxp, np.int32)
    offs = np.zeros(maxc + 1, np.int32)
    starts = np.zeros(maxc, np.int32)
    holes = np.zeros(maxc, np.int32)
    n = oracle().orc_find_contours(P(b), w, h, P(pts), maxp, P(offs), P(starts),
                                   P(holes), maxc)
    assert n >= 0
    return [pts[2 * offs[i]:2 * offs[i + 1]].reshape(-1, 2).copy() for i in range(n)], starts[:n].copy(), holes[:n].copy()


def neighbour_masks(binary):
    """8-neighbour masks of a binary image as the frame pass stores them (bit s: the neighbour in direction s = E,NE,N,NW,W,SW,S,SE
    is set), with the 1-pixel frame zeroed first as cvFindContours does."""
    bz = (np.asarray(binary) > 0).astype(np.uint8)
    bz[0, :] = bz[-1, :] = 0
    bz[:, 0] = bz[:, -1] = 0
    pad = np.pad(bz, 1)
    h, w = bz.shape
    dx = [1, 1, 0, -1, -1, -1, 0, 1]
    dy = [0, -1, -1, -1, 0, 1, 1, 1]
    out = np.zeros_like(bz)
    for k in range(8):
        out |= pad[1 + dy[k]:1 + dy[k] + h, 1 + dx[k]:1 + dx[k] + w] << k
    return out
