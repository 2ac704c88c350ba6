"""Host C++ mirror of the reference API (lib/libopencv-ar.so): symbol surface and setup-side functions on CPU,
the full cvarArMultRegistration path through the ARTest-equivalent sample on GPU."""
import ctypes as C
import json
import os
import re
import subprocess

import numpy as np
import pytest

import helpers as H
from helpers import P

LIB = os.path.join(H.PKG, "lib", "libopencv-ar.so.1.0.0")


@pytest.fixture(scope="module")
def host():
    import __graft_entry__ as g
    g.build()
    return C.CDLL(LIB)


# /root/reference/include/opencvar/opencvar.h:84-262 (21 cvar*) and acmath.h:65-223 (30 ac*, acMatrixTranslate included)
CVAR = ["cvarReadCamera", "cvarCameraScale", "cvarCameraProjection", "cvarGlMatrix", "cvarFindSquares", "cvarSquareInit",
        "cvarReverseSquare", "cvarFindCamera", "cvarLoadTemplateTag", "cvarLoadTag", "cvarCompareSquare", "cvarDrawSquares",
        "cvarGetSquare", "cvarSquare", "cvarRotSquare", "cvarInvertPerspective", "cvarSquareToMatrix", "cvarSquare2Rect",
        "cvarGetAllSquares", "cvarTrack", "cvarArMultRegistration"]
AC = ["acVectorPrint", "acVectorAdd", "acVectorDeduct", "acVectorCrossProduct", "acVectorNormal", "acVectorMagnitude",
      "acVectorNormalise", "acVectorNormal2", "acRad2Deg", "acDeg2Rad", "acMatrixRotate", "acMatrixTranslate", "acMatrixScale",
      "acMatrixIdentity", "acMatrixDotProduct", "acMatrixMultiply", "acMatrixPrint", "acMatrixTranspose", "acMatrixToQuaternion",
      "acQuaternionToMatrix", "acAngle", "acCalcLength", "acMatrix4Invert", "acMatrix4GetDeterminant", "acMatrixDecompose",
      "acArray2DRotateub", "acArray2DPrintub", "acArray2DToBit", "acBitToArray2D", "acBitRotate"]


def test_exports_reference_symbols_unmangled(host):
    for name in CVAR + AC:
        assert hasattr(host, name), name
    out = subprocess.check_output(["objdump", "-p", LIB]).decode()
    assert "SONAME" in out and "libopencv-ar.so.1.0" in out  # CMakeLists.txt:28-39 of the reference


def test_acmath_host_against_golden(host):
    g = json.load(open(os.path.join(H.ROOT, "tests", "golden", "acmath_golden.json")))
    host.acBitToArray2D.argtypes = [C.c_longlong, C.c_void_p, C.c_int, C.c_int]
    host.acBitRotate.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    for c in g["array2d_to_bit"]:
        a = np.array(c["arr"], np.uint8)
        bit = C.c_longlong(0)
        host.acArray2DToBit(P(a), c["w"], c["h"], C.byref(bit))
        assert bit.value == c["bit"]
    for c in g["bit_rotate"]:
        b = C.c_longlong(c["bit"])
        host.acBitRotate(C.byref(b), c["rot"], c["n"], c["n"])
        assert b.value == c["out"]
    for c in g["quaternion"]:
        m = np.array(c["m"]); q = np.zeros(4)
        host.acMatrixToQuaternion(P(m), P(q))
        assert np.array_equal(q, np.array(c["q"]))
        m2 = np.zeros(16)
        host.acQuaternionToMatrix(P(q), P(m2))
        assert np.array_equal(m2, np.array(c["m2"]))


def test_template_loader_and_camera(host):
    host.cvarLoadTemplateTag.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
    tp = H.template_pixels()
    for name in H.TEMPLATE_ORDER:
        t = H.Template()
        path = os.path.join(H.ROOT, "assets", "templates", name + ".png").encode()
        assert host.cvarLoadTemplateTag(C.byref(t), path, 0.01) == 1
        assert [int(c) for c in t.code] == tp[name][1]
    t = H.Template()
    assert host.cvarLoadTemplateTag(C.byref(t), b"/nonexistent.png", 0.01) == 0
    host.cvarLoadTag.argtypes = [C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_double]
    for name in H.BIG_TEMPLATES:   # cvarLoadTag (opencvar.cpp:311-321) with the header's default 8x8 and the sizes below it
        codes = tp[name][1]
        n = tp[name][0].shape[0] - 2
        host.cvarLoadTag(C.byref(t), codes[0], n, n, 0.5)
        assert [int(c) for c in t.code] == codes and (t.width, t.height, t.scale) == (n, n, 0.5)
    cam, ref = H.Camera(), H.oracle_camera(1920, 1080)
    assert host.cvarReadCamera(None, C.byref(cam)) == 1
    host.cvarCameraScale(C.byref(cam), 1920, 1080)
    assert bytes(cam) == bytes(ref)
    assert host.cvarReadCamera(b"/nonexistent/camera.yml", C.byref(cam)) == 0   # opencvar.cpp:54-55


CAMERA_YAML = """%YAML:1.0
calibration_time: "Sun Oct  4 10:00:00 2026"
imageSize: [ 1280, 720 ]
flags: 0
cameraMatrix: !!opencv-matrix
   rows: 3
   cols: 3
   dt: d
   data: [ 1.0245e+03, 0., 6.395e+02, 0.,
       1.0251e+03, 3.5925e+02, 0., 0., 1. ]
distCoeffs: !!opencv-matrix
   rows: 5
   cols: 1
   dt: d
   data: [ -1.25e-01, 2.5e-01, 0., -3.0e-04,
       -1.0e-01 ]
avg_reprojection_error: 3.1e-01
"""

CAMERA_XML = """<?xml version="1.0"?>
<opencv_storage>
<imageSize>
  1280 720</imageSize>
<cameraMatrix type_id="opencv-matrix">
  <rows>3</rows>
  <cols>3</cols>
  <dt>d</dt>
  <data>
    1.0245e+03 0. 6.395e+02 0. 1.0251e+03 3.5925e+02 0. 0. 1.</data></cameraMatrix>
<distCoeffs type_id="opencv-matrix">
  <rows>5</rows>
  <cols>1</cols>
  <dt>d</dt>
  <data>
    -1.25e-01 2.5e-01 0. -3.0e-04 -1.0e-01</data></distCoeffs>
</opencv_storage>
"""


def test_png_reader_colour_types_and_filters(host):
    """cvarLoadTemplateTag (opencvar.cpp:284-309: cvLoadImage(GRAYSCALE) -> inner ROI -> threshold 100 -> flip -> code) on
    the PNG fixtures of tests/golden/png (tools/make_templates.py): 8-bit grey written by PIL, and colour types 0 / 2 (the
    reference's own 2x2 template is an RGB file) / 3 (palette) / 4 (grey + alpha) / 6 (RGBA), each with every row filter
    (None, Sub, Up, Average, Paeth) forced by the fixture writer and two IDAT chunks.  Every file gives the codes the
    reference's compiled acmath.cpp gives for its grid (tests/golden/templates.json)."""
    host.cvarLoadTemplateTag.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
    tp = H.template_pixels()
    png_dir = os.path.join(H.ROOT, "tests", "golden", "png")
    index = json.load(open(os.path.join(png_dir, "index.json")))
    seen = set()
    for fn, name in sorted(index.items()):
        t = H.Template()
        assert host.cvarLoadTemplateTag(C.byref(t), os.path.join(png_dir, fn).encode(), 0.01) == 1, fn
        assert [int(c) for c in t.code] == tp[name][1], fn
        assert (t.width, t.height) == (tp[name][0].shape[1] - 2, tp[name][0].shape[0] - 2), fn
        m = re.match(r".*-c(\d)-f(\d)\.png$", fn)
        if m:
            seen.add((int(m.group(1)), int(m.group(2))))
    assert seen == {(c, f) for c in (0, 2, 3, 4, 6) for f in range(5)}
    # what the reader does not decode is refused as a whole (return 0), never mis-read: 16-bit samples, Adam7 interlacing
    import struct
    import zlib

    def png(depth, ctype, interlace, raw):
        def chunk(t, d):
            return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
        return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 4, 4, depth, ctype, 0, 0, interlace)) +
                chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))
    import tempfile as _tf
    with _tf.TemporaryDirectory() as d:
        for label, data in (("16bit", png(16, 0, 0, b"".join(b"\0" + bytes(8) for _ in range(4)))),
                            ("interlaced", png(8, 0, 1, bytes(40)))):
            p = os.path.join(d, label + ".png")
            open(p, "wb").write(data)
            assert host.cvarLoadTemplateTag(C.byref(H.Template()), p.encode(), 0.01) == 0, label
    # damaged files are refused (return 0, opencvar.cpp:286-288), never half-read
    good = open(os.path.join(png_dir, "4x4-01-c6-f4.png"), "rb").read()
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        for label, data in (("truncated", good[:len(good) // 2]), ("not-a-png", b"P5 4 4 255 " + bytes(16)), ("empty", b"")):
            p = os.path.join(d, label + ".png")
            open(p, "wb").write(data)
            assert host.cvarLoadTemplateTag(C.byref(H.Template()), p.encode(), 0.01) == 0, label


@pytest.mark.skipif(not os.path.exists("/root/reference/template/2x2-01.png"), reason="reference tree not present (build container only)")
def test_reference_template_files_give_appendix_c_codes(host):
    """The reference's own three template PNGs (the 2x2 is an RGB file), read where they lie: SURVEY Appendix C's codes."""
    host.cvarLoadTemplateTag.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
    expect = {"2x2-01": [0x8, 0x2, 0x1, 0x4], "3x3-01": [0x174, 0x117, 0x5d, 0x1d1], "4x4-01": [0xdeed, 0x96ff, 0xb77b, 0xff69]}
    for name, codes in expect.items():
        t = H.Template()
        assert host.cvarLoadTemplateTag(C.byref(t), f"/root/reference/template/{name}.png".encode(), 0.01) == 1
        assert [int(c) for c in t.code] == codes, name


@pytest.mark.parametrize("text,ext", [(CAMERA_YAML, "yml"), (CAMERA_XML, "xml")])
def test_camera_file_reader(host, tmp_path, text, ext):
    """cvarReadCamera(filename) (opencvar.cpp:53-71): imageSize, cameraMatrix, distCoeffs of an OpenCV FileStorage
    document in the layout OpenCV's calibration sample writes, then the GL projection of cvarCameraProjection.
    Parity unpinned: the reference ships no camera file and OpenCV is absent; the expected values are the file's."""
    path = tmp_path / ("camera." + ext)
    path.write_text(text)
    cam = H.Camera()
    assert host.cvarReadCamera(str(path).encode(), C.byref(cam)) == 1
    assert (cam.width, cam.height) == (1280, 720)
    assert list(cam.cameraMatrix) == [1024.5, 0.0, 639.5, 0.0, 1025.1, 359.25, 0.0, 0.0, 1.0]
    assert list(cam.distCoeffs) == [-0.125, 0.25, 0.0, -3.0e-04, -0.1]
    # glProjection = transpose of cvarCameraProjection(glstyle=0) (opencvar.cpp:73-76, 104-127)
    p = np.zeros(16)
    host.cvarCameraProjection(C.byref(cam), P(p), 0)
    assert np.array_equal(np.array(list(cam.glProjection)).reshape(4, 4), p.reshape(4, 4).T)
    assert p[0] == 2 * 1024.5 / 1280 and p[5] == 2 * 1025.1 / 720
    # truncated document: missing distCoeffs -> 0, like a node that cannot be read
    bad = tmp_path / ("broken." + ext)
    bad.write_text(text[: text.index("distCoeffs")])
    assert host.cvarReadCamera(str(bad).encode(), C.byref(cam)) == 0


class _Pt(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float)]


class _Rect(C.Structure):
    _fields_ = [("x", C.c_int), ("y", C.c_int), ("width", C.c_int), ("height", C.c_int)]


def test_geometry_helpers_follow_the_reference(host):
    """cvarSquare2Rect (opencvar.cpp:546-562: float->int truncation at every assignment), cvarRotSquare (464-501:
    src[(rot-1+i)%4] = old[i]), cvarTrack (592-617: all four corners within 20 px under some cyclic shift, pt1 replaced),
    cvarSquare (437-458), cvarReverseSquare, cvarSquareToMatrix against the oracle's pose."""
    rng = np.random.default_rng(3)
    host.cvarSquare2Rect.restype = _Rect
    host.cvarSquare2Rect.argtypes = [C.c_void_p]
    host.cvarTrack.argtypes = [C.c_void_p, C.c_void_p]
    host.cvarRotSquare.argtypes = [C.c_void_p, C.c_int]
    for _ in range(50):
        pts = (rng.random((4, 2)) * 700 - 30).astype(np.float32)
        arr = (_Pt * 4)(*[_Pt(float(a), float(b)) for a, b in pts])
        r = host.cvarSquare2Rect(arr)
        x, y, x2, y2 = 50000, 50000, -50000, -50000
        for px, py in pts:      # the reference's loop, integer state compared with float coordinates
            if px < x: x = int(px)
            if px > x2: x2 = int(px)
            if py < y: y = int(py)
            if py > y2: y2 = int(py)
        assert (r.x, r.y, r.width, r.height) == (x, y, x2 - x, y2 - y)
        for rot in (1, 2, 3, 4):
            a = (_Pt * 4)(*[_Pt(float(p), float(q)) for p, q in pts])
            host.cvarRotSquare(a, rot)
            exp = np.zeros_like(pts)
            for i in range(4):
                exp[(rot - 1 + i) % 4] = pts[i]
            assert np.array_equal(np.array([[p.x, p.y] for p in a], np.float32), exp)
        shift = int(rng.integers(0, 4))
        near = np.roll(pts, -shift, axis=0) + rng.uniform(-10, 10, (4, 2)).astype(np.float32)   # pt2[(i+j)%4] ~ pt1[i]
        a1 = (_Pt * 4)(*[_Pt(float(p), float(q)) for p, q in pts])
        a2 = (_Pt * 4)(*[_Pt(float(p), float(q)) for p, q in np.roll(near, 2 * shift, axis=0)])
        got = host.cvarTrack(a1, a2)
        p2 = np.array([[p.x, p.y] for p in a2], np.float32)
        exp_hit, exp_pts = 0, pts
        for j in range(4):
            if all(np.hypot(*(pts[i] - p2[(i + j) % 4])) < 20 for i in range(4)):
                exp_hit, exp_pts = 1, np.array([p2[(i + j) % 4] for i in range(4)])
                break
        assert got == exp_hit
        assert np.array_equal(np.array([[p.x, p.y] for p in a1], np.float32), exp_pts)
    far = (_Pt * 4)(*[_Pt(1000.0 + i, 1000.0) for i in range(4)])
    assert host.cvarTrack((_Pt * 4)(*[_Pt(float(i), 0.0) for i in range(4)]), far) == 0
    # pose through the public helper against the oracle (same tolerance as the GPU parity bar)
    cam = H.oracle_camera(640, 480)
    sq = np.array([[200, 150], [330, 160], [320, 290], [190, 280]], np.float32)
    m_host, m_ref = np.zeros(16), np.zeros(16)
    host.cvarSquareToMatrix.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double]
    host.cvarSquareToMatrix(P(sq), C.byref(cam), P(m_host), 1.0)
    H.oracle().orc_square_to_matrix.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]
    H.oracle().orc_square_to_matrix(P(sq), C.byref(cam), 1.0, P(m_ref))
    assert np.abs(m_host - m_ref).max() <= 1e-4 * max(1.0, np.abs(m_ref).max())


class _CvMat(C.Structure):   # include/shim/opencv/cv.h (OpenCV's layout): type, step, refcount, hdr_refcount, data, rows, cols
    _fields_ = [("type", C.c_int), ("step", C.c_int), ("refcount", C.c_void_p), ("hdr_refcount", C.c_int), ("data", C.c_void_p),
                ("rows", C.c_int), ("cols", C.c_int)]


def test_find_camera_solves_the_marker_rectangle_and_refuses_anything_else(host, capfd):
    """cvarFindCamera (opencvar.cpp:261-278) is only ever handed cvarSquareInit's rectangle (229-245, 529-536): that is what
    the library solves (distortion coefficients included); other object points are refused loudly, not mis-solved."""
    cam = H.oracle_camera(640, 480)
    cam.distCoeffs[0], cam.distCoeffs[1] = -0.1, 0.03
    obj, img = np.zeros(12), np.array([200, 150, 330, 160, 320, 290, 190, 280], np.float64)
    mo, mi = _CvMat(), _CvMat()
    mo.data, mo.rows, mo.cols = obj.ctypes.data, 4, 3
    mi.data, mi.rows, mi.cols = img.ctypes.data, 4, 2
    host.cvarSquareInit.argtypes = [C.c_void_p, C.c_double]
    host.cvarSquareInit(C.byref(mo), 0.75)
    assert obj.tolist() == [-0.75, -1, 0, 0.75, -1, 0, 0.75, 1, 0, -0.75, 1, 0]
    got, ref = np.zeros(16), np.zeros(16)
    host.cvarFindCamera.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    host.cvarFindCamera(C.byref(cam), C.byref(mo), C.byref(mi), P(got))
    H.oracle().orc_square_to_matrix.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]
    H.oracle().orc_square_to_matrix(P(img.astype(np.float32)), C.byref(cam), 0.75, P(ref))
    assert np.abs(got - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max())
    obj[4] = -0.5   # no longer the rectangle
    untouched = np.full(16, 7.0)
    host.cvarFindCamera(C.byref(cam), C.byref(mo), C.byref(mi), P(untouched))
    assert (untouched == 7.0).all()
    assert "not the marker rectangle" in capfd.readouterr().err


@pytest.mark.gpu
def test_artest_headless_matches_oracle():
    exe = os.path.join(H.PKG, "bin", "artest_headless")
    png = os.path.join(H.ROOT, "assets", "templates", "2x2-01.png")
    out = subprocess.check_output([exe, png, "1", "0"], timeout=300).decode()
    cfg = H.synth_config(1)
    frame, _ = H.synth_frame(cfg, 0, ["2x2-01"])
    tpls, cam = H.oracle_templates(["2x2-01"]), H.oracle_camera(cfg.width, cfg.height)
    m1, _, _ = H.oracle_registration(frame, tpls, cam)
    m2, _, _ = H.oracle_registration(frame, tpls, cam, prev=m1)
    assert "template 2x2 codes 8 2 1 4" in out and "frame greyed in place: yes" in out
    counts = [int(x) for x in re.findall(r"pass \d: (\d+) marker", out)]
    assert counts == [len(m1), len(m2)]
    gls = [np.array([float(v) for v in line.split()[1:]]) for line in out.splitlines() if line.strip().startswith("gl")]
    for got, ref in zip(gls, m1 + m2):
        r = np.array(ref.glMatrix)
        assert np.abs(got - r).max() <= 1e-4 * max(1.0, np.abs(r).max()) + 1e-6
