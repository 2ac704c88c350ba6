#!/usr/bin/env python3
"""Writes assets/templates/*.png, tests/golden/templates.json and the PNG fixtures under tests/golden/png/.

Shipped templates: the pixel grids are the decoded content of the reference's template/{2x2,3x3,4x4}-01.png
(SURVEY.md Appendix C).  When /root/reference is present the grids are re-checked against a PIL decode of those
files; the PNGs written here are produced by PIL from the grids (data, not a copy of the reference's files).

Synthetic templates 5x5 .. 8x8 (the reference's cvarLoadTag defaults to 8x8 codes, include/opencvar/opencvar.h:174-175;
acArray2DToBit packs up to 64 cells into a long long, src/acmath.cpp:546-554): seeded random grids, among them 8x8 grids
whose first-read cell is 1 (the code is a NEGATIVE long long).  Their codes are computed by the reference's OWN
acArray2DToBit / acBitRotate -- src/acmath.cpp compiled where it lies into oracle/_ref/libacmath_ref.so (oracle/Makefile,
target `ref`) -- on the buffer cvarLoadTemplateTag hands it (opencvar.cpp:284-309: inner region, threshold > 100, vertical
flip, widthStep = align4(width) with zero padding, read with stride `width`): the same procedure reproduces SURVEY
Appendix C's codes of the three shipped templates, which is asserted below.  Run in the build container only; the JSON
and the PNGs (data) are what travels.

PNG fixtures (tests/golden/png/): every template as 8-bit grey written by PIL, plus -- written by the small encoder
below so that the row filter is chosen, not left to an encoder's heuristic -- the colour types 0 (grey), 2 (RGB),
3 (palette), 4 (grey + alpha), 6 (RGBA), each with filter types 0..4 (None, Sub, Up, Average, Paeth); PIL decodes every
one of them back to the grid here, so the fixtures are valid PNGs by an independent decoder."""
import ctypes as C
import json
import os
import struct
import zlib

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GRIDS = {
    "2x2-01": [[0,0,0,0],[0,1,0,0],[0,0,1,0],[0,0,0,0]],
    "3x3-01": [[0,0,0,0,0],[0,1,1,1,0],[0,1,1,0,0],[0,1,0,1,0],[0,0,0,0,0]],
    "4x4-01": [[0,0,0,0,0,0],[0,1,0,1,1,0],[0,0,1,1,1,0],[0,0,1,1,1,0],[0,1,0,1,1,0],[0,0,0,0,0,0]],
}
# SURVEY.md Appendix C, column (A): codes under the stride quirk with zero padding
CODES = {"2x2-01": [0x8,0x2,0x1,0x4], "3x3-01": [0x174,0x117,0x5d,0x1d1], "4x4-01": [0xdeed,0x96ff,0xb77b,0xff69]}
SHIPPED = list(GRIDS)


def synthetic_grids():
    """name -> full grid (1-cell black frame + n x n random cells); deterministic"""
    rng = np.random.default_rng(0x0C0A2013)
    out = {}
    for n in (5, 6, 7, 8):
        inner = (rng.random((n, n)) < 0.5).astype(np.uint8)
        out[f"{n}x{n}-s1"] = np.pad(inner, 1)
    # 8x8 whose code is negative: acArray2DToBit reads row 0 of the FLIPPED inner region first, its last column first, into
    # the most significant bit -- the bottom-right inner cell of the PNG
    neg = (rng.random((8, 8)) < 0.5).astype(np.uint8)
    neg[7, 7] = 1
    neg[0, 0] = 0   # (keeps the rotated codes distinct from code[0])
    out["8x8-neg"] = np.pad(neg, 1)
    allc = (rng.random((8, 8)) < 0.5).astype(np.uint8)
    allc[0, 0] = allc[0, 7] = allc[7, 0] = allc[7, 7] = 1   # every rotation's code is negative
    allc[0, 1] = 0
    out["8x8-corners"] = np.pad(allc, 1)
    return out


def reference_codes(grid, ref):
    """code[0..3] as cvarLoadTemplateTag + cvarLoadTag compute them, through the compiled reference acmath"""
    g = np.asarray(grid, np.uint8)
    inner = (g[1:-1, 1:-1] * 255 > 100).astype(np.uint8)[::-1]          # ROI, cvThreshold(100 -> 1), cvFlip (vertical)
    h, w = inner.shape
    step = (w + 3) & ~3                                                # IplImage widthStep of an 8-bit 1-channel image
    buf = np.zeros((h, step), np.uint8)
    buf[:, :w] = inner
    bit = C.c_longlong(0)
    ref.acArray2DToBit(C.c_void_p(buf.ctypes.data), w, h, C.byref(bit))  # reads w*h bytes with stride w (the quirk)
    codes = []
    for i in range(4):
        b = C.c_longlong(bit.value)
        ref.acBitRotate(C.byref(b), i, w, h)
        codes.append(b.value)
    return codes


# ---- a PNG encoder with a chosen row filter (8-bit, non-interlaced) -----------------------------------------------------------
def _filter_row(ft, cur, up, bpp):
    cur = [int(v) for v in cur]
    up = [int(v) for v in up]
    out = []
    for i, v in enumerate(cur):
        a = cur[i - bpp] if i >= bpp else 0
        b = up[i]
        c = up[i - bpp] if i >= bpp else 0
        if ft == 0: pred = 0
        elif ft == 1: pred = a
        elif ft == 2: pred = b
        elif ft == 3: pred = (a + b) >> 1
        else:
            p = a + b - c
            pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
            pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
        out.append((v - pred) & 255)
    return bytes([ft]) + bytes(out)


def write_png(path, rows, ctype, ft, palette=None):
    """rows: uint8 array [h][w*channels] of raw samples for colour type `ctype`; every row uses filter `ft`"""
    h = len(rows)
    bpp = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    w = rows.shape[1] // bpp
    raw = b""
    prev = np.zeros(rows.shape[1], np.uint8)
    for y in range(h):
        raw += _filter_row(ft, rows[y], prev, bpp)
        prev = rows[y]

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0))
    if palette is not None:
        png += chunk(b"PLTE", bytes(palette))
    comp = zlib.compress(raw, 9)
    half = len(comp) // 2
    png += chunk(b"IDAT", comp[:half]) + chunk(b"IDAT", comp[half:])   # two IDAT chunks: the reader must concatenate them
    png += chunk(b"IEND", b"")
    with open(path, "wb") as f:
        f.write(png)


def variant_rows(grid, ctype, rng):
    """raw samples of colour type ctype that decode (cvLoadImage GRAYSCALE, threshold 100) to the grid; cells are 0 / 255 like
    the reference's files; alpha is arbitrary (OpenCV drops it); the palette is shuffled"""
    g = (np.asarray(grid, np.uint8) * 255)
    h, w = g.shape
    palette = None
    if ctype == 0:
        rows = g.copy()
    elif ctype == 2:
        rows = np.repeat(g[:, :, None], 3, axis=2).reshape(h, 3 * w)
    elif ctype == 3:
        # index 0 unused, black at 5, white at 2: an index is not a grey value
        palette = [9, 9, 9, 77, 77, 77, 255, 255, 255, 1, 2, 3, 4, 5, 6, 0, 0, 0]
        rows = np.where(g > 0, 2, 5).astype(np.uint8)
    elif ctype == 4:
        rows = np.stack([g, rng.integers(0, 256, g.shape, dtype=np.uint8)], axis=2).reshape(h, 2 * w)
    else:
        rows = np.concatenate([np.repeat(g[:, :, None], 3, axis=2), rng.integers(0, 256, (h, w, 1), dtype=np.uint8)], axis=2).reshape(h, 4 * w)
    return np.ascontiguousarray(rows), palette


def main():
    os.makedirs(os.path.join(ROOT, "assets/templates"), exist_ok=True)
    png_dir = os.path.join(ROOT, "tests/golden/png")
    os.makedirs(png_dir, exist_ok=True)
    ref_path = os.path.join(ROOT, "oracle/_ref/libacmath_ref.so")
    assert os.path.exists(ref_path), "build oracle/_ref first: make -C oracle ref (needs /root/reference)"
    ref = C.CDLL(ref_path)
    ref.acBitRotate.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    out = {}
    for name, g in GRIDS.items():
        a = (np.array(g, dtype=np.uint8) * 255)
        path = f"/root/reference/template/{name}.png"
        if os.path.exists(path):
            r = np.array(Image.open(path).convert("L"))
            assert (r == a).all(), name
        Image.fromarray(a, "L").save(os.path.join(ROOT, f"assets/templates/{name}.png"))
        got = reference_codes(g, ref)
        assert got == CODES[name], (name, [hex(c) for c in got])   # the procedure reproduces SURVEY Appendix C
        out[name] = {"pixels": a.tolist(), "codes": CODES[name]}
    for name, g in synthetic_grids().items():
        codes = reference_codes(g, ref)
        out[name] = {"pixels": (g * 255).tolist(), "codes": codes, "source": "synthetic grid; codes by the reference's acmath.cpp (oracle/_ref)"}
    assert out["8x8-neg"]["codes"][0] < 0 and all(c < 0 for c in out["8x8-corners"]["codes"])
    with open(os.path.join(ROOT, "tests/golden/templates.json"), "w") as f:
        json.dump(out, f, indent=1)
    # PNG fixtures
    rng = np.random.default_rng(7)
    index = {}
    for name, v in out.items():
        g = (np.array(v["pixels"], np.uint8) > 0).astype(np.uint8)
        p = os.path.join(png_dir, f"{name}.png")
        Image.fromarray(g * 255, "L").save(p)
        index[f"{name}.png"] = name
        if name in ("2x2-01", "3x3-01", "4x4-01", "5x5-s1", "8x8-neg"):
            for ctype in (0, 2, 3, 4, 6):
                for ft in range(5):
                    rows, pal = variant_rows(g, ctype, rng)
                    fn = f"{name}-c{ctype}-f{ft}.png"
                    write_png(os.path.join(png_dir, fn), rows, ctype, ft, pal)
                    back = np.array(Image.open(os.path.join(png_dir, fn)).convert("RGB"))[..., 0]   # an independent decoder agrees
                    assert (back == g * 255).all(), fn
                    index[fn] = name
    with open(os.path.join(png_dir, "index.json"), "w") as f:
        json.dump(index, f, indent=0, sort_keys=True)
    print("ok:", len(out), "templates,", len(index), "png fixtures")


if __name__ == "__main__":
    main()
