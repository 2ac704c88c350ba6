// decode_core.h -- per-candidate homography, bilinear un-warp of the code patch, bit readout, template match.
//
// Replaces what the reference does per (quad, template) in cvarArMultRegistration
// (/root/reference/src/opencvar.cpp:706-760): cvarSquare (437-458) + cvarInvertPerspective (510-516:
// cvGetPerspectiveTransform in double rounded to float32, cvWarpPerspective = invert in double, 1/32-px
// coordinate quantisation, 15-bit bilinear weights, constant-0 border), BGR2GRAY (identity on grey),
// cvThreshold(>100), acArray2DToBit with the stride quirk (acmath.cpp:546-554 read with stride w from a
// widthStep-aligned buffer; padding reads as 0), the 4-code comparison and the orient 2/4 corner rotation.
// Only the inner tw x th samples of the (tw+2) x (th+2) patch are ever read by the reference, so only
// those are computed.
#pragma once
#include "hd.h"
#include <math.h>

namespace ocvar {

struct TemplateRec {  // == CvarTemplate (include/opencvar/opencvar.h), 48 bytes
    int width, height;
    double scale;
    long long code[4];
};

// cvGetPerspectiveTransform(src quad -> (0,0) (W-1,0) (W-1,H-1) (0,H-1)), in double, rounded to float32.  Four point pairs
// determine the map, so it is written down in closed form instead of eliminating the 8x8 system: the projective map of the
// unit square onto the source quad (Heckbert), scaled to the destination rectangle, inverted by its adjugate and normalised to
// m[8] = 1.  No pivot search and no indexed rows: on the GPU the system's 72 doubles lived in scratch memory.  (OpenCV 2.4
// solves by SVD, the oracle by Jacobi SVD, round 2's device code by Gaussian elimination: three roundings of the same
// rational function, equal after the cast to float32 up to rare last-bit ties -- tools/fuzz_parity.py compares every decoded
// code, orientation and corner with the oracle.)
OCVAR_HD bool perspective_from_quad(const float* src, int W, int H, float* m) {
    const double x0 = src[0], y0 = src[1], x1 = src[2], y1 = src[3], x2 = src[4], y2 = src[5], x3 = src[6], y3 = src[7];
    const double dx1 = x1 - x2, dy1 = y1 - y2, dx2 = x3 - x2, dy2 = y3 - y2, sx = x0 - x1 + x2 - x3, sy = y0 - y1 + y2 - y3;
    const double den = dx1 * dy2 - dx2 * dy1;
    if (den == 0.0 || W < 2 || H < 2) return false;
    const double g = (sx * dy2 - dx2 * sy) / den, h = (dx1 * sy - sx * dy1) / den;
    // unit square -> quad, with the rectangle's scale folded in: (X, Y) = ((W-1) u, (H-1) v)
    const double iu = 1.0 / (double)(W - 1), iv = 1.0 / (double)(H - 1);
    const double a = (x1 - x0 + g * x1) * iu, b = (x3 - x0 + h * x3) * iv, c = x0;
    const double d = (y1 - y0 + g * y1) * iu, e = (y3 - y0 + h * y3) * iv, f = y0;
    const double gg = g * iu, hh = h * iv;
    // inverse of [a b c; d e f; gg hh 1] up to scale: the adjugate
    const double A0 = e - f * hh, A1 = c * hh - b, A2 = b * f - c * e;
    const double A3 = f * gg - d, A4 = a - c * gg, A5 = c * d - a * f;
    const double A6 = d * hh - e * gg, A7 = b * gg - a * hh, A8 = a * e - b * d;
    if (A8 == 0.0) return false;
    const double s = 1.0 / A8;
    m[0] = (float)(A0 * s); m[1] = (float)(A1 * s); m[2] = (float)(A2 * s);
    m[3] = (float)(A3 * s); m[4] = (float)(A4 * s); m[5] = (float)(A5 * s);
    m[6] = (float)(A6 * s); m[7] = (float)(A7 * s);
    m[8] = 1.0f;
    return true;
}

OCVAR_HD void invert_map(const float* m32, double* M) {
    double S[9];
    for (int i = 0; i < 9; i++) S[i] = m32[i];
    double d = S[0] * (S[4] * S[8] - S[5] * S[7]) - S[1] * (S[3] * S[8] - S[5] * S[6]) + S[2] * (S[3] * S[7] - S[4] * S[6]);
    if (d != 0.) {
        d = 1. / d;
        M[0] = (S[4] * S[8] - S[5] * S[7]) * d;
        M[1] = (S[2] * S[7] - S[1] * S[8]) * d;
        M[2] = (S[1] * S[5] - S[2] * S[4]) * d;
        M[3] = (S[5] * S[6] - S[3] * S[8]) * d;
        M[4] = (S[0] * S[8] - S[2] * S[6]) * d;
        M[5] = (S[2] * S[3] - S[0] * S[5]) * d;
        M[6] = (S[3] * S[7] - S[4] * S[6]) * d;
        M[7] = (S[1] * S[6] - S[0] * S[7]) * d;
        M[8] = (S[0] * S[4] - S[1] * S[3]) * d;
    } else {
        for (int i = 0; i < 9; i++) M[i] = 0;
    }
}

// One destination pixel (x,y) of cvWarpPerspective(INTER_LINEAR, fill 0) on a single-channel crop.
// (px(ix, iy): the crop's pixel at column ix, row iy -- the device reads the panelled grey plane through it)
template <class Px>
OCVAR_HD int warp_sample_px(Px px, int cw, int ch, const double* M, int x, int y) {
    const double X0 = M[0] * 0 + M[1] * y + M[2];
    const double Y0 = M[3] * 0 + M[4] * y + M[5];
    const double W0 = M[6] * 0 + M[7] * y + M[8];
    double W = W0 + M[6] * x;
    W = W ? 32. / W : 0;
    const double fX = fmax(-2147483648.0, fmin(2147483647.0, (X0 + M[0] * x) * W));
    const double fY = fmax(-2147483648.0, fmin(2147483647.0, (Y0 + M[3] * x) * W));
    const int X = (int)rint(fX), Y = (int)rint(fY);
    int ix = X >> 5, iy = Y >> 5;
    ix = ix > 32767 ? 32767 : (ix < -32768 ? -32768 : ix);
    iy = iy > 32767 ? 32767 : (iy < -32768 ? -32768 : iy);
    const int fx = X & 31, fy = Y & 31;
    int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
    if ((fx | fy) == 0) {  // first entry of OpenCV's fixed-point bilinear table after its sum fix-up
        w00 = 32767;
        w11 = 1;
    }
    const bool x0ok = ix >= 0 && ix < cw, x1ok = ix + 1 >= 0 && ix + 1 < cw;
    const bool y0ok = iy >= 0 && iy < ch, y1ok = iy + 1 >= 0 && iy + 1 < ch;
    const int p00 = (x0ok && y0ok) ? px(ix, iy) : 0;
    const int p01 = (x1ok && y0ok) ? px(ix + 1, iy) : 0;
    const int p10 = (x0ok && y1ok) ? px(ix, iy + 1) : 0;
    const int p11 = (x1ok && y1ok) ? px(ix + 1, iy + 1) : 0;
    return (p00 * w00 + p01 * w01 + p10 * w10 + p11 * w11 + 16384) >> 15;
}
OCVAR_HD int warp_sample(const uint8_t* crop, int cw, int ch, int stride, const double* M, int x, int y) {
    return warp_sample_px([=](int ix, int iy) -> int { return crop[(long long)iy * stride + ix]; }, cw, ch, M, x, y);
}

// The bit of flat index k (k = i*tw + j of acArray2DToBit's arr[i*w+j]) lives at pixel (k % ws, k / ws) of
// the widthStep-aligned tw x th buffer, or in padding (0) when k % ws >= tw.
OCVAR_HD bool code_cell(int k, int tw, int th, int* cx, int* cy) {
    const int ws = (tw + 3) & ~3;
    *cx = k % ws;
    *cy = k / ws;
    return *cx < tw && *cy < th;
}

OCVAR_HD long long read_code(const uint8_t* crop, int cw, int ch, int stride, const float* patPoint, int tw, int th) {
    float m32[9];
    double M[9];
    if (!perspective_from_quad(patPoint, tw + 2, th + 2, m32))
        for (int i = 0; i < 9; i++) m32[i] = 0.f;
    invert_map(m32, M);
    long long bit = 0;
    for (int i = 0; i < th; i++)
        for (int j = tw - 1; j >= 0; j--) {
            int cx, cy, v = 0;
            if (code_cell(i * tw + j, tw, th, &cx, &cy)) v = warp_sample(crop, cw, ch, stride, M, cx + 1, cy + 1) > 100;
            bit = (bit << 1) | v;
        }
    return bit;
}

OCVAR_HD int match_orient(long long bit, const TemplateRec& t) {
    for (int k = 0; k < 4; k++)
        if (bit == t.code[k]) return k + 1;
    return 0;
}

// cvarRotSquare (opencvar.cpp:464-501): src[(rot-1+i)%4] = old[i]
OCVAR_HD void rot_square(float* sq, int rot) {
    float t[8];
    for (int i = 0; i < 8; i++) t[i] = sq[i];
    for (int i = 0; i < 4; i++) {
        const int d = (rot - 1 + i) & 3;
        sq[2 * d] = t[2 * i];
        sq[2 * d + 1] = t[2 * i + 1];
    }
}

// cvarSquare2Rect (546-562) grown by 5 px (683-686) and clipped by cvSetImageROI (688).
OCVAR_HD void crop_rect(const int* quad, int W, int H, int* x0, int* y0, int* cw, int* ch) {
    int minx = 50000, miny = 50000, maxx = -50000, maxy = -50000;
    for (int i = 0; i < 4; i++) {
        minx = quad[2 * i] < minx ? quad[2 * i] : minx;
        maxx = quad[2 * i] > maxx ? quad[2 * i] : maxx;
        miny = quad[2 * i + 1] < miny ? quad[2 * i + 1] : miny;
        maxy = quad[2 * i + 1] > maxy ? quad[2 * i + 1] : maxy;
    }
    int rx = minx - 5, ry = miny - 5, rx1 = maxx + 5, ry1 = maxy + 5;
    rx = rx < 0 ? 0 : rx;
    ry = ry < 0 ? 0 : ry;
    rx1 = rx1 > W ? W : rx1;
    ry1 = ry1 > H ? H : ry1;
    *x0 = rx;
    *y0 = ry;
    *cw = rx1 - rx;
    *ch = ry1 - ry;
}

}  // namespace ocvar
