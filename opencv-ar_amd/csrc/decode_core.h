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

// 8x8 DLT system, Gaussian elimination with partial pivoting, double; result rounded to float32.
OCVAR_HD bool perspective_from_quad(const float* src, int W, int H, float* m) {
    const float dst[8] = {0.f, 0.f, (float)(W - 1), 0.f, (float)(W - 1), (float)(H - 1), 0.f, (float)(H - 1)};
    double A[8][9];
    for (int i = 0; i < 4; i++) {
        const double sx = src[2 * i], sy = src[2 * i + 1], dx = dst[2 * i], dy = dst[2 * i + 1];
        A[i][0] = sx; A[i][1] = sy; A[i][2] = 1; A[i][3] = 0; A[i][4] = 0; A[i][5] = 0;
        A[i][6] = -sx * dx; A[i][7] = -sy * dx; A[i][8] = dx;
        A[i + 4][0] = 0; A[i + 4][1] = 0; A[i + 4][2] = 0; A[i + 4][3] = sx; A[i + 4][4] = sy; A[i + 4][5] = 1;
        A[i + 4][6] = -sx * dy; A[i + 4][7] = -sy * dy; A[i + 4][8] = dy;
    }
    for (int c = 0; c < 8; c++) {
        int piv = c;
        double best = fabs(A[c][c]);
        for (int r = c + 1; r < 8; r++) {
            const double v = fabs(A[r][c]);
            if (v > best) {
                best = v;
                piv = r;
            }
        }
        if (best == 0.0) return false;
        if (piv != c)
            for (int k = 0; k < 9; k++) {
                const double t = A[c][k];
                A[c][k] = A[piv][k];
                A[piv][k] = t;
            }
        for (int r = c + 1; r < 8; r++) {
            const double f = A[r][c] / A[c][c];
            for (int k = c; k < 9; k++) A[r][k] = A[r][k] - f * A[c][k];
        }
    }
    for (int c = 7; c >= 0; c--) {
        double s = A[c][8];
        for (int k = c + 1; k < 8; k++) s = s - A[c][k] * A[k][8];
        A[c][8] = s / A[c][c];
    }
    for (int i = 0; i < 8; i++) m[i] = (float)A[i][8];
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
OCVAR_HD int warp_sample(const uint8_t* crop, int cw, int ch, int stride, const double* M, int x, int y) {
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
    const int p00 = (x0ok && y0ok) ? crop[(long long)iy * stride + ix] : 0;
    const int p01 = (x1ok && y0ok) ? crop[(long long)iy * stride + ix + 1] : 0;
    const int p10 = (x0ok && y1ok) ? crop[(long long)(iy + 1) * stride + ix] : 0;
    const int p11 = (x1ok && y1ok) ? crop[(long long)(iy + 1) * stride + ix + 1] : 0;
    return (p00 * w00 + p01 * w01 + p10 * w10 + p11 * w11 + 16384) >> 15;
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
