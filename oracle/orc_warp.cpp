// orc_warp.cpp -- oracle: perspective un-warp and code readout (TEST INFRASTRUCTURE, see oracle.h).
//
// Restates the OpenCV 2.4.x arithmetic behind these reference call sites:
//   opencvar.cpp:513  cvGetPerspectiveTransform(src, dst, M32)   
//   opencvar.cpp:514  cvWarpPerspective(crop, pat, M32)          
//   opencvar.cpp:723-724 cvCvtColor + cvThreshold(>100 -> 1)     
//   opencvar.cpp:728  acArray2DToBit on a widthStep-aligned buffer with stride w (quirk D1)
// The 8x8 system is solved by Gaussian elimination with partial pivoting in double (OpenCV 2.4 uses
// an SVD solve; both are double and the result is rounded to float32 -- parity unpinned).
#include "oracle.h"
#include <cmath>
#include <cstring>
#include <vector>

static bool solve8(double A[8][9]) {
    for (int c = 0; c < 8; c++) {
        int piv = c;
        double best = fabs(A[c][c]);
        for (int r = c + 1; r < 8; r++)
            if (fabs(A[r][c]) > best) {
                best = fabs(A[r][c]);
                piv = r;
            }
        if (best == 0.0) return false;
        if (piv != c)
            for (int k = 0; k < 9; k++) {
                double t = A[c][k];
                A[c][k] = A[piv][k];
                A[piv][k] = t;
            }
        for (int r = c + 1; r < 8; r++) {
            double f = A[r][c] / A[c][c];
            for (int k = c; k < 9; k++) A[r][k] = A[r][k] - f * A[c][k];
        }
    }
    for (int c = 7; c >= 0; c--) {
        double s = A[c][8];
        for (int k = c + 1; k < 8; k++) s = s - A[c][k] * A[k][8];
        A[c][8] = s / A[c][c];
    }
    return true;
}

extern "C" void orc_get_perspective_transform(const float* src, const float* dst, float* m) {
    double A[8][9];
    for (int i = 0; i < 4; i++) {
        double sx = src[2 * i], sy = src[2 * i + 1], dx = dst[2 * i], dy = dst[2 * i + 1];
        double r0[9] = {sx, sy, 1, 0, 0, 0, -sx * dx, -sy * dx, dx};
        double r1[9] = {0, 0, 0, sx, sy, 1, -sx * dy, -sy * dy, dy};
        memcpy(A[i], r0, sizeof r0);
        memcpy(A[i + 4], r1, sizeof r1);
    }
    if (!solve8(A)) {
        for (int i = 0; i < 9; i++) m[i] = 0;
        return;
    }
    for (int i = 0; i < 8; i++) m[i] = (float)A[i][8];
    m[8] = 1.0f;
}

static inline int round_half_even(double v) { return (int)nearbyint(v); }

extern "C" void orc_warp_perspective_gray(const uint8_t* src, int sw, int sh, int sstride, const float* m32,
                                          uint8_t* dst, int dw, int dh) {
    double S[9], M[9];
    for (int i = 0; i < 9; i++) S[i] = m32[i];
    // cv::invert 3x3 closed form
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
    for (int y = 0; y < dh; y++) {
        double X0 = M[0] * 0 + M[1] * y + M[2];
        double Y0 = M[3] * 0 + M[4] * y + M[5];
        double W0 = M[6] * 0 + M[7] * y + M[8];
        for (int x = 0; x < dw; x++) {
            double W = W0 + M[6] * x;
            W = W ? 32. / W : 0;
            double fX = fmax(-2147483648.0, fmin(2147483647.0, (X0 + M[0] * x) * W));
            double fY = fmax(-2147483648.0, fmin(2147483647.0, (Y0 + M[3] * x) * W));
            int X = round_half_even(fX), Y = round_half_even(fY);
            int ix = X >> 5, iy = Y >> 5;
            if (ix > 32767) ix = 32767;
            if (ix < -32768) ix = -32768;
            if (iy > 32767) iy = 32767;
            if (iy < -32768) iy = -32768;
            int fx = X & 31, fy = Y & 31;
            int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
            if (fx == 0 && fy == 0) {  // BilinearTab_i[0] after the fixed-point sum fix-up (no output effect)
                w00 = 32767;
                w11 = 1;
            }
            auto px = [&](int xx, int yy) -> int {
                if (xx < 0 || yy < 0 || xx >= sw || yy >= sh) return 0;
                return src[(size_t)yy * sstride + xx];
            };
            int v = px(ix, iy) * w00 + px(ix + 1, iy) * w01 + px(ix, iy + 1) * w10 + px(ix + 1, iy + 1) * w11;
            dst[(size_t)y * dw + x] = (uint8_t)((v + 16384) >> 15);
        }
    }
}

extern "C" long long orc_readout_bits(const uint8_t* gray, int w, int h, int stride, const float* patPoint, int tw,
                                      int th) {
    int W = tw + 2, H = th + 2;
    float dstq[8] = {0, 0, (float)(W - 1), 0, (float)(W - 1), (float)(H - 1), 0, (float)(H - 1)};  // cvarSquare ccw=0
    float m[9];
    orc_get_perspective_transform(patPoint, dstq, m);
    std::vector<uint8_t> pat((size_t)W * H);
    orc_warp_perspective_gray(gray, w, h, stride, m, pat.data(), W, H);
    // inner tw x th -> cvCreateImage(8,1): widthStep = align4(tw), padding defined as 0
    int ws = (tw + 3) & ~3;
    std::vector<unsigned char> buf((size_t)ws * th + (size_t)tw * th, 0);
    for (int y = 0; y < th; y++)
        for (int x = 0; x < tw; x++) buf[(size_t)y * ws + x] = pat[(size_t)(y + 1) * W + (x + 1)] > 100 ? 1 : 0;
    long long bit = 0;
    orc_acArray2DToBit(buf.data(), tw, th, &bit);  // reads with stride tw (quirk D1)
    return bit;
}
