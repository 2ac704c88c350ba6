// orc_warp.cpp -- oracle: perspective un-warp and code readout (TEST INFRASTRUCTURE, see oracle.h).
//
// Restates the OpenCV 2.4.x arithmetic behind these reference call sites:
//   opencvar.cpp:513  cvGetPerspectiveTransform(src, dst, M32)   
//   opencvar.cpp:514  cvWarpPerspective(crop, pat, M32)          
//   opencvar.cpp:723-724 cvCvtColor + cvThreshold(>100 -> 1)     
//   opencvar.cpp:728  acArray2DToBit on a widthStep-aligned buffer with stride w (quirk D1)
// The 8x8 system is solved the way OpenCV 2.4 does it -- cvSolve(..., CV_SVD): a one-sided (Hestenes) Jacobi SVD in
// double, x = V diag(1/w) U^T b -- and the result is rounded to float32.  This is deliberately NOT the device path's
// solver (Gaussian elimination with partial pivoting, opencv-ar_amd/csrc/decode_core.h): the two agree to a few ulps of
// double, and the parity tests / tools/fuzz_parity.py show whether that ever survives the rounding to float32 and the
// 1/32-pixel quantisation of the warp into a different code bit (recorded in DESIGN.md section 5).  Parity unpinned: the
// sweep order and stopping rule of OpenCV's own JacobiSVD are not reproduced bit for bit.
#include "oracle.h"
#include <cmath>
#include <cstring>
#include <vector>

// x = pinv(A) b for an 8x8 A (row-major) by one-sided Jacobi: columns of W = A V are made mutually orthogonal by plane
// rotations accumulated in V; then w_j = |W_j|, U_j = W_j / w_j.
static bool solve8_svd(const double A[8][8], const double b[8], double x[8]) {
    double W[8][8], V[8][8];
    for (int i = 0; i < 8; i++)
        for (int j = 0; j < 8; j++) {
            W[i][j] = A[i][j];
            V[i][j] = i == j ? 1.0 : 0.0;
        }
    for (int sweep = 0; sweep < 60; sweep++) {
        bool rotated = false;
        for (int p = 0; p < 7; p++)
            for (int q = p + 1; q < 8; q++) {
                double alpha = 0, beta = 0, gamma = 0;
                for (int i = 0; i < 8; i++) {
                    alpha += W[i][p] * W[i][p];
                    beta += W[i][q] * W[i][q];
                    gamma += W[i][p] * W[i][q];
                }
                if (fabs(gamma) <= 1e-300 || fabs(gamma) <= 2.220446049250313e-16 * sqrt(alpha * beta)) continue;
                rotated = true;
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
                for (int i = 0; i < 8; i++) {
                    const double wp = W[i][p], wq = W[i][q];
                    W[i][p] = c * wp - sn * wq;
                    W[i][q] = sn * wp + c * wq;
                    const double vp = V[i][p], vq = V[i][q];
                    V[i][p] = c * vp - sn * vq;
                    V[i][q] = sn * vp + c * vq;
                }
            }
        if (!rotated) break;
    }
    double wmax = 0, w[8];
    for (int j = 0; j < 8; j++) {
        double n2 = 0;
        for (int i = 0; i < 8; i++) n2 += W[i][j] * W[i][j];
        w[j] = sqrt(n2);
        if (w[j] > wmax) wmax = w[j];
    }
    if (wmax == 0.0) return false;
    for (int i = 0; i < 8; i++) x[i] = 0.0;
    for (int j = 0; j < 8; j++) {
        if (w[j] <= 8 * 2.220446049250313e-16 * wmax) continue;   // cvSVBkSb: singular values below the threshold are dropped
        double ub = 0;
        for (int i = 0; i < 8; i++) ub += W[i][j] * b[i];
        const double f = ub / (w[j] * w[j]);   // (U_j . b) / w_j with U_j = W_j / w_j
        for (int i = 0; i < 8; i++) x[i] += V[i][j] * f;
    }
    return true;
}

extern "C" void orc_get_perspective_transform(const float* src, const float* dst, float* m) {
    double A[8][8], b[8], x[8];
    for (int i = 0; i < 4; i++) {
        const double sx = src[2 * i], sy = src[2 * i + 1], dx = dst[2 * i], dy = dst[2 * i + 1];
        const double r0[8] = {sx, sy, 1, 0, 0, 0, -sx * dx, -sy * dx};
        const double r1[8] = {0, 0, 0, sx, sy, 1, -sx * dy, -sy * dy};
        memcpy(A[i], r0, sizeof r0);
        memcpy(A[i + 4], r1, sizeof r1);
        b[i] = dx;
        b[i + 4] = dy;
    }
    if (!solve8_svd(A, b, x)) {
        for (int i = 0; i < 9; i++) m[i] = 0;
        return;
    }
    for (int i = 0; i < 8; i++) m[i] = (float)x[i];
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
