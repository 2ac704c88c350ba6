// pose_core.h -- planar pose of one quad -> OpenGL modelview matrix, one lane per surviving marker.
//
// Replaces cvarSquareToMatrix (/root/reference/src/opencvar.cpp:524-540) and what it reaches:
// cvarSquareInit (229-245), cvarFindCamera (261-278: cvFindExtrinsicCameraParams2 + cvRodrigues2) and
// cvarGlMatrix (133-152) with the acMatrixToQuaternion / acQuaternionToMatrix round trip
// (/root/reference/src/acmath.cpp:215-276).
//
// Same estimator as OpenCV's planar branch (undistorted normalised points -> 4-point homography -> r1,r2 normalised,
// r3 = r1 x r2, nearest rotation, then Levenberg-Marquardt on the pixel reprojection error of the full camera model:
// intrinsics + the 5 distortion coefficients k1 k2 p1 p2 k3 of CvarCamera::distCoeffs, as cvFindExtrinsicCameraParams2
// is handed them, opencvar.cpp:261-272), written for a GPU lane: fixed-size arrays, Newton iteration for the nearest
// rotation (no SVD), Cholesky for the 6x6 normal equations.  With all coefficients zero (cvarReadCamera(NULL)) the
// distortion terms are skipped altogether.  The contract is the converged minimum in the same basin (tolerance 1e-4
// relative on glMatrix).
#pragma once
#include "hd.h"
#include <float.h>
#include <math.h>

namespace ocvar {

struct CameraRec {  // == CvarCamera (include/opencvar/opencvar.h), 248 bytes
    int width, height;
    double cameraMatrix[9];
    double distCoeffs[5];
    double glProjection[16];
};

template <bool WITH_J>
OCVAR_HD void rodrigues_t(const double* r, double* R, double* J /* 3x9 when WITH_J */) {
    double rx = r[0], ry = r[1], rz = r[2];
    const double theta = sqrt(rx * rx + ry * ry + rz * rz);
    OCVAR_UNROLL
    for (int k = 0; k < 9; k++) R[k] = (k % 4 == 0) ? 1.0 : 0.0;
    if (theta < DBL_EPSILON) {
        if (WITH_J) {
            OCVAR_UNROLL
            for (int k = 0; k < 27; k++) J[k] = 0;
            J[5] = J[15] = J[19] = -1;
            J[7] = J[11] = J[21] = 1;
        }
        return;
    }
    const double c = cos(theta), s = sin(theta), c1 = 1. - c, it = 1. / theta;
    rx *= it;
    ry *= it;
    rz *= it;
    const double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
    const double rx_[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
    OCVAR_UNROLL
    for (int k = 0; k < 9; k++) R[k] = c * R[k] + c1 * rrt[k] + s * rx_[k];
    if (WITH_J) {
        const double drrt[27] = {rx + rx, ry, rz, ry, 0, 0, rz, 0, 0, 0, rx, 0, rx, ry + ry, rz, 0, rz, 0,
                                 0, 0, rx, 0, 0, ry, rx, ry, rz + rz};
        const double drx[27] = {0, 0, 0, 0, 0, -1, 0, 1, 0, 0, 0, 1, 0, 0, 0, -1, 0, 0, 0, -1, 0, 1, 0, 0, 0, 0, 0};
        OCVAR_UNROLL
        for (int i = 0; i < 3; i++) {
            const double ri = i == 0 ? rx : (i == 1 ? ry : rz);
            const double a0 = -s * ri, a1 = (s - 2 * c1 * it) * ri, a2 = c1 * it, a3 = (c - s * it) * ri, a4 = s * it;
            OCVAR_UNROLL
            for (int k = 0; k < 9; k++)
                J[i * 9 + k] = a0 * ((k % 4 == 0) ? 1.0 : 0.0) + a1 * rrt[k] + a2 * drrt[i * 9 + k] + a3 * rx_[k] + a4 * drx[i * 9 + k];
        }
    }
}

OCVAR_HD void rodrigues(const double* r, double* R, double* J /* 3x9 or null */) {
    if (J) rodrigues_t<true>(r, R, J);
    else rodrigues_t<false>(r, R, nullptr);
}

// nearest rotation (orthogonal polar factor) by Newton's iteration Q <- (Q + Q^-T)/2
OCVAR_HD void nearest_rotation(double* Q) {
    for (int it = 0; it < 30; it++) {
        double C[9];
        C[0] = Q[4] * Q[8] - Q[5] * Q[7];
        C[1] = Q[5] * Q[6] - Q[3] * Q[8];
        C[2] = Q[3] * Q[7] - Q[4] * Q[6];
        C[3] = Q[2] * Q[7] - Q[1] * Q[8];
        C[4] = Q[0] * Q[8] - Q[2] * Q[6];
        C[5] = Q[1] * Q[6] - Q[0] * Q[7];
        C[6] = Q[1] * Q[5] - Q[2] * Q[4];
        C[7] = Q[2] * Q[3] - Q[0] * Q[5];
        C[8] = Q[0] * Q[4] - Q[1] * Q[3];
        const double det = Q[0] * C[0] + Q[1] * C[1] + Q[2] * C[2];
        if (det == 0) return;
        double delta = 0;
        OCVAR_UNROLL
        for (int k = 0; k < 9; k++) {
            const double n = 0.5 * (Q[k] + C[k] / det);  // cofactor/det = inverse transpose
            delta += (n - Q[k]) * (n - Q[k]);
            Q[k] = n;
        }
        if (delta < 1e-30) break;
    }
}

OCVAR_HD void rotation_to_rvec(const double* Rin, double* r) {
    double R[9];
    OCVAR_UNROLL
    for (int k = 0; k < 9; k++) R[k] = Rin[k];
    nearest_rotation(R);
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    const double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1) * 0.5;
    c = c > 1. ? 1. : (c < -1. ? -1. : c);
    double theta = acos(c);
    if (s < 1e-5) {
        if (c > 0) {
            rx = ry = rz = 0;
        } else {
            double t = (R[0] + 1) * 0.5;
            rx = sqrt(t > 0 ? t : 0);
            t = (R[4] + 1) * 0.5;
            ry = sqrt(t > 0 ? t : 0) * (R[1] < 0 ? -1. : 1.);
            t = (R[8] + 1) * 0.5;
            rz = sqrt(t > 0 ? t : 0) * (R[2] < 0 ? -1. : 1.);
            if (fabs(rx) < fabs(ry) && fabs(rx) < fabs(rz) && (R[5] > 0) != (ry * rz > 0)) rz = -rz;
            theta /= sqrt(rx * rx + ry * ry + rz * rz);
            rx *= theta;
            ry *= theta;
            rz *= theta;
        }
    } else {
        const double v = theta / (2 * s);
        rx *= v;
        ry *= v;
        rz *= v;
    }
    r[0] = rx;
    r[1] = ry;
    r[2] = rz;
}

// Homography of the marker rectangle (+-ratio, +-1) -> 4 normalised image points.  Four points determine it, so it is written
// down in closed form (the projective map of the unit square onto a quadrilateral, composed with the rectangle's affine map onto
// the unit square) instead of eliminating an 8x8 system: no pivot search, no indexed rows -- on the GPU the system's 72 doubles
// lived in scratch memory and every access was a memory round trip.  Scaled to H[8] = 1 like the solve it replaces; it only
// seeds the refinement below, which converges to the same minimum (tests/test_device_cores_cpu.py against the oracle's solve).
OCVAR_HD bool rect_homography(double ratio, const double* m, double* H) {
    const double x0 = m[0], y0 = m[1], x1 = m[2], y1 = m[3], x2 = m[4], y2 = m[5], x3 = m[6], y3 = m[7];
    // unit square (0,0) (1,0) (1,1) (0,1) -> p0 p1 p2 p3
    const double dx1 = x1 - x2, dy1 = y1 - y2, dx2 = x3 - x2, dy2 = y3 - y2, sx = x0 - x1 + x2 - x3, sy = y0 - y1 + y2 - y3;
    const double den = dx1 * dy2 - dx2 * dy1;
    if (den == 0) return false;
    const double g = (sx * dy2 - dx2 * sy) / den, h = (dx1 * sy - sx * dy1) / den;
    const double a = x1 - x0 + g * x1, b = x3 - x0 + h * x3, c = x0;
    const double d = y1 - y0 + g * y1, e = y3 - y0 + h * y3, f = y0;
    // rectangle -> unit square: u = (X + ratio) / (2 ratio), v = (Y + 1) / 2
    const double iu = 0.5 / ratio, iv = 0.5;
    const double h0 = a * iu, h1 = b * iv, h2 = a * 0.5 + b * 0.5 + c;
    const double h3 = d * iu, h4 = e * iv, h5 = d * 0.5 + e * 0.5 + f;
    const double h6 = g * iu, h7 = h * iv, h8 = g * 0.5 + h * 0.5 + 1.0;
    if (h8 == 0) return false;
    const double s = 1.0 / h8;
    H[0] = h0 * s; H[1] = h1 * s; H[2] = h2 * s;
    H[3] = h3 * s; H[4] = h4 * s; H[5] = h5 * s;
    H[6] = h6 * s; H[7] = h7 * s; H[8] = 1;
    return true;
}

// reprojection of the 4 rectangle corners, residual e = proj - img, optional 8x6 Jacobian.  dist: k1 k2 p1 p2 k3 or null
// (pinhole).
template <bool WITH_J, bool DIST>
OCVAR_HD void reproject_t(double ratio, const double* p, double fx, double fy, double cx, double cy, const double* dist, const double* img, double* e, double* J) {
    double R[9], dR[27];
    rodrigues_t<WITH_J>(p, R, dR);
    const double MX[4] = {-ratio, ratio, ratio, -ratio}, MY[4] = {-1, -1, 1, 1};
    OCVAR_UNROLL
    for (int i = 0; i < 4; i++) {
        const double X = MX[i], Y = MY[i];
        double x = R[0] * X + R[1] * Y + p[3];
        double y = R[3] * X + R[4] * Y + p[4];
        double z = R[6] * X + R[7] * Y + p[5];
        z = z ? 1. / z : 1;
        x *= z;
        y *= z;
        if (!DIST) {
            e[2 * i] = x * fx + cx - img[2 * i];
            e[2 * i + 1] = y * fy + cy - img[2 * i + 1];
            if (WITH_J) {
                double* jx = J + 12 * i;
                double* jy = jx + 6;
                OCVAR_UNROLL
                for (int j = 0; j < 3; j++) {
                    const double* d = dR + 9 * j;
                    const double dx0 = X * d[0] + Y * d[1], dy0 = X * d[3] + Y * d[4], dz0 = X * d[6] + Y * d[7];
                    jx[j] = fx * z * (dx0 - x * dz0);
                    jy[j] = fy * z * (dy0 - y * dz0);
                }
                jx[3] = fx * z; jx[4] = 0; jx[5] = -fx * x * z;
                jy[3] = 0; jy[4] = fy * z; jy[5] = -fy * y * z;
            }
            continue;
        }
        // cvProjectPoints2: radial (k1 k2 k3) and tangential (p1 p2) distortion of the normalised point
        const double k1 = dist[0], k2 = dist[1], p1 = dist[2], p2 = dist[3], k3 = dist[4];
        const double r2 = x * x + y * y;
        const double cd = 1 + (k1 + (k2 + k3 * r2) * r2) * r2;
        const double xd = x * cd + 2 * p1 * x * y + p2 * (r2 + 2 * x * x);
        const double yd = y * cd + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y;
        e[2 * i] = xd * fx + cx - img[2 * i];
        e[2 * i + 1] = yd * fy + cy - img[2 * i + 1];
        if (WITH_J) {
            double* jx = J + 12 * i;
            double* jy = jx + 6;
            // partial derivatives of the distorted point with respect to the undistorted one
            const double dcd = 2 * (k1 + (2 * k2 + 3 * k3 * r2) * r2);   // d(cd)/d(r2) * 2: d(cd) = dcd * (x dx + y dy)
            const double xdx = cd + x * x * dcd + 2 * p1 * y + 6 * p2 * x, xdy = x * y * dcd + 2 * p1 * x + 2 * p2 * y;
            const double ydx = x * y * dcd + 2 * p1 * x + 2 * p2 * y, ydy = cd + y * y * dcd + 6 * p1 * y + 2 * p2 * x;
            OCVAR_UNROLL
            for (int j = 0; j < 6; j++) {
                double dx, dy;
                if (j < 3) {
                    const double* d = dR + 9 * j;
                    const double dx0 = X * d[0] + Y * d[1], dy0 = X * d[3] + Y * d[4], dz0 = X * d[6] + Y * d[7];
                    dx = z * (dx0 - x * dz0);
                    dy = z * (dy0 - y * dz0);
                } else {
                    dx = j == 3 ? z : (j == 5 ? -x * z : 0.0);
                    dy = j == 4 ? z : (j == 5 ? -y * z : 0.0);
                }
                jx[j] = fx * (xdx * dx + xdy * dy);
                jy[j] = fy * (ydx * dx + ydy * dy);
            }
        }
    }
}

OCVAR_HD void reproject(double ratio, const double* p, const double* K, const double* dist, const double* img, double* e, double* J) {
    if (dist) {
        if (J) reproject_t<true, true>(ratio, p, K[0], K[4], K[2], K[5], dist, img, e, J);
        else reproject_t<false, true>(ratio, p, K[0], K[4], K[2], K[5], dist, img, e, nullptr);
    } else {
        if (J) reproject_t<true, false>(ratio, p, K[0], K[4], K[2], K[5], nullptr, img, e, J);
        else reproject_t<false, false>(ratio, p, K[0], K[4], K[2], K[5], nullptr, img, e, nullptr);
    }
}

// solves A x = b for symmetric positive definite 6x6 A (Cholesky); false if not positive definite
OCVAR_HD bool chol6(const double* A, const double* b, double* x) {
    double L[36];
    OCVAR_UNROLL
    for (int i = 0; i < 6; i++)
        OCVAR_UNROLL
        for (int j = 0; j <= i; j++) {
            double s = A[i * 6 + j];
            OCVAR_UNROLL
            for (int k = 0; k < j; k++) s -= L[i * 6 + k] * L[j * 6 + k];
            if (i == j) {
                if (!(s > 0)) return false;
                L[i * 6 + i] = sqrt(s);
            } else {
                L[i * 6 + j] = s / L[j * 6 + j];
            }
        }
    double y[6];
    OCVAR_UNROLL
    for (int i = 0; i < 6; i++) {
        double s = b[i];
        OCVAR_UNROLL
        for (int k = 0; k < i; k++) s -= L[i * 6 + k] * y[k];
        y[i] = s / L[i * 6 + i];
    }
    OCVAR_UNROLL
    for (int i = 5; i >= 0; i--) {
        double s = y[i];
        OCVAR_UNROLL
        for (int k = i + 1; k < 6; k++) s -= L[k * 6 + i] * x[k];
        x[i] = s / L[i * 6 + i];
    }
    return true;
}

OCVAR_HD double norm_n(const double* v, int n) {
    double s = 0;
    OCVAR_UNROLL
    for (int i = 0; i < n; i++) s += v[i] * v[i];
    return sqrt(s);
}

OCVAR_HD void gl_from_pose(const double* R, const double* t, double* m) {
    // cvarGlMatrix: m = R^T laid out in 4x4, through a quaternion with x,y negated, then translation.
    OCVAR_UNROLL
    for (int k = 0; k < 16; k++) m[k] = 0;
    OCVAR_UNROLL
    for (int j = 0; j < 3; j++)
        OCVAR_UNROLL
        for (int i = 0; i < 3; i++) m[i * 4 + j] = R[j * 3 + i];
    double x, y, z, w, s;
    const double tr = 1 + m[0] + m[5] + m[10];
    if (tr > 0.00000001) {
        s = sqrt(tr) * 2;
        x = (m[9] - m[6]) / s;
        y = (m[2] - m[8]) / s;
        z = (m[4] - m[1]) / s;
        w = 0.25 * s;
    } else if (m[0] > m[5] && m[0] > m[10]) {
        s = sqrt(1 + m[0] - m[5] - m[10]) * 2;
        x = 0.25 * s;
        y = (m[4] + m[1]) / s;
        z = (m[2] + m[8]) / s;
        w = (m[9] - m[6]) / s;
    } else if (m[5] > m[10]) {
        s = sqrt(1 + m[5] - m[0] - m[10]) * 2;
        x = (m[4] + m[1]) / s;
        y = 0.25 * s;
        z = (m[9] + m[6]) / s;
        w = (m[2] - m[8]) / s;
    } else {
        s = sqrt(1 + m[10] - m[0] - m[5]) * 2;
        x = (m[2] + m[8]) / s;
        y = (m[9] + m[6]) / s;
        z = 0.25 * s;
        w = (m[4] - m[1]) / s;
    }
    x = -x;
    y = -y;
    const double xx = x * x, xy = x * y, xz = x * z, xw = x * w, yy = y * y, yz = y * z, yw = y * w, zz = z * z, zw = z * w;
    m[0] = 1 - 2 * (yy + zz);
    m[1] = 2 * (xy - zw);
    m[2] = 2 * (xz + yw);
    m[4] = 2 * (xy + zw);
    m[5] = 1 - 2 * (xx + zz);
    m[6] = 2 * (yz - xw);
    m[8] = 2 * (xz - yw);
    m[9] = 2 * (yz + xw);
    m[10] = 1 - 2 * (xx + yy);
    m[12] = t[0];
    m[13] = t[1];
    m[14] = -t[2];
    m[15] = 1;
}

// fx, fy, cx, cy: the intrinsics; dist: k1 k2 p1 p2 k3 (read only when DIST)
template <bool DIST>
OCVAR_HD void square_to_glmatrix_t(const float* sq, double fx, double fy, double cx, double cy, const double* dist, double ratio, double* gl) {
    const double K[6] = {fx, 0, cx, 0, fy, cy};
    double img[8], mn[8];
    OCVAR_UNROLL
    for (int i = 0; i < 4; i++) {
        img[2 * i] = sq[2 * i];
        img[2 * i + 1] = sq[2 * i + 1];
        double x = (img[2 * i] - K[2]) * (1. / K[0]), y = (img[2 * i + 1] - K[5]) * (1. / K[4]);
        if (DIST) {   // cvUndistortPoints: 5 fixed-point iterations of the inverse distortion
            const double x0 = x, y0 = y;
            for (int it = 0; it < 5; it++) {
                const double r2 = x * x + y * y;
                const double icd = 1. / (1 + ((dist[4] * r2 + dist[1]) * r2 + dist[0]) * r2);
                const double ddx = 2 * dist[2] * x * y + dist[3] * (r2 + 2 * x * x);
                const double ddy = dist[2] * (r2 + 2 * y * y) + 2 * dist[3] * x * y;
                x = (x0 - ddx) * icd;
                y = (y0 - ddy) * icd;
            }
        }
        mn[2 * i] = x;
        mn[2 * i + 1] = y;
    }
    double p[6] = {0, 0, 0, 0, 0, 0};
    double h[9];
    if (rect_homography(ratio, mn, h)) {
        const double n1 = sqrt(h[0] * h[0] + h[3] * h[3] + h[6] * h[6]), n2 = sqrt(h[1] * h[1] + h[4] * h[4] + h[7] * h[7]);
        const double s1 = 1. / fmax(n1, DBL_EPSILON), s2 = 1. / fmax(n2, DBL_EPSILON), s3 = 2. / fmax(n1 + n2, DBL_EPSILON);
        const double a0 = h[0] * s1, a1 = h[3] * s1, a2 = h[6] * s1, b0 = h[1] * s2, b1 = h[4] * s2, b2 = h[7] * s2;
        double Rm[9] = {a0, b0, a1 * b2 - a2 * b1, a1, b1, a2 * b0 - a0 * b2, a2, b2, a0 * b1 - a1 * b0};
        // Rodrigues round trip (orthonormalise), then matrix -> vector once more as OpenCV does
        double rv[3];
        rotation_to_rvec(Rm, rv);
        rodrigues_t<false>(rv, Rm, nullptr);
        rotation_to_rvec(Rm, p);
        p[3] = h[2] * s3;
        p[4] = h[5] * s3;
        p[5] = h[8] * s3;
    }
    // Levenberg-Marquardt as CvLevMarq(6, 8, 20 iterations | FLT_EPSILON relative step)
    double prev[6], J[48], e[8], JtJ[36], Jte[6], N[36], dx[6];
    int lambdaLg10 = -3;
    double prevErr = DBL_MAX;
    for (int iters = 0;;) {
        reproject_t<true, DIST>(ratio, p, fx, fy, cx, cy, dist, img, e, J);
        OCVAR_UNROLL
        for (int i = 0; i < 6; i++) {
            OCVAR_UNROLL
            for (int j = 0; j < 6; j++) {
                double s = 0;
                OCVAR_UNROLL
                for (int k = 0; k < 8; k++) s += J[k * 6 + i] * J[k * 6 + j];
                JtJ[i * 6 + j] = s;
            }
            double s = 0;
            OCVAR_UNROLL
            for (int k = 0; k < 8; k++) s += J[k * 6 + i] * e[k];
            Jte[i] = s;
        }
        OCVAR_UNROLL
        for (int i = 0; i < 6; i++) prev[i] = p[i];
        if (iters == 0) prevErr = norm_n(e, 8);
        double errNorm;
        for (bool first = true;; first = false) {
            if (!first) {
                reproject_t<false, DIST>(ratio, p, fx, fy, cx, cy, dist, img, e, nullptr);
                errNorm = norm_n(e, 8);
                if (!(errNorm > prevErr && ++lambdaLg10 <= 16)) break;
            }
            const double lambda = exp(lambdaLg10 * 2.302585092994046);
            OCVAR_UNROLL
            for (int k = 0; k < 36; k++) N[k] = JtJ[k];
            OCVAR_UNROLL
            for (int i = 0; i < 6; i++) N[i * 7] *= 1. + lambda;
            if (!chol6(N, Jte, dx))
                OCVAR_UNROLL
                for (int i = 0; i < 6; i++) dx[i] = 0;
            OCVAR_UNROLL
            for (int i = 0; i < 6; i++) p[i] = prev[i] - dx[i];
        }
        lambdaLg10 = lambdaLg10 - 1 > -16 ? lambdaLg10 - 1 : -16;
        double d[6];
        OCVAR_UNROLL
        for (int i = 0; i < 6; i++) d[i] = p[i] - prev[i];
        if (++iters >= 20 || norm_n(d, 6) / norm_n(prev, 6) < FLT_EPSILON) break;
        prevErr = errNorm;
    }
    double R[9];
    rodrigues_t<false>(p, R, nullptr);
    gl_from_pose(R, p + 3, gl);
}

OCVAR_HD void square_to_glmatrix(const float* sq, const CameraRec& cam, double ratio, double* gl) {
    const double fx = cam.cameraMatrix[0], fy = cam.cameraMatrix[4], cx = cam.cameraMatrix[2], cy = cam.cameraMatrix[5];
    const double kd[5] = {cam.distCoeffs[0], cam.distCoeffs[1], cam.distCoeffs[2], cam.distCoeffs[3], cam.distCoeffs[4]};
    if (kd[0] != 0 || kd[1] != 0 || kd[2] != 0 || kd[3] != 0 || kd[4] != 0) square_to_glmatrix_t<true>(sq, fx, fy, cx, cy, kd, ratio, gl);
    else square_to_glmatrix_t<false>(sq, fx, fy, cx, cy, kd, ratio, gl);
}

}  // namespace ocvar
