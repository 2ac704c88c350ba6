// orc_pose.cpp -- oracle: planar pose of a quad (TEST INFRASTRUCTURE, see oracle.h).
//
// Restates: opencvar.cpp:524-540 cvarSquareToMatrix, 229-245 cvarSquareInit, 261-278 cvarFindCamera,
// 133-152 cvarGlMatrix, and the OpenCV 2.4.x algorithms behind cvFindExtrinsicCameraParams2 +
// cvRodrigues2: normalised image points, planar branch (4-point homography ->
// h1,h2 normalised, r3 = r1 x r2, orthonormalised through a Rodrigues round trip), then CvLevMarq
// (lambda 1e-3, diag*(1+lambda), <= 20 iterations or relative step < FLT_EPSILON) on the pixel
// reprojection error with analytic Jacobians.  Parity unpinned at the OpenCV boundary.
#include "oracle.h"
#include <cfloat>
#include <cmath>
#include <cstring>

namespace {

// cyclic Jacobi eigen-decomposition of a symmetric n x n matrix (n <= 9): A = V diag(w) V^T
void jacobi_eig(int n, const double* Ain, double* w, double* V) {
    double A[81];
    memcpy(A, Ain, sizeof(double) * n * n);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) V[i * n + j] = (i == j);
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0;
        for (int i = 0; i < n; i++)
            for (int j = i + 1; j < n; j++) off += A[i * n + j] * A[i * n + j];
        if (off < 1e-300) break;
        for (int p = 0; p < n; p++)
            for (int q = p + 1; q < n; q++) {
                double apq = A[p * n + q];
                if (fabs(apq) < 1e-300) continue;
                double theta = (A[q * n + q] - A[p * n + p]) / (2 * apq);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
                double c = 1 / sqrt(t * t + 1), s = t * c;
                for (int k = 0; k < n; k++) {
                    double akp = A[k * n + p], akq = A[k * n + q];
                    A[k * n + p] = c * akp - s * akq;
                    A[k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; k++) {
                    double apk = A[p * n + k], aqk = A[q * n + k];
                    A[p * n + k] = c * apk - s * aqk;
                    A[q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; k++) {
                    double vkp = V[k * n + p], vkq = V[k * n + q];
                    V[k * n + p] = c * vkp - s * vkq;
                    V[k * n + q] = s * vkp + c * vkq;
                }
            }
    }
    for (int i = 0; i < n; i++) w[i] = A[i * n + i];
}

// x = pinv(A) b for symmetric PSD A (the cvSVD + cvSVBkSb pair of CvLevMarq::step)
void solve_sym(int n, const double* A, const double* b, double* x) {
    double w[9], V[81];
    jacobi_eig(n, A, w, V);
    double sum = 0;
    for (int i = 0; i < n; i++) sum += fabs(w[i]);
    double thr = 2 * DBL_EPSILON * sum;
    for (int i = 0; i < n; i++) x[i] = 0;
    for (int k = 0; k < n; k++) {
        if (fabs(w[k]) <= thr) continue;
        double d = 0;
        for (int i = 0; i < n; i++) d += V[i * n + k] * b[i];
        d /= w[k];
        for (int i = 0; i < n; i++) x[i] += V[i * n + k] * d;
    }
}

void mat3_mul(const double* A, const double* B, double* C) {
    double t[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) t[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
    memcpy(C, t, sizeof t);
}

// orthogonal polar factor U V^T of a 3x3 matrix (what cvRodrigues2 does with cvSVD)
void polar3(const double* A, double* Q) {
    double AtA[9], w[3], V[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) AtA[i * 3 + j] = A[i] * A[j] + A[3 + i] * A[3 + j] + A[6 + i] * A[6 + j];
    jacobi_eig(3, AtA, w, V);
    double S[9];  // (A^T A)^(-1/2) = V diag(1/sqrt(w)) V^T
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += V[i * 3 + k] * V[j * 3 + k] / sqrt(w[k] > 1e-300 ? w[k] : 1e-300);
            S[i * 3 + j] = s;
        }
    mat3_mul(A, S, Q);
}

// 4-point homography (plane -> normalised image), exact solve of the 8x8 DLT system.
bool homography4(const double* M, const double* m, double* H) {
    double A[8][9];
    for (int i = 0; i < 4; i++) {
        double X = M[2 * i], Y = M[2 * i + 1], x = m[2 * i], y = m[2 * i + 1];
        double r0[9] = {X, Y, 1, 0, 0, 0, -x * X, -x * Y, x};
        double r1[9] = {0, 0, 0, X, Y, 1, -y * X, -y * Y, y};
        memcpy(A[2 * i], r0, sizeof r0);
        memcpy(A[2 * i + 1], r1, sizeof r1);
    }
    for (int c = 0; c < 8; c++) {
        int piv = c;
        for (int r = c + 1; r < 8; r++)
            if (fabs(A[r][c]) > fabs(A[piv][c])) piv = r;
        if (A[piv][c] == 0) return false;
        for (int k = 0; k < 9; k++) {
            double t = A[c][k];
            A[c][k] = A[piv][k];
            A[piv][k] = t;
        }
        for (int r = 0; r < 8; r++) {
            if (r == c) continue;
            double f = A[r][c] / A[c][c];
            for (int k = c; k < 9; k++) A[r][k] -= f * A[c][k];
        }
    }
    for (int i = 0; i < 8; i++) H[i] = A[i][8] / A[i][i];
    H[8] = 1;
    return true;
}

// cvProjectPoints2 with the 5-coefficient distortion model k = (k1, k2, p1, p2, k3) and its Jacobian with respect to
// (rvec, tvec) (OpenCV 2.4 calibration.cpp; dist == nullptr or all zero: the pinhole projection)
void project(const double* obj, const double* r, const double* t, const double* K, const double* dist, double* proj,
             double* J /*8x6 or null*/) {
    double R[9], dRdr[27];
    orc_rodrigues_vec2mat(r, R, dRdr);
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    double k[5] = {0, 0, 0, 0, 0};
    if (dist)
        for (int i = 0; i < 5; i++) k[i] = dist[i];
    for (int i = 0; i < 4; i++) {
        const double X = obj[3 * i], Y = obj[3 * i + 1], Z = obj[3 * i + 2];
        double x = R[0] * X + R[1] * Y + R[2] * Z + t[0];
        double y = R[3] * X + R[4] * Y + R[5] * Z + t[1];
        double z = R[6] * X + R[7] * Y + R[8] * Z + t[2];
        z = z ? 1. / z : 1;
        x *= z;
        y *= z;
        const double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
        const double a1 = 2 * x * y, a2 = r2 + 2 * x * x, a3 = r2 + 2 * y * y;
        const double cdist = 1 + k[0] * r2 + k[1] * r4 + k[4] * r6;
        const double xd = x * cdist + k[2] * a1 + k[3] * a2;
        const double yd = y * cdist + k[2] * a3 + k[3] * a1;
        proj[2 * i] = xd * fx + cx;
        proj[2 * i + 1] = yd * fy + cy;
        if (J) {
            double* jx = J + (2 * i) * 6;
            double* jy = J + (2 * i + 1) * 6;
            double dxp[6], dyp[6];   // d(x)/d(param), d(y)/d(param) of the undistorted normalised point
            for (int j = 0; j < 3; j++) {
                const double* d = dRdr + j * 9;
                const double dx0 = X * d[0] + Y * d[1] + Z * d[2];
                const double dy0 = X * d[3] + Y * d[4] + Z * d[5];
                const double dz0 = X * d[6] + Y * d[7] + Z * d[8];
                dxp[j] = z * (dx0 - x * dz0);
                dyp[j] = z * (dy0 - y * dz0);
            }
            dxp[3] = z; dxp[4] = 0; dxp[5] = -x * z;
            dyp[3] = 0; dyp[4] = z; dyp[5] = -y * z;
            for (int j = 0; j < 6; j++) {
                const double dr2 = 2 * x * dxp[j] + 2 * y * dyp[j];
                const double dcdist = k[0] * dr2 + 2 * k[1] * r2 * dr2 + 3 * k[4] * r4 * dr2;
                const double da1 = 2 * (x * dyp[j] + y * dxp[j]);
                jx[j] = fx * (dxp[j] * cdist + x * dcdist + k[2] * da1 + k[3] * (dr2 + 4 * x * dxp[j]));
                jy[j] = fy * (dyp[j] * cdist + y * dcdist + k[2] * (dr2 + 4 * y * dyp[j]) + k[3] * da1);
            }
        }
    }
}

double norm_n(const double* v, int n) {
    double s = 0;
    for (int i = 0; i < n; i++) s += v[i] * v[i];
    return sqrt(s);
}

}  // namespace

extern "C" void orc_rodrigues_vec2mat(const double* r, double* R, double* J) {
    double rx = r[0], ry = r[1], rz = r[2];
    double theta = sqrt(rx * rx + ry * ry + rz * rz);
    if (theta < DBL_EPSILON) {
        double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        memcpy(R, I, sizeof I);
        if (J) {
            memset(J, 0, sizeof(double) * 27);
            J[5] = J[15] = J[19] = -1;
            J[7] = J[11] = J[21] = 1;
        }
        return;
    }
    const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    double c = cos(theta), s = sin(theta), c1 = 1. - c, itheta = theta ? 1. / theta : 0.;
    rx *= itheta;
    ry *= itheta;
    rz *= itheta;
    double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
    double r_x[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
    for (int k = 0; k < 9; k++) R[k] = c * I[k] + c1 * rrt[k] + s * r_x[k];
    if (J) {
        double drrt[27] = {rx + rx, ry, rz, ry, 0, 0, rz, 0, 0, 0, rx, 0, rx, ry + ry, rz, 0, rz, 0,
                           0, 0, rx, 0, 0, ry, rx, ry, rz + rz};
        double d_r_x_[27] = {0, 0, 0, 0, 0, -1, 0, 1, 0, 0, 0, 1, 0, 0, 0, -1, 0, 0, 0, -1, 0, 1, 0, 0, 0, 0, 0};
        for (int i = 0; i < 3; i++) {
            double ri = i == 0 ? rx : i == 1 ? ry : rz;
            double a0 = -s * ri, a1 = (s - 2 * c1 * itheta) * ri, a2 = c1 * itheta;
            double a3 = (c - s * itheta) * ri, a4 = s * itheta;
            for (int k = 0; k < 9; k++)
                J[i * 9 + k] = a0 * I[k] + a1 * rrt[k] + a2 * drrt[i * 9 + k] + a3 * r_x[k] + a4 * d_r_x_[i * 9 + k];
        }
    }
}

extern "C" void orc_rodrigues_mat2vec(const double* Rin, double* r) {
    double R[9];
    polar3(Rin, R);
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1) * 0.5;
    c = c > 1. ? 1. : c < -1. ? -1. : c;
    double theta = acos(c);
    if (s < 1e-5) {
        if (c > 0)
            rx = ry = rz = 0;
        else {
            double t;
            t = (R[0] + 1) * 0.5;
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
        double vth = 1 / (2 * s);
        vth *= theta;
        rx *= vth;
        ry *= vth;
        rz *= vth;
    }
    r[0] = rx;
    r[1] = ry;
    r[2] = rz;
}

extern "C" void orc_find_extrinsic(const double* obj, const double* img, const double* K, const double* dist, double* rvec, double* tvec) {
    double param[6] = {0, 0, 0, 0, 0, 0};
    // normalised image points: cvUndistortPoints, 5 fixed-point iterations of the inverse distortion (the identity for
    // zero coefficients)
    double k[5] = {0, 0, 0, 0, 0};
    if (dist)
        for (int i = 0; i < 5; i++) k[i] = dist[i];
    double mn[8], Mxy[8];
    for (int i = 0; i < 4; i++) {
        double x = (img[2 * i] - K[2]) * (1. / K[0]), y = (img[2 * i + 1] - K[5]) * (1. / K[4]);
        const double x0 = x, y0 = y;
        for (int j = 0; j < 5; j++) {
            const double r2 = x * x + y * y;
            const double icdist = 1. / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
            const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x);
            const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y;
            x = (x0 - deltaX) * icdist;
            y = (y0 - deltaY) * icdist;
        }
        mn[2 * i] = x;
        mn[2 * i + 1] = y;
        Mxy[2 * i] = obj[3 * i];  // object plane is z = 0: R_transform = I, T_transform = -centroid = 0
        Mxy[2 * i + 1] = obj[3 * i + 1];
    }
    double cxo = 0, cyo = 0;
    for (int i = 0; i < 4; i++) {
        cxo += obj[3 * i];
        cyo += obj[3 * i + 1];
    }
    cxo /= 4;
    cyo /= 4;
    double tt[3] = {-cxo, -cyo, 0};
    for (int i = 0; i < 4; i++) {
        Mxy[2 * i] += tt[0];
        Mxy[2 * i + 1] += tt[1];
    }
    double h[9], Rm[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, t[3] = {0, 0, 0};
    if (homography4(Mxy, mn, h)) {
        double h1n = sqrt(h[0] * h[0] + h[3] * h[3] + h[6] * h[6]);
        double h2n = sqrt(h[1] * h[1] + h[4] * h[4] + h[7] * h[7]);
        double s1 = 1. / fmax(h1n, DBL_EPSILON), s2 = 1. / fmax(h2n, DBL_EPSILON), s3 = 2. / fmax(h1n + h2n, DBL_EPSILON);
        double a[3] = {h[0] * s1, h[3] * s1, h[6] * s1}, b[3] = {h[1] * s2, h[4] * s2, h[7] * s2};
        t[0] = h[2] * s3;
        t[1] = h[5] * s3;
        t[2] = h[8] * s3;
        double cr[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
        double Hm[9] = {a[0], b[0], cr[0], a[1], b[1], cr[1], a[2], b[2], cr[2]};
        double rv[3];
        orc_rodrigues_mat2vec(Hm, rv);
        orc_rodrigues_vec2mat(rv, Hm, nullptr);
        for (int i = 0; i < 3; i++) t[i] = Hm[i * 3] * tt[0] + Hm[i * 3 + 1] * tt[1] + Hm[i * 3 + 2] * tt[2] + t[i];
        memcpy(Rm, Hm, sizeof Rm);
    }
    orc_rodrigues_mat2vec(Rm, param);
    param[3] = t[0];
    param[4] = t[1];
    param[5] = t[2];

    // CvLevMarq(6, 8, (ITER+EPS, 20, FLT_EPSILON), completeSymm)
    double prev[6], J[48], err[8], proj[8], JtJ[36], JtErr[6], JtJN[36], dx[6];
    int lambdaLg10 = -3, iters = 0;
    double prevErrNorm = DBL_MAX;
    for (;;) {
        project(obj, param, param + 3, K, dist, proj, J);
        for (int i = 0; i < 8; i++) err[i] = proj[i] - img[i];
        for (int i = 0; i < 6; i++) {
            for (int j = 0; j < 6; j++) {
                double s = 0;
                for (int k = 0; k < 8; k++) s += J[k * 6 + i] * J[k * 6 + j];
                JtJ[i * 6 + j] = s;
            }
            double s = 0;
            for (int k = 0; k < 8; k++) s += J[k * 6 + i] * err[k];
            JtErr[i] = s;
        }
        memcpy(prev, param, sizeof prev);
        auto step = [&]() {
            double lambda = exp(lambdaLg10 * log(10.));
            memcpy(JtJN, JtJ, sizeof JtJN);
            for (int i = 0; i < 6; i++) JtJN[i * 7] *= 1. + lambda;
            solve_sym(6, JtJN, JtErr, dx);
            for (int i = 0; i < 6; i++) param[i] = prev[i] - dx[i];
        };
        step();
        if (iters == 0) prevErrNorm = norm_n(err, 8);
        double errNorm;
        for (;;) {
            project(obj, param, param + 3, K, dist, proj, nullptr);
            for (int i = 0; i < 8; i++) err[i] = proj[i] - img[i];
            errNorm = norm_n(err, 8);
            if (errNorm > prevErrNorm) {
                if (++lambdaLg10 <= 16) {
                    step();
                    continue;
                }
            }
            break;
        }
        lambdaLg10 = lambdaLg10 - 1 > -16 ? lambdaLg10 - 1 : -16;
        double diff[6];
        for (int i = 0; i < 6; i++) diff[i] = param[i] - prev[i];
        if (++iters >= 20 || norm_n(diff, 6) / norm_n(prev, 6) < FLT_EPSILON) break;
        prevErrNorm = errNorm;
    }
    memcpy(rvec, param, 3 * sizeof(double));
    memcpy(tvec, param + 3, 3 * sizeof(double));
}

extern "C" void orc_gl_matrix(const double* R, const double* t, double* m) {
    // opencvar.cpp:133-152
    memset(m, 0, 16 * sizeof(double));
    for (int j = 0; j < 3; j++)
        for (int i = 0; i < 3; i++) m[i * 4 + j] = R[j * 3 + i];
    double q[4];
    orc_acMatrixToQuaternion(m, q);
    q[1] = -q[1];
    q[2] = -q[2];
    orc_acQuaternionToMatrix(q, m);
    m[12] = t[0];
    m[13] = t[1];
    m[14] = -t[2];
    m[15] = 1;
}

extern "C" void orc_square_to_matrix(const float* pts, const OrcCamera* cam, double ratio, double* m16) {
    double obj[12] = {-ratio, -1, 0, ratio, -1, 0, ratio, 1, 0, -ratio, 1, 0};  // opencvar.cpp:229-245
    double img[8];
    for (int i = 0; i < 8; i++) img[i] = pts[i];
    double rvec[3], tvec[3], R[9];
    orc_find_extrinsic(obj, img, cam->cameraMatrix, cam->distCoeffs, rvec, tvec);
    orc_rodrigues_vec2mat(rvec, R, nullptr);
    orc_gl_matrix(R, tvec, m16);
}
