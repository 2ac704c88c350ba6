// acmath_host.cpp -- host implementation of include/opencvar/acmath.h (API of the reference's
// /root/reference/src/acmath.cpp, written from its documented behaviour; tiny host-side helpers, no GPU work).
// Hot-path members (bit codec 486-580, quaternion pair 215-276, acCalcLength 293-298) are checked against the
// reference's compiled acmath.cpp in tests/test_host_mirror.py; the rest is kept for symbol compatibility.
#include "opencvar/acmath.h"
#include <cmath>
#include <cstdio>
#include <cstring>

static const double kPi = 3.1415927;  // the reference's AC_PI (acmath.cpp:35) -- behaviour-defining for deg<->rad

extern "C" {

void acVectorPrint(double* v) { std::printf("%g\t%g\t%g\t\n", v[0], v[1], v[2]); }
void acVectorAdd(double* a, double* b, double* o) { for (int i = 0; i < 3; i++) o[i] = a[i] + b[i]; }
void acVectorDeduct(double* a, double* b, double* o) { for (int i = 0; i < 3; i++) o[i] = a[i] - b[i]; }
void acVectorCrossProduct(double* a, double* b, double* p) {
    const double x = a[1] * b[2] - a[2] * b[1], y = -(b[2] * a[0] - b[0] * a[2]), z = a[0] * b[1] - a[1] * b[0];
    p[0] = x; p[1] = y; p[2] = z;
}
void acVectorNormal(double* v1, double* v2, double* v3, double* n) {
    double e1[3], e2[3];
    acVectorDeduct(v2, v1, e1);
    acVectorDeduct(v3, v1, e2);
    acVectorCrossProduct(e1, e2, n);
}
double acVectorMagnitude(double* v) { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }
void acVectorNormalise(double* in, double* out) {
    const double m = acVectorMagnitude(in);
    for (int i = 0; i < 3; i++) out[i] = in[i] / m;
}
void acVectorNormal2(double* v1, double* v2, double* v3, double* nv) {
    double n[3];
    acVectorNormal(v1, v2, v3, n);
    acVectorNormalise(n, nv);
}
double acRad2Deg(double rad) { return (rad * 180) / kPi; }
double acDeg2Rad(double deg) { return (deg * kPi) / 180; }

double acMatrixDotProduct(double* m1, double* m2, int col, int row) {
    double s = 0;
    for (int i = 0; i < 4; i++) s += m1[row * 4 + i] * m2[i * 4 + col];
    return s;
}
void acMatrixMultiply(double* m1, double* m2, double* out) {
    double t[16];
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) t[r * 4 + c] = acMatrixDotProduct(m1, m2, c, r);
    std::memcpy(out, t, sizeof t);
}
void acMatrixIdentity(double* m) {
    for (int i = 0; i < 16; i++) m[i] = (i % 5 == 0) ? 1.0 : 0.0;
}
void acMatrixRotate(double deg, double x, double y, double z, double* m) {
    const double c = std::cos(acDeg2Rad(deg)), s = std::sin(acDeg2Rad(deg)), k = 1 - c;
    const double mag = std::sqrt(x * x + y * y + z * z);
    x /= mag; y /= mag; z /= mag;
    double t[16] = {x * x * k + c,     y * x * k + z * s, x * z * k - y * s, 0,
                    x * y * k - z * s, y * y * k + c,     y * z * k + x * s, 0,
                    x * z * k + y * s, y * z * k - x * s, z * z * k + c,     0,
                    0, 0, 0, 1};
    acMatrixMultiply(t, m, m);
}
void acMatrixTranslate(double x, double y, double z, double* m) {
    double t[16];
    acMatrixIdentity(t);
    t[12] = x; t[13] = y; t[14] = z;  // OpenGL layout, like acMatrixRotate / acMatrixScale
    acMatrixMultiply(t, m, m);
}
void acMatrixScale(double x, double y, double z, double* m) {
    double t[16];
    acMatrixIdentity(t);
    t[0] = x; t[5] = y; t[10] = z;
    acMatrixMultiply(t, m, m);
}
void acMatrixPrint(double* m) {
    for (int r = 0; r < 4; r++) {
        for (int c = 0; c < 4; c++) std::printf("%f\t", m[r * 4 + c]);
        std::printf("\n");
    }
    std::printf("\n");
}
void acMatrixTranspose(double* m) {
    for (int r = 0; r < 4; r++)
        for (int c = r + 1; c < 4; c++) {
            const double t = m[r * 4 + c];
            m[r * 4 + c] = m[c * 4 + r];
            m[c * 4 + r] = t;
        }
}

void acMatrixToQuaternion(double* m, double* q) {
    double w, x, y, z, s;
    const double tr = 1 + m[0] + m[5] + m[10];
    if (tr > 0.00000001) {
        s = std::sqrt(tr) * 2;
        x = (m[9] - m[6]) / s; y = (m[2] - m[8]) / s; z = (m[4] - m[1]) / s; w = 0.25 * s;
    } else if (m[0] > m[5] && m[0] > m[10]) {
        s = std::sqrt(1 + m[0] - m[5] - m[10]) * 2;
        x = 0.25 * s; y = (m[4] + m[1]) / s; z = (m[2] + m[8]) / s; w = (m[9] - m[6]) / s;
    } else if (m[5] > m[10]) {
        s = std::sqrt(1 + m[5] - m[0] - m[10]) * 2;
        x = (m[4] + m[1]) / s; y = 0.25 * s; z = (m[9] + m[6]) / s; w = (m[2] - m[8]) / s;
    } else {
        s = std::sqrt(1 + m[10] - m[0] - m[5]) * 2;
        x = (m[2] + m[8]) / s; y = (m[9] + m[6]) / s; z = 0.25 * s; w = (m[4] - m[1]) / s;
    }
    q[0] = w; q[1] = x; q[2] = y; q[3] = z;
}
void acQuaternionToMatrix(double* q, double* m) {
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    m[0] = 1 - 2 * (y * y + z * z); m[1] = 2 * (x * y - z * w);     m[2] = 2 * (x * z + y * w);
    m[4] = 2 * (x * y + z * w);     m[5] = 1 - 2 * (x * x + z * z); m[6] = 2 * (y * z - x * w);
    m[8] = 2 * (x * z - y * w);     m[9] = 2 * (y * z + x * w);     m[10] = 1 - 2 * (x * x + y * y);
}

double acAngle(AcPointi* pt1, AcPointi* pt2, AcPointi* pt0) {
    const double dx1 = pt1->x - pt0->x, dy1 = pt1->y - pt0->y, dx2 = pt2->x - pt0->x, dy2 = pt2->y - pt0->y;
    return (dx1 * dx2 + dy1 * dy2) / std::sqrt((dx1 * dx1 + dy1 * dy1) * (dx2 * dx2 + dy2 * dy2) + 1e-10);
}
double acCalcLength(AcPointf a, AcPointf b) {
    const double dx = a.x - b.x, dy = a.y - b.y;
    return std::sqrt(dx * dx + dy * dy);
}

static double minor3(const double* m, int r, int c) {
    double a[9];
    int k = 0;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            if (i != r && j != c) a[k++] = m[i * 4 + j];
    return a[0] * (a[4] * a[8] - a[5] * a[7]) - a[1] * (a[3] * a[8] - a[5] * a[6]) + a[2] * (a[3] * a[7] - a[4] * a[6]);
}
double acMatrix4GetDeterminant(double m[]) {
    double d = 0;
    for (int c = 0; c < 4; c++) d += ((c & 1) ? -1.0 : 1.0) * m[c] * minor3(m, 0, c);
    return d;
}
void acMatrix4Invert(double m[]) {
    double adj[16];
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) adj[c * 4 + r] = (((r + c) & 1) ? -1.0 : 1.0) * minor3(m, r, c);
    const double inv = 1.0 / acMatrix4GetDeterminant(m);
    for (int i = 0; i < 16; i++) m[i] = inv * adj[i];
}
void acMatrixDecompose(double m[], double t[], double s[], double r[]) {
    for (int i = 0; i < 3; i++) {
        t[i] = m[i * 4 + 3];
        s[i] = std::sqrt(m[i * 4] * m[i * 4] + m[i * 4 + 1] * m[i * 4 + 1] + m[i * 4 + 2] * m[i * 4 + 2]);
    }
    acMatrixIdentity(r);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) r[i * 4 + j] = m[i * 4 + j] / s[i];
}

void acArray2DRotateub(unsigned char* arr, int w, int h, int rot) {
    unsigned char t[4096];
    if (w * h > (int)sizeof t) return;
    std::memcpy(t, arr, (size_t)w * h);
    for (int i = 0; i < h; i++)
        for (int j = 0; j < w; j++) {
            if (rot == 1) arr[i * w + j] = t[(h - 1 - j) * w + i];
            else if (rot == 2) arr[i * w + j] = t[(h - 1 - i) * w + (h - 1 - j)];
            else if (rot == 3) arr[i * w + j] = t[j * w + (h - 1 - i)];
        }
}
void acArray2DPrintub(unsigned char* arr, int w, int h) {
    for (int i = 0; i < h; i++) {
        for (int j = 0; j < w; j++) std::printf("%d ", arr[i * w + j]);
        std::printf("\n");
    }
}
void acArray2DToBit(unsigned char* arr, int w, int h, long long int* bit) {
    long long b = 0;
    for (int i = 0; i < h; i++)
        for (int j = w - 1; j >= 0; j--) b = (b << 1) | arr[i * w + j];
    *bit = b;
}
void acBitToArray2D(long long int bit, unsigned char* arr, int w, int h) {
    for (int i = h - 1; i >= 0; i--)
        for (int j = 0; j < w; j++) {
            arr[i * w + j] = (unsigned char)(bit & 1);
            bit >>= 1;
        }
}
void acBitRotate(long long int* bit, int rot, int w, int h) {
    unsigned char g[64];
    acBitToArray2D(*bit, g, w, h);
    acArray2DRotateub(g, w, h, rot);
    acArray2DToBit(g, w, h, bit);
}

}  // extern "C"
