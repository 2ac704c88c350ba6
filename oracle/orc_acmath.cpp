// orc_acmath.cpp -- oracle: the acmath subset on the hot path + template/camera setup
// (TEST INFRASTRUCTURE, see oracle.h).  PINNED: tests compare every function here against
// oracle/_ref/libacmath_ref.so, i.e. the reference's own src/acmath.cpp compiled in place.
//
// Follows: acmath.cpp:201-209 (transpose), 215-247 (matrix->quaternion), 253-276 (quaternion->matrix),
// 293-298 (length), 486-525 (grid rotation), 546-554 / 559-566 (bit codec), 575-580 (bit rotate);
// opencvar.cpp:284-321 (template load), 39-51,81-127 (default camera, scale, projection).
#include "oracle.h"
#include <cmath>
#include <cstring>
#include <vector>

extern "C" void orc_acArray2DToBit(const unsigned char* arr, int w, int h, long long* bit) {
    long long b = 0;
    for (int i = 0; i < h; i++)
        for (int j = w - 1; j >= 0; j--) b = (b << 1) | arr[i * w + j];
    *bit = b;
}

extern "C" void orc_acBitToArray2D(long long bit, unsigned char* arr, int w, int h) {
    for (int i = h - 1; i >= 0; i--)
        for (int j = 0; j < w; j++) {
            arr[i * w + j] = (unsigned char)(bit & 1);
            bit >>= 1;
        }
}

extern "C" void orc_acArray2DRotateub(unsigned char* arr, int w, int h, int rot) {
    std::vector<unsigned char> t(arr, arr + w * h);
    for (int i = 0; i < h; i++)
        for (int j = 0; j < w; j++) {
            if (rot == 1) arr[i * w + j] = t[(h - 1 - j) * w + i];
            else if (rot == 2) arr[i * w + j] = t[(h - 1 - i) * w + (h - 1 - j)];
            else if (rot == 3) arr[i * w + j] = t[j * w + (h - 1 - i)];
        }
}

extern "C" void orc_acBitRotate(long long* bit, int rot, int w, int h) {
    unsigned char arr[64];
    orc_acBitToArray2D(*bit, arr, w, h);
    orc_acArray2DRotateub(arr, w, h, rot);
    orc_acArray2DToBit(arr, w, h, bit);
}

extern "C" void orc_acMatrixTranspose(double* m) {
    double t[16];
    memcpy(t, m, sizeof t);
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) m[j * 4 + i] = t[i * 4 + j];
}

extern "C" void orc_acMatrixToQuaternion(const double* m, double* q) {
    double x, y, z, w, s;
    double t = 1 + m[0] + m[5] + m[10];
    if (t > 0.00000001) {
        s = sqrt(t) * 2;
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
    q[0] = w;
    q[1] = x;
    q[2] = y;
    q[3] = z;
}

extern "C" void orc_acQuaternionToMatrix(const double* q, double* m) {
    double xx = q[1] * q[1], xy = q[1] * q[2], xz = q[1] * q[3], xw = q[1] * q[0];
    double yy = q[2] * q[2], yz = q[2] * q[3], yw = q[2] * q[0];
    double zz = q[3] * q[3], zw = q[3] * q[0];
    m[0] = 1 - 2 * (yy + zz);
    m[1] = 2 * (xy - zw);
    m[2] = 2 * (xz + yw);
    m[4] = 2 * (xy + zw);
    m[5] = 1 - 2 * (xx + zz);
    m[6] = 2 * (yz - xw);
    m[8] = 2 * (xz - yw);
    m[9] = 2 * (yz + xw);
    m[10] = 1 - 2 * (xx + yy);
}

extern "C" double orc_acCalcLength(double x1, double y1, double x2, double y2) {
    double dx = x1 - x2, dy = y1 - y2;
    return sqrt(dx * dx + dy * dy);
}

extern "C" void orc_load_tag(OrcTemplate* tpl, long long bit, int width, int height, double scale) {
    tpl->width = width;
    tpl->height = height;
    tpl->scale = scale;
    for (int i = 0; i < 4; i++) {
        tpl->code[i] = bit;
        orc_acBitRotate(&tpl->code[i], i, width, height);
    }
}

extern "C" void orc_load_template_pixels(OrcTemplate* tpl, const uint8_t* pixels, int w, int h, double scale) {
    // opencvar.cpp:291-301: ROI(1,1,w-2,h-2) -> copy into an 8UC1 image (widthStep = align4) -> >100 -> 1
    // -> vertical flip -> acArray2DToBit with stride = width (quirk D1; padding defined as 0).
    int iw = w - 2, ih = h - 2;
    int ws = (iw + 3) & ~3;
    std::vector<unsigned char> buf((size_t)ws * ih + (size_t)iw * ih, 0);
    for (int y = 0; y < ih; y++)
        for (int x = 0; x < iw; x++) {
            int sy = ih - 1 - y;  // cvFlip mode 0
            buf[(size_t)y * ws + x] = pixels[(size_t)(sy + 1) * w + (x + 1)] > 100 ? 1 : 0;
        }
    long long bit = 0;
    orc_acArray2DToBit(buf.data(), iw, ih, &bit);
    orc_load_tag(tpl, bit, iw, ih, scale);
}

static void camera_projection(OrcCamera* c) {
    double nearp = 0.1, farp = 5000.0;
    double* p = c->glProjection;
    memset(p, 0, sizeof(double) * 16);
    p[0] = 2. * c->cameraMatrix[0] / c->width;
    p[5] = 2. * c->cameraMatrix[4] / c->height;
    p[2] = 2. * (c->cameraMatrix[2] / c->width) - 1.;
    p[6] = 2. * (c->cameraMatrix[5] / c->height) - 1.;
    p[10] = -(farp + nearp) / (farp - nearp);
    p[11] = -2. * farp * nearp / (farp - nearp);
    p[14] = -1;
    orc_acMatrixTranspose(p);
}

extern "C" void orc_camera_default(OrcCamera* c) {
    c->width = 640;
    c->height = 480;
    double K[9] = {500, 0, 320, 0, 500, 240, 0, 0, 1};
    memcpy(c->cameraMatrix, K, sizeof K);
    memset(c->distCoeffs, 0, sizeof c->distCoeffs);
    camera_projection(c);
}

extern "C" void orc_camera_scale(OrcCamera* c, int width, int height) {
    double ru = (double)width / c->width, rv = (double)height / c->height;
    c->cameraMatrix[0] *= ru;
    c->cameraMatrix[4] *= rv;
    c->cameraMatrix[2] *= ru;
    c->cameraMatrix[5] *= rv;
    c->width = width;
    c->height = height;
    camera_projection(c);
}
