// orc_imgproc.cpp -- oracle: image stages of cvarFindSquares (TEST INFRASTRUCTURE, see oracle.h).
//
// Restates, for one channel, the OpenCV 2.4.x arithmetic behind these reference call sites:
//   opencvar.cpp:625      cvCvtColor BGR2GRAY                       
//   opencvar.cpp:175      cvPyrDown(timg, pyr, CV_GAUSSIAN_5x5)     
//   opencvar.cpp:176      cvPyrUp(pyr, timg)                        
//   opencvar.cpp:180      cvCvtColor(timg, tgray) -- identity on equal channels
//   opencvar.cpp:181-182  cvAdaptiveThreshold(GAUSSIAN_C, BINARY, 7, 8)
// Parity unpinned at the OpenCV boundary (no OpenCV here, no reference tests).
#include "oracle.h"
#include <vector>
#include <cstdlib>

static inline int reflect101(int p, int len) {
    if (len == 1) return 0;
    while ((unsigned)p >= (unsigned)len) {
        if (p < 0) p = -p;
        else p = 2 * len - 2 - p;
    }
    return p;
}

extern "C" void orc_bgr2gray(const uint8_t* bgr, int w, int h, int stride, uint8_t* gray, int gstride) {
    for (int y = 0; y < h; y++) {
        const uint8_t* s = bgr + (size_t)y * stride;
        uint8_t* d = gray + (size_t)y * gstride;
        for (int x = 0; x < w; x++)
            d[x] = (uint8_t)((s[3 * x] * 1868 + s[3 * x + 1] * 9617 + s[3 * x + 2] * 4899 + 8192) >> 14);
    }
}

// cvPyrDown on the even-sized ROI (sw x sh) -> (sw/2 x sh/2).
static void pyr_down(const uint8_t* src, int sw, int sh, int stride, std::vector<uint8_t>& dst) {
    int dw = sw / 2, dh = sh / 2;
    dst.assign((size_t)dw * dh, 0);
    std::vector<int> rows((size_t)sh * dw);
    for (int y = 0; y < sh; y++) {
        const uint8_t* s = src + (size_t)y * stride;
        for (int x = 0; x < dw; x++) {
            int xm2 = reflect101(2 * x - 2, sw), xm1 = reflect101(2 * x - 1, sw);
            int xp1 = reflect101(2 * x + 1, sw), xp2 = reflect101(2 * x + 2, sw);
            rows[(size_t)y * dw + x] = s[2 * x] * 6 + (s[xm1] + s[xp1]) * 4 + s[xm2] + s[xp2];
        }
    }
    for (int y = 0; y < dh; y++) {
        const int* r0 = &rows[(size_t)reflect101(2 * y - 2, sh) * dw];
        const int* r1 = &rows[(size_t)reflect101(2 * y - 1, sh) * dw];
        const int* r2 = &rows[(size_t)(2 * y) * dw];
        const int* r3 = &rows[(size_t)reflect101(2 * y + 1, sh) * dw];
        const int* r4 = &rows[(size_t)reflect101(2 * y + 2, sh) * dw];
        for (int x = 0; x < dw; x++)
            dst[(size_t)y * dw + x] = (uint8_t)((r2[x] * 6 + (r1[x] + r3[x]) * 4 + r0[x] + r4[x] + 128) >> 8);
    }
}

// cvPyrUp (sw x sh) -> (2sw x 2sh).
static void pyr_up(const std::vector<uint8_t>& src, int sw, int sh, std::vector<uint8_t>& dst) {
    int dw = 2 * sw, dh = 2 * sh;
    dst.assign((size_t)dw * dh, 0);
    std::vector<int> hr((size_t)sh * dw);
    for (int y = 0; y < sh; y++) {
        const uint8_t* s = &src[(size_t)y * sw];
        int* r = &hr[(size_t)y * dw];
        if (sw == 1) {
            r[0] = r[1] = s[0] * 8;
            continue;
        }
        r[0] = s[0] * 6 + s[1] * 2;
        r[1] = (s[0] + s[1]) * 4;
        for (int x = 1; x < sw - 1; x++) {
            r[2 * x] = s[x - 1] + s[x] * 6 + s[x + 1];
            r[2 * x + 1] = (s[x] + s[x + 1]) * 4;
        }
        r[2 * sw - 2] = s[sw - 2] + s[sw - 1] * 7;
        r[2 * sw - 1] = s[sw - 1] * 8;
    }
    for (int y = 0; y < sh; y++) {
        int ym = (y == 0) ? reflect101(-2, dh) / 2 : y - 1;
        int yp = (y == sh - 1) ? sh - 1 : y + 1;
        const int* r0 = &hr[(size_t)ym * dw];
        const int* r1 = &hr[(size_t)y * dw];
        const int* r2 = &hr[(size_t)yp * dw];
        uint8_t* d0 = &dst[(size_t)(2 * y) * dw];
        uint8_t* d1 = &dst[(size_t)(2 * y + 1) * dw];
        for (int x = 0; x < dw; x++) {
            d1[x] = (uint8_t)(((r1[x] + r2[x]) * 4 + 32) >> 6);
            d0[x] = (uint8_t)((r0[x] + r1[x] * 6 + r2[x] + 32) >> 6);
        }
    }
}

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

extern "C" void orc_binarise(const uint8_t* gray, int w, int h, int stride, uint8_t* bin, uint8_t* up_out) {
    int sw = w & -2, sh = h & -2;  // opencvar.cpp:158
    if (sw < 2 || sh < 2) return;
    std::vector<uint8_t> pyr, up;
    pyr_down(gray, sw, sh, stride, pyr);
    pyr_up(pyr, sw / 2, sh / 2, up);
    if (up_out)
        for (size_t i = 0; i < up.size(); i++) up_out[i] = up[i];
    // 7x7 Gaussian, 8-bit fixed point kernel {8,28,56,72,56,28,8}, BORDER_REPLICATE.
    static const int k[7] = {8, 28, 56, 72, 56, 28, 8};
    std::vector<int> hs((size_t)sw * sh);
    for (int y = 0; y < sh; y++)
        for (int x = 0; x < sw; x++) {
            int acc = 0;
            for (int t = -3; t <= 3; t++) acc += k[t + 3] * up[(size_t)y * sw + clampi(x + t, 0, sw - 1)];
            hs[(size_t)y * sw + x] = acc;
        }
    for (int y = 0; y < sh; y++)
        for (int x = 0; x < sw; x++) {
            int acc = 0;
            for (int t = -3; t <= 3; t++) acc += k[t + 3] * hs[(size_t)clampi(y + t, 0, sh - 1) * sw + x];
            int mean = (acc + 32768) >> 16;
            bin[(size_t)y * sw + x] = ((int)up[(size_t)y * sw + x] - mean > -8) ? 255 : 0;
        }
}
