// binarise.hip -- fused grey + pyramid denoise + adaptive threshold + neighbour-mask kernel (gfx950).
//
// One LDS-tiled pass replaces the image stages of cvarFindSquares
// (/root/reference/src/opencvar.cpp:156-184): cvCloneImage, cvPyrDown(5x5), cvPyrUp, cvCvtColor(BGR2GRAY),
// cvAdaptiveThreshold(GAUSSIAN_C, 7x7, delta 8) -- and, in frame mode, the BGR2GRAY of
// cvarArMultRegistration (opencvar.cpp:624-627).  It also does the part of cvFindContours that is local
// (opencvar.cpp:183-184): zeroing the 1-px frame, and spotting where a border can begin.
//
// Per output tile of 64x32 pixels a workgroup stages, all in LDS:
//   G  80x48 u8   grey source incl. the 8-px halo (BORDER_REFLECT_101 applied at load)
//   Ph 48x38 u16  horizontal [1 4 6 4 1] at even columns      P 22x38 u8  pyrDown result ((v+128)>>8)
//   U  40x72 u8   pyrUp result ((v+32)>>6), replicate-clamped to the image for the Gaussian
//   H  40x66 u16  horizontal [8 28 56 72 56 28 8]              B 34x66 u8  threshold (src - mean > -8), frame zeroed
// and writes the grey plane (frame mode), the 8-bit non-zero-neighbour mask per pixel, and the list of
// plausible border starts.  HBM traffic per pixel: 3 B read + 1 B grey + 1 B mask (frame mode).
// Integer arithmetic throughout, so the result is bit-identical to the sequential definition.
#include "kernels.h"

namespace ocvar {

constexpr int TW = TILE_W, TH = TILE_H, NT = 256;
constexpr int GW = TW + 16, GH = TH + 16;
constexpr int PW = TW / 2 + 6, PH = TH / 2 + 6;
constexpr int UW = TW + 8, UH = TH + 8;
constexpr int BW = TW + 2, BH = TH + 2;

struct TileLds {
    uint8_t G[GH][GW];
    uint16_t Ph[GH][PW + 2];
    uint8_t P[PH][PW + 2];
    uint8_t U[UH][UW];
    uint16_t Hh[UH][BW + 2];
    uint8_t B[BH][BW + 2];
};

__device__ __forceinline__ int reflect101(int p, int len) {
    if (len == 1) return 0;
    while ((unsigned)p >= (unsigned)len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}

__device__ __forceinline__ int grey_of(const uint8_t* p) {
    return (p[0] * 1868 + p[1] * 9617 + p[2] * 4899 + 8192) >> 14;
}

// index of the pyramid sample that stands in for virtual index v (pyrUp borders: -1 -> 1, n -> n-1)
__device__ __forceinline__ int pyr_index(int v, int n) { return v < 0 ? (n > 1 ? 1 : 0) : (v >= n ? n - 1 : v); }

template <bool BGR>
__device__ void binarise_tile(TileLds& L, const uint8_t* src, long long src_stride, int roi_index, int img_w, int img_h,
                              int sw, int sh, int X0, int Y0, uint8_t* gray_out, long long gray_stride, uint8_t* bgr_out,
                              uint8_t* nbr, StartCand* cands, int* n_cands, int cap_cands, int* err) {
    const int tid = threadIdx.x;
    const int pw = sw >> 1, ph = sh >> 1;
    const int px0 = X0 / 2 - 3, py0 = Y0 / 2 - 3;

    // 1. grey tile with halo
    for (int i = tid; i < GW * GH; i += NT) {
        const int gx = i % GW, gy = i / GW;
        const int x = reflect101(X0 - 8 + gx, sw), y = reflect101(Y0 - 8 + gy, sh);
        const uint8_t* p = src + (long long)y * src_stride + (BGR ? 3 * x : x);
        L.G[gy][gx] = BGR ? (uint8_t)grey_of(p) : *p;
    }
    __syncthreads();

    if (BGR) {  // grey plane (and the reference's in-place grey of the caller's frame)
        for (int i = tid; i < TW * TH; i += NT) {
            const int tx = i % TW, ty = i / TW;
            const int x = X0 + tx, y = Y0 + ty;
            if (x < sw && y < sh) {
                const uint8_t g = L.G[ty + 8][tx + 8];
                gray_out[(long long)y * gray_stride + x] = g;
                if (bgr_out) {
                    uint8_t* q = bgr_out + (long long)y * src_stride + 3 * x;
                    q[0] = q[1] = q[2] = g;
                }
            }
        }
        // odd width / height: the last column / row lies outside the even working size but still is greyed
        if (img_w > sw && X0 + TW >= sw) {
            for (int ty = tid; ty < TH + 1; ty += NT) {
                const int y = Y0 + ty;
                const bool last_row = (ty == TH);
                if ((!last_row && y < sh) || (last_row && img_h > sh && Y0 + TH >= sh)) {
                    const int yy = last_row ? sh : y;
                    const uint8_t g = (uint8_t)grey_of(src + (long long)yy * src_stride + 3 * sw);
                    gray_out[(long long)yy * gray_stride + sw] = g;
                    if (bgr_out) {
                        uint8_t* q = bgr_out + (long long)yy * src_stride + 3 * sw;
                        q[0] = q[1] = q[2] = g;
                    }
                }
            }
        }
        if (img_h > sh && Y0 + TH >= sh) {
            for (int tx = tid; tx < TW; tx += NT) {
                const int x = X0 + tx;
                if (x < sw) {
                    const uint8_t g = (uint8_t)grey_of(src + (long long)sh * src_stride + 3 * x);
                    gray_out[(long long)sh * gray_stride + x] = g;
                    if (bgr_out) {
                        uint8_t* q = bgr_out + (long long)sh * src_stride + 3 * x;
                        q[0] = q[1] = q[2] = g;
                    }
                }
            }
        }
    }

    // 2a. pyrDown, horizontal
    for (int i = tid; i < GH * PW; i += NT) {
        const int tx = i % PW, gy = i / PW;
        int c0 = 2 * (pyr_index(px0 + tx, pw) - px0);
        c0 = c0 < 0 ? 0 : (c0 > GW - 5 ? GW - 5 : c0);  // no-op for any image of 2 or more columns
        const uint8_t* g = &L.G[gy][c0];
        L.Ph[gy][tx] = (uint16_t)(g[0] + 4 * g[1] + 6 * g[2] + 4 * g[3] + g[4]);
    }
    __syncthreads();
    // 2b. pyrDown, vertical
    for (int i = tid; i < PH * PW; i += NT) {
        const int tx = i % PW, ty = i / PW;
        int r0 = 2 * (pyr_index(py0 + ty, ph) - py0);
        r0 = r0 < 0 ? 0 : (r0 > GH - 5 ? GH - 5 : r0);
        const int v = L.Ph[r0][tx] + 4 * L.Ph[r0 + 1][tx] + 6 * L.Ph[r0 + 2][tx] + 4 * L.Ph[r0 + 3][tx] + L.Ph[r0 + 4][tx];
        L.P[ty][tx] = (uint8_t)((v + 128) >> 8);
    }
    __syncthreads();
    // 3. pyrUp at replicate-clamped coordinates
    for (int i = tid; i < UH * UW; i += NT) {
        const int ux = i % UW, uy = i / UW;
        int ex = X0 - 4 + ux, ey = Y0 - 4 + uy;
        ex = ex < 0 ? 0 : (ex > sw - 1 ? sw - 1 : ex);
        ey = ey < 0 ? 0 : (ey > sh - 1 ? sh - 1 : ey);
        int kx = (ex >> 1) - px0, ky = (ey >> 1) - py0;
        kx = kx < 1 ? 1 : (kx > PW - 2 ? PW - 2 : kx);  // no-ops, see the halo arithmetic in DESIGN.md
        ky = ky < 1 ? 1 : (ky > PH - 2 ? PH - 2 : ky);
        int r0, r1, r2 = 0;  // horizontally up-sampled values of rows ky-1, ky, ky+1 (as needed)
        if (ex & 1) {
            if (ey & 1) {
                r0 = 4 * (L.P[ky][kx] + L.P[ky][kx + 1]);
                r1 = 4 * (L.P[ky + 1][kx] + L.P[ky + 1][kx + 1]);
                L.U[uy][ux] = (uint8_t)((4 * (r0 + r1) + 32) >> 6);
            } else {
                r0 = 4 * (L.P[ky - 1][kx] + L.P[ky - 1][kx + 1]);
                r1 = 4 * (L.P[ky][kx] + L.P[ky][kx + 1]);
                r2 = 4 * (L.P[ky + 1][kx] + L.P[ky + 1][kx + 1]);
                L.U[uy][ux] = (uint8_t)((r0 + 6 * r1 + r2 + 32) >> 6);
            }
        } else {
            if (ey & 1) {
                r0 = L.P[ky][kx - 1] + 6 * L.P[ky][kx] + L.P[ky][kx + 1];
                r1 = L.P[ky + 1][kx - 1] + 6 * L.P[ky + 1][kx] + L.P[ky + 1][kx + 1];
                L.U[uy][ux] = (uint8_t)((4 * (r0 + r1) + 32) >> 6);
            } else {
                r0 = L.P[ky - 1][kx - 1] + 6 * L.P[ky - 1][kx] + L.P[ky - 1][kx + 1];
                r1 = L.P[ky][kx - 1] + 6 * L.P[ky][kx] + L.P[ky][kx + 1];
                r2 = L.P[ky + 1][kx - 1] + 6 * L.P[ky + 1][kx] + L.P[ky + 1][kx + 1];
                L.U[uy][ux] = (uint8_t)((r0 + 6 * r1 + r2 + 32) >> 6);
            }
        }
    }
    __syncthreads();
    // 4. Gaussian, horizontal
    for (int i = tid; i < UH * BW; i += NT) {
        const int bx = i % BW, uy = i / BW;
        const uint8_t* u = &L.U[uy][bx];
        L.Hh[uy][bx] = (uint16_t)(8 * (u[0] + u[6]) + 28 * (u[1] + u[5]) + 56 * (u[2] + u[4]) + 72 * u[3]);
    }
    __syncthreads();
    // 5. Gaussian, vertical + threshold + zero frame
    for (int i = tid; i < BH * BW; i += NT) {
        const int bx = i % BW, by = i / BW;
        const int x = X0 - 1 + bx, y = Y0 - 1 + by;
        int b = 0;
        if (x >= 1 && x <= sw - 2 && y >= 1 && y <= sh - 2) {
            const int acc = 8 * (L.Hh[by][bx] + L.Hh[by + 6][bx]) + 28 * (L.Hh[by + 1][bx] + L.Hh[by + 5][bx]) +
                            56 * (L.Hh[by + 2][bx] + L.Hh[by + 4][bx]) + 72 * L.Hh[by + 3][bx];
            const int mean = (acc + 32768) >> 16;
            b = ((int)L.U[by + 3][bx + 3] - mean > -8) ? 1 : 0;
        }
        L.B[by][bx] = (uint8_t)b;
    }
    __syncthreads();
    // 6. neighbour masks + plausible border starts
    const int lane = tid & 63;
    for (int i = tid; i < TW * TH; i += NT) {
        const int tx = i % TW, ty = i / TW;
        const int x = X0 + tx, y = Y0 + ty;
        const int bx = tx + 1, by = ty + 1;
        int type = -1;
        if (x < sw && y < sh) {
            const int c = L.B[by][bx];
            const int e = L.B[by][bx + 1], ne = L.B[by - 1][bx + 1], n = L.B[by - 1][bx], nw = L.B[by - 1][bx - 1];
            const int w = L.B[by][bx - 1], swp = L.B[by + 1][bx - 1], s = L.B[by + 1][bx], se = L.B[by + 1][bx + 1];
            nbr[(long long)y * sw + x] = (uint8_t)(e | (ne << 1) | (n << 2) | (nw << 3) | (w << 4) | (swp << 5) | (s << 6) | (se << 7));
            if (c && !(w | nw | n | ne)) type = 0;        // can be the raster-first pixel of a component
            else if (!c && w && n) type = 1;              // can be the raster-first pixel of a hole
        }
        const unsigned long long mask = __ballot(type >= 0);
        if (mask) {
            int base = 0;
            const int leader = __ffsll((long long)mask) - 1;
            if (lane == leader) base = atomicAdd(n_cands, __popcll(mask));
            base = __shfl(base, leader);
            if (type >= 0) {
                const int slot = base + __popcll(mask & ((1ull << lane) - 1ull));
                if (slot < cap_cands) {
                    StartCand sc;
                    sc.roi = roi_index;
                    sc.pos = y * sw + x;
                    sc.is_hole = type;
                    cands[slot] = sc;
                } else {
                    atomicOr(err, ERR_CAND_OVERFLOW);
                }
            }
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(NT) void binarise_frames_kernel(Workspace ws, const uint8_t* bgr, int row_stride,
                                                             size_t frame_stride, int grey_in_place) {
    __shared__ TileLds L;
    const int tiles_x = (ws.sw + TW - 1) / TW;
    const int f = blockIdx.y;
    const int X0 = (blockIdx.x % tiles_x) * TW, Y0 = (blockIdx.x / tiles_x) * TH;
    const uint8_t* src = bgr + (size_t)f * frame_stride;
    binarise_tile<true>(L, src, row_stride, f, ws.W, ws.H, ws.sw, ws.sh, X0, Y0, ws.gray + (size_t)f * ws.W * ws.H, ws.W,
                        nullptr, ws.nbr_frame + (size_t)f * ws.sw * ws.sh,
                        ws.cands_frame, ws.counters + CNT_FRAME_CANDS, ws.cap_frame_cands, ws.counters + CNT_ERR);
}

// The reference greys the caller's frame in place (opencvar.cpp:624-627).  Done as its own pass over the
// grey plane so that no tile ever reads a half-written BGR pixel of a neighbouring tile's halo.
__global__ __launch_bounds__(NT) void grey_writeback_kernel(Workspace ws, uint8_t* bgr, int row_stride, size_t frame_stride) {
    const int f = blockIdx.y;
    const uint8_t* g = ws.gray + (size_t)f * ws.W * ws.H;
    uint8_t* dst = bgr + (size_t)f * frame_stride;
    const long long n = (long long)ws.W * ws.H;
    for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < n; i += (long long)gridDim.x * NT) {
        const int x = (int)(i % ws.W), y = (int)(i / ws.W);
        const uint8_t v = g[i];
        uint8_t* q = dst + (long long)y * row_stride + 3 * x;
        q[0] = q[1] = q[2] = v;
    }
}

__global__ __launch_bounds__(NT) void binarise_crops_kernel(Workspace ws) {
    __shared__ TileLds L;
    int n_tiles = ws.counters[CNT_CROP_TILES];
    if (n_tiles > ws.cap_crop_tiles) n_tiles = ws.cap_crop_tiles;
    for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const TileDesc td = ws.tiles_crop[t];
        const Roi r = ws.rois_crop[td.roi];
        const uint8_t* src = ws.gray + (size_t)r.frame * ws.W * ws.H + (size_t)r.y0 * ws.W + r.x0;
        binarise_tile<false>(L, src, ws.W, td.roi, r.w, r.h, r.sw, r.sh, td.x0, td.y0, nullptr, 0, nullptr,
                             ws.nbr_crop + r.nbr_off, ws.cands_crop, ws.counters + CNT_CROP_CANDS, ws.cap_crop_cands,
                             ws.counters + CNT_ERR);
    }
}

void launch_binarise_frames(const Workspace& ws, const uint8_t* d_bgr, int row_stride, size_t frame_stride, int grey_in_place,
                            hipStream_t stream) {
    const int tiles = ((ws.sw + TW - 1) / TW) * ((ws.sh + TH - 1) / TH);
    if (tiles <= 0 || ws.n_frames <= 0) return;
    hipLaunchKernelGGL(binarise_frames_kernel, dim3(tiles, ws.n_frames), dim3(NT), 0, stream, ws, d_bgr, row_stride, frame_stride,
                       grey_in_place);
    if (grey_in_place)
        hipLaunchKernelGGL(grey_writeback_kernel, dim3(1024, ws.n_frames), dim3(NT), 0, stream, ws, const_cast<uint8_t*>(d_bgr),
                           row_stride, frame_stride);
}

void launch_binarise_crops(const Workspace& ws, hipStream_t stream) {
    hipLaunchKernelGGL(binarise_crops_kernel, dim3(2048), dim3(NT), 0, stream, ws);
}

}  // namespace ocvar
