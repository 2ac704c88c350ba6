// follow.hip -- border following + polygon approximation + quad filter, and the per-frame ordering that turns
// the surviving quads into the reference's square sequence and into crop work for the second pass (gfx950).
//
// follow_kernel replaces the contour half of cvarFindSquares (/root/reference/src/opencvar.cpp:183-214) for
// every ROI of a pass at once: one lane per plausible border start (trace_core.h), waves pull 64 starts at a
// time from a ticket counter so that a long border only holds up its own wave.
// order_and_crops_kernel replaces cvarGetAllSquares (564-590), the tracking loop (635-668) and the crop
// rectangle set-up of the candidate loop (676-693).
#include "kernels.h"
#include "trace_core.h"

namespace ocvar {

template <bool CROP>
__device__ __forceinline__ void follow_one(const Workspace& ws, const StartCand c) {
    int sw, sh, img_w, img_h;   // sw: row stride of the neighbour plane (positions are y*stride + x)
    const uint8_t* nbr;
    if (CROP) {
        const Roi r = ws.rois_crop[c.roi];
        sw = r.ns; sh = r.sh; img_w = r.w; img_h = r.h;
        nbr = ws.nbr_crop + r.nbr_off;
    } else {
        sw = ws.ns; sh = ws.sh; img_w = ws.W; img_h = ws.H;
        nbr = ws.nbr_frame + (size_t)c.roi * ws.ns * ws.sh;
    }
    const int plane = sw * sh;
    if (c.pos <= 0 || c.pos >= plane) return;
    if (!c.is_hole && earlier_start_behind(nbr, sw, plane, c.pos, 0, BACK_STEPS)) return;
    const int max_steps = 4 * plane + 16;
    const TraceStats st = trace_border<false>(nbr, sw, plane, c.pos, c.is_hole, nullptr, 0, max_steps);
    if (st.status == TRACE_OVERRUN) {
        atomicOr(ws.counters + CNT_ERR, ERR_TRACE_OVERRUN);
        return;
    }
    if (!worth_approximating(st)) return;
    const int need = 2 * st.npts + 2 * (st.npts + 2);
    const long long off = atomicAdd(reinterpret_cast<unsigned long long*>(ws.counters + CNT_POOL_INTS), (unsigned long long)need);
    if (off + need > ws.cap_pool_ints) {
        atomicOr(ws.counters + CNT_ERR, ERR_POOL_OVERFLOW);
        return;
    }
    int* pts = ws.pool + off;
    DpSlice* stack = reinterpret_cast<DpSlice*>(pts + 2 * st.npts);
    trace_border<true>(nbr, sw, plane, c.pos, c.is_hole, pts, st.npts, max_steps);
    int dst[2 * (DP_MAX_OUT + 1)];
    const int m = approx_poly_dp(pts, st.npts, st.perimeter * 0.02, dst, stack);
    if (m != 4 || !quad_filter(dst, img_w, img_h)) return;
    QuadRec q;
    q.roi = c.roi;
    q.start = c.pos;
    for (int k = 0; k < 8; k++) q.pt[k] = dst[k];
    if (CROP) {
        const int slot = atomicAdd(ws.counters + CNT_CROP_QUADS, 1);
        if (slot >= ws.cap_crop_quads) {
            atomicOr(ws.counters + CNT_ERR, ERR_QUAD_OVERFLOW);
            return;
        }
        ws.quads_crop[slot] = q;
        // cvarGetSquare keeps the LAST quad of the sequence = the earliest discovered one (opencvar.cpp:401-430)
        atomicMin(ws.best_crop + c.roi, ((unsigned long long)(unsigned)c.pos << 32) | (unsigned)slot);
    } else {
        const int slot = atomicAdd(ws.n_quads_frame + c.roi, 1);
        if (slot >= MAXQ) {
            atomicOr(ws.counters + CNT_ERR, ERR_QUAD_OVERFLOW);
            return;
        }
        ws.quads_frame[(size_t)c.roi * MAXQ + slot] = q;
    }
}

template <bool CROP>
__global__ __launch_bounds__(256) void follow_kernel(Workspace ws) {
    const StartCand* cands = CROP ? ws.cands_crop : ws.cands_frame;
    int n = ws.counters[CROP ? CNT_CROP_CANDS : CNT_FRAME_CANDS];
    const int cap = CROP ? ws.cap_crop_cands : ws.cap_frame_cands;
    if (n > cap) n = cap;
    int* ticket = ws.counters + (CROP ? CNT_TICKET_C : CNT_TICKET_F);
    const int lane = threadIdx.x & 63;
    for (;;) {
        int base = 0;
        if (lane == 0) base = atomicAdd(ticket, 64);
        base = __shfl(base, 0);
        if (base >= n) break;
        const int idx = base + lane;
        if (idx < n) follow_one<CROP>(ws, cands[idx]);
    }
}

__global__ __launch_bounds__(256) void order_and_crops_kernel(Workspace ws) {
    __shared__ int s_start[MAXQ];
    __shared__ float s_sq[MAXQ][8];
    __shared__ int s_n;
    const int f = blockIdx.x;
    const int tid = threadIdx.x;
    int n = ws.n_quads_frame[f];
    if (n > MAXQ) n = MAXQ;
    const QuadRec* q = ws.quads_frame + (size_t)f * MAXQ;
    for (int i = tid; i < n; i += blockDim.x) s_start[i] = q[i].start;
    __syncthreads();
    // sequence order of cvarFindSquares: contours come out last-discovered first (opencvar.cpp:187-214)
    for (int i = tid; i < n; i += blockDim.x) {
        const int mine = s_start[i];
        int rank = 0;
        for (int u = 0; u < n; u++) rank += s_start[u] > mine;
        for (int k = 0; k < 8; k++) s_sq[rank][k] = (float)q[i].pt[k];
    }
    __syncthreads();
    if (tid == 0) {
        int nr = 0, nn = n;
        const int np = ws.n_prev[f];
        if (np > 0)
            nn = track_markers(ws.prev + (size_t)f * MAXM, np < MAXM ? np : MAXM, &s_sq[0][0], n, ws.reserve + (size_t)f * MAXM,
                               MAXM, &nr);
        ws.n_reserve[f] = nr;
        ws.n_squares[f] = nn;
        s_n = nn;
    }
    __syncthreads();
    n = s_n;
    for (int i = tid; i < n; i += blockDim.x) {
        float* out = ws.squares + ((size_t)f * MAXQ + i) * 8;
        int quad[8];
        for (int k = 0; k < 8; k++) {
            out[k] = s_sq[i][k];
            quad[k] = (int)s_sq[i][k];
        }
        int x0, y0, cw, ch, roi_index = -1;
        crop_rect(quad, ws.W, ws.H, &x0, &y0, &cw, &ch);
        const int sw = cw & ~1, sh = ch & ~1, ns = (sw + 3) & ~3;
        if (sw >= 2 && sh >= 2) {
            const int r = atomicAdd(ws.counters + CNT_CROP_ROIS, 1);
            const long long plane = (long long)ns * sh;
            const long long off = (long long)atomicAdd(ws.crop_pixels, (unsigned long long)plane);
            const int ntx = (sw + MARCH_STRIP - 1) / MARCH_STRIP, nty = (sh + MARCH_CROP_ROWS - 1) / MARCH_CROP_ROWS;
            if (r >= ws.cap_crop_rois || off + plane > ws.cap_crop_pixels) {
                atomicOr(ws.counters + CNT_ERR, ERR_CROP_OVERFLOW);
            } else {
                const int tbase = atomicAdd(ws.counters + CNT_CROP_TILES, ntx * nty);
                if (tbase + ntx * nty > ws.cap_crop_tiles) {
                    atomicOr(ws.counters + CNT_ERR, ERR_TILE_OVERFLOW);
                } else {
                    Roi roi;
                    roi.frame = f; roi.x0 = x0; roi.y0 = y0; roi.w = cw; roi.h = ch; roi.sw = sw; roi.sh = sh; roi.ns = ns;
                    roi.owner = i; roi.nbr_off = off;
                    ws.rois_crop[r] = roi;
                    ws.best_crop[r] = ~0ull;
                    for (int t = 0; t < ntx * nty; t++) {
                        TileDesc td;
                        td.roi = r; td.x0 = t % ntx; td.y0 = (t / ntx) * MARCH_CROP_ROWS;
                        ws.tiles_crop[tbase + t] = td;
                    }
                    roi_index = r;
                }
            }
        }
        ws.crop_of[(size_t)f * MAXQ + i] = roi_index;
    }
}

void launch_follow_frames(const Workspace& ws, hipStream_t stream) {
    hipLaunchKernelGGL(follow_kernel<false>, dim3(1024), dim3(256), 0, stream, ws);
}
void launch_follow_crops(const Workspace& ws, hipStream_t stream) {
    hipLaunchKernelGGL(follow_kernel<true>, dim3(1024), dim3(256), 0, stream, ws);
}
void launch_order_and_crops(const Workspace& ws, hipStream_t stream) {
    if (ws.n_frames > 0) hipLaunchKernelGGL(order_and_crops_kernel, dim3(ws.n_frames), dim3(256), 0, stream, ws);
}

}  // namespace ocvar
