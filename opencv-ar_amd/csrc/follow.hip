// follow.hip -- border following + polygon approximation + quad filter, and the per-frame ordering that turns
// the surviving quads into the reference's square sequence and into crop work for the second pass (gfx950).
//
// follow_kernel (tier 1, a router), follow_mid_kernel (tier 2) and follow_long_kernel (tier 3) replace the contour half
// of cvarFindSquares (/root/reference/src/opencvar.cpp:183-214) for every ROI of a pass at once: every plausible border
// start (trace_core.h) is a work item pulled from a ticket counter; tier 1 decides which starts are worth following,
// tier 2 follows 64 borders per wave (one lane each), tier 3 one image-sized border per wave.
// order_and_crops_kernel replaces cvarGetAllSquares (564-590), the tracking loop (635-668) and the crop
// rectangle set-up of the candidate loop (676-693).
#include "kernels.h"
#include "trace_core.h"

namespace ocvar {

// Wave-uniform values.  The wave-cooperative code below (tier-2 finishing, tier 3) keeps all 64 lanes on the same
// border, so its control flow is uniform by construction -- but the compiler only knows that for values it can prove
// uniform.  For a value that lives in a VGPR it builds EXEC-masked "divergent" control flow instead, and it may then
// place a cross-lane operation (ds_bpermute, DPP) where some lanes are masked off; a loop with `continue` around such
// code hung on hardware.  Passing every control value through v_readfirstlane makes the branches scalar (and moves
// the walker's state to SGPRs / the scalar ALU).
//
// One more trap of the same family: `if (lane == 0) ...` as the last statement of a loop body followed by
// `if (lane == 0) ticket = atomicAdd(...)` as the first statement of the next iteration.  The optimiser threads the two
// identical predicates through the back-edge and rotates the loop so that lanes 1..63 iterate WITHOUT lane 0 (their
// ticket is the phi's initial 0, v_readfirstlane then reads lane 1): those lanes re-run task 0 forever.  The ticket
// fetch therefore tests an opaque copy of the lane id (ticket_lane()) that the compiler cannot relate to `lane`.
__device__ __forceinline__ int ticket_lane() {
    int l = (int)(threadIdx.x & 63);
    asm volatile("" : "+v"(l));
    return l;
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ unsigned uni(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ unsigned long long uni(unsigned long long v) {
    return ((unsigned long long)uni((unsigned)(v >> 32)) << 32) | uni((unsigned)v);
}
__device__ __forceinline__ double uni(double v) { return __longlong_as_double((long long)uni((unsigned long long)__double_as_longlong(v))); }

// Profiling build (-DOCVAR_PROF, `make prof`): every tier-2 wave accumulates cycle counts (s_memtime) and event counts and
// adds them once, at its end, to the 64-bit slots at counters + CNT_PROF (read back through ocvar_hip_counters, values
// 10..41; tools/prof_tier2.py).  Compiled out of the product build.
#ifdef OCVAR_PROF
#define PROF_NOW() __builtin_readcyclecounter()
#define PROF_DECL() unsigned long long prof_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define PROF_ADD(slot, v) do { prof_acc[slot] += (unsigned long long)(v); } while (0)
#define PROF_FLUSH(base) do { if ((threadIdx.x & 63) == 0) for (int i_ = 0; i_ < 16; i_++) if (i_ != 7) atomicAdd(reinterpret_cast<unsigned long long*>(ws.counters + CNT_PROF) + (base) + i_, prof_acc[i_]); } while (0)
#else
#define PROF_NOW() 0ull
#define PROF_DECL() do { } while (0)
#define PROF_ADD(slot, v) do { (void)(v); } while (0)
#define PROF_FLUSH(base) do { } while (0)
#endif

// Waves per workgroup of the follower kernels.  Their waves are independent (no barrier after the table set-up), so a workgroup
// is ONE wave: with several contexts in flight the GPU is full of another context's binarise workgroups, whose retiring waves
// free 80 registers on one SIMD at a time -- a 4-wave workgroup must find room on all four SIMDs of a CU at once (and four times
// the LDS) and waits for it (round 2: tier 3 took 4 ms in-region for 0.3 ms of work), a 1-wave workgroup takes the first gap.
constexpr int FW = 1;
constexpr int FOLLOW_THREADS = 64 * FW;

// geometry of the ROI a start belongs to
struct PlaneRef {
    const uint8_t* nbr;
    int ns, sh, img_w, img_h, plane;
};

template <bool CROP>
__device__ __forceinline__ PlaneRef plane_of(const Workspace& ws, int roi) {
    PlaneRef p;
    if (CROP) {
        const Roi r = ws.rois_crop[roi];
        p.ns = r.ns; p.sh = r.sh; p.img_w = r.w; p.img_h = r.h;
        p.nbr = ws.nbr_crop + r.nbr_off;
    } else {
        p.ns = ws.ns; p.sh = ws.sh; p.img_w = ws.W; p.img_h = ws.H;
        p.nbr = ws.nbr_frame + (size_t)roi * nbr_plane_bytes(ws.ns, ws.sh);
    }
    p.plane = p.ns * p.sh;
    return p;
}

// Douglas-Peucker + quad filter on stored points, then publication of the quad (one lane).
template <bool CROP>
__device__ __forceinline__ void emit_quad(const Workspace& ws, const StartCand c, const int* dst);

template <bool CROP>
__device__ __forceinline__ void approximate_and_emit(const Workspace& ws, const StartCand c, const PlaneRef& pl, const int* pts,
                                                     int npts, double perimeter, DpSlice* stack, bool emit = true) {
    int dst[2 * (DP_MAX_OUT + 1)];
    const int m = approx_poly_dp(pts, npts, perimeter * 0.02, dst, stack);
    if (m != 4 || !quad_filter(dst, pl.img_w, pl.img_h)) return;
    if (emit) emit_quad<CROP>(ws, c, dst);
}

template <bool CROP>
__device__ __forceinline__ void emit_quad(const Workspace& ws, const StartCand c, const int* dst) {
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
        if (slot >= ws.maxq) {
            atomicOr(ws.counters + CNT_ERR, ERR_QUAD_OVERFLOW);
            return;
        }
        ws.quads_frame[(size_t)c.roi * ws.maxq + slot] = q;
    }
}

// ---- wave-cooperative statistics + polygon approximation ------------------------------------------------------
// One border at a time, all 64 lanes on its stored points (packed x | y << 16).  The points are first staged in LDS
// (one coalesced pass that also yields the bounding box), so every later access of the approximation -- the scans of
// cvApproxPoly (farthest point from the start, farthest point from a chord) strided over the lanes, and the wave-uniform
// reads of slice end points on the dependent chain -- costs an LDS access instead of a global-memory round trip, and the
// wave reductions run on DPP (six VALU steps) instead of twelve ds_bpermute exchanges.  Results are bit-identical to
// trace_core.h::approx_poly_dp / stats_of_points ("first maximum wins" = largest value, then smallest index; the
// perimeter is a sum of float32 values small enough that every partial sum is exact in double, so the summation order
// does not matter).  The slice stack never holds more than DP_MAX_OUT + 1 entries (trace_core.h).
template <int CTRL, int ROWS = 0xf>
__device__ __forceinline__ unsigned dpp_self(unsigned v) {   // lanes the pattern does not reach keep their own value
    return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROWS, 0xf, false);
}
template <int CTRL, int ROWS = 0xf>
__device__ __forceinline__ unsigned dpp_zero(unsigned v) {   // lanes the pattern does not reach receive 0
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWS, 0xf, false);
}
// Reduction ladder of a 64-lane wave: xor 1, xor 2 (quad_perm), half-row mirror, row mirror, then lane 15 of rows 0/2 to
// rows 1/3 and lane 31 to rows 2/3; lane 63 ends up with the result.
#define OCVAR_WAVE_LADDER(STEP) \
    STEP(0xB1, 0xf) STEP(0x4E, 0xf) STEP(0x141, 0xf) STEP(0x140, 0xf) STEP(0x142, 0xa) STEP(0x143, 0xc)

__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
#define STEP(C, R) { const unsigned o = dpp_self<C, R>(v); v = o > v ? o : v; }
    OCVAR_WAVE_LADDER(STEP)
#undef STEP
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#define STEP(C, R) { const unsigned long long o = ((unsigned long long)dpp_self<C, R>((unsigned)(v >> 32)) << 32) | dpp_self<C, R>((unsigned)v); v = o > v ? o : v; }
    OCVAR_WAVE_LADDER(STEP)
#undef STEP
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), 63) << 32) |
           (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, 63);
}
typedef unsigned short us2v __attribute__((ext_vector_type(2)));
// both halves of a packed point at once (v_pk_min_u16 / v_pk_max_u16)
__device__ __forceinline__ unsigned pk_min(unsigned a, unsigned b) { return __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(us2v, a), __builtin_bit_cast(us2v, b))); }
__device__ __forceinline__ unsigned pk_max(unsigned a, unsigned b) { return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(us2v, a), __builtin_bit_cast(us2v, b))); }
__device__ __forceinline__ unsigned wave_pk_min(unsigned v) {
#define STEP(C, R) v = pk_min(v, dpp_self<C, R>(v));
    OCVAR_WAVE_LADDER(STEP)
#undef STEP
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned wave_pk_max(unsigned v) {
#define STEP(C, R) v = pk_max(v, dpp_self<C, R>(v));
    OCVAR_WAVE_LADDER(STEP)
#undef STEP
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ double wave_sum_f64(double v) {
#define STEP(C, R) { const unsigned long long b = (unsigned long long)__double_as_longlong(v); \
                     v += __longlong_as_double((long long)(((unsigned long long)dpp_zero<C, R>((unsigned)(b >> 32)) << 32) | dpp_zero<C, R>((unsigned)b))); }
    OCVAR_WAVE_LADDER(STEP)
#undef STEP
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), 63) << 32) |
                                            (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, 63)));
}

__device__ __forceinline__ int px_of(unsigned p) { return (int)(p & 0xffffu); }
__device__ __forceinline__ int py_of(unsigned p) { return (int)(p >> 16); }

constexpr int LDS_PTS = SLAB_PTS;   // points of one border a wave can stage in LDS
constexpr int DP_DST = 2 * (DP_MAX_OUT + 1);

// Per-wave LDS scratch of the finishing code.
struct WaveScratch {
    DpSlice stack[DP_STACK];
    int dst[DP_DST];
};

// cvApproxPoly on packed points (src: LDS or global memory; the address space is resolved at the inlined call site).
// K32: keys of the maximum searches are (value << 11 | ~index), valid when count <= 1024 and the squared diagonal of the
// bounding box is < 2^21 (every squared distance and every cross product of the border is then < 2^21).
// Returns the vertex count (DP_MAX_OUT+1 = "more than 4 after clean-up"); the vertices are left in dst (LDS).
template <bool K32>
__device__ __forceinline__ int wave_approx_dp(const unsigned* src, int count, double parameter, int* dst, DpSlice* stack) {
    const int lane = threadIdx.x & 63;
    float eps = (float)parameter;
    eps *= eps;
    int top = 0, new_count = 0;
    DpSlice slice = {0, 0}, right = {0, 0};
    int sx = 0, sy = 0;
    bool le_eps = false;
    int pos = 0;
    for (int it = 0; it < 3; it++) {
        pos = (pos + right.start) % count;
        const unsigned sp = uni(src[pos]);
        sx = px_of(sp);
        sy = py_of(sp);
        int max_dist = 0;
        if (K32) {
            unsigned best = 0;
            for (int j = 1 + lane; j < count; j += 64) {
                int q = pos + j;
                q = q >= count ? q - count : q;
                const unsigned p = src[q];
                const int dx = px_of(p) - sx, dy = py_of(p) - sy;
                const unsigned dist = (unsigned)(dx * dx + dy * dy);
                const unsigned key = dist ? (dist << 11) | (~(unsigned)j & 0x7ffu) : 0u;
                best = key > best ? key : best;
            }
            best = wave_max_u32(best);
            max_dist = (int)(best >> 11);
            if (best) right.start = (int)(~best & 0x7ffu);
        } else {
            unsigned long long best = 0;
            for (int j = 1 + lane; j < count; j += 64) {
                int q = pos + j;
                q = q >= count ? q - count : q;
                const unsigned p = src[q];
                const int dx = px_of(p) - sx, dy = py_of(p) - sy;
                const unsigned dist = (unsigned)(dx * dx + dy * dy);
                const unsigned long long key = dist ? ((unsigned long long)dist << 32) | (unsigned)~(unsigned)j : 0ull;
                best = key > best ? key : best;
            }
            best = wave_max_u64(best);
            max_dist = (int)(best >> 32);
            if (best) right.start = (int)~(unsigned)best;
        }
        le_eps = (float)max_dist <= eps;
    }
    if (le_eps) {
        dst[0] = sx;
        dst[1] = sy;
        new_count = 1;
    } else {
        slice.start = pos;
        slice.end = right.start += slice.start;
        right.start -= right.start >= count ? count : 0;
        right.end = slice.start;
        if (right.end < right.start) right.end += count;
        if (lane == 0) {
            stack[0] = right;
            stack[1] = slice;
        }
        top = 2;
    }
    for (int guard = 4 * count + 16; top > 0;) {
        if (--guard < 0) return DP_MAX_OUT + 1;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        --top;
        slice.start = uni(stack[top].start);
        slice.end = uni(stack[top].end);
        const int e = slice.end >= count ? slice.end - count : slice.end;
        const int b = slice.start >= count ? slice.start - count : slice.start;
        const unsigned ep = uni(src[e]), bp = uni(src[b]);
        const int ex = px_of(ep), ey = py_of(ep);
        sx = px_of(bp);
        sy = py_of(bp);
        if (slice.end > slice.start + 1) {
            const int dx = ex - sx, dy = ey - sy;
            int max_dist = 0;
            if (K32) {
                unsigned best = 0;
                for (int i = slice.start + 1 + lane; i < slice.end; i += 64) {
                    const int q = i >= count ? i - count : i;
                    const unsigned p = src[q];
                    int d = (py_of(p) - sy) * dx - (px_of(p) - sx) * dy;
                    d = d < 0 ? -d : d;
                    const unsigned key = d ? ((unsigned)d << 11) | (~(unsigned)i & 0x7ffu) : 0u;
                    best = key > best ? key : best;
                }
                best = wave_max_u32(best);
                max_dist = (int)(best >> 11);
                // (i < 2 * count <= 2048: the index fits its 11 bits)
                if (best) right.start = (int)(~best & 0x7ffu);
            } else {
                unsigned long long best = 0;
                for (int i = slice.start + 1 + lane; i < slice.end; i += 64) {
                    const int q = i >= count ? i - count : i;
                    const unsigned p = src[q];
                    int d = (py_of(p) - sy) * dx - (px_of(p) - sx) * dy;
                    d = d < 0 ? -d : d;
                    const unsigned long long key = d ? ((unsigned long long)(unsigned)d << 32) | (unsigned)~(unsigned)i : 0ull;
                    best = key > best ? key : best;
                }
                best = wave_max_u64(best);
                max_dist = (int)(best >> 32);
                if (best) right.start = (int)~(unsigned)best;
            }
            le_eps = (double)max_dist * max_dist <= (double)eps * ((double)dx * dx + (double)dy * dy);
        } else {
            le_eps = true;
        }
        if (le_eps) {
            if (new_count >= DP_MAX_OUT) return DP_MAX_OUT + 1;
            dst[2 * new_count] = sx;
            dst[2 * new_count + 1] = sy;
            new_count++;
        } else {
            right.end = slice.end;
            slice.end = right.start;
            if (new_count + top + 2 > DP_MAX_OUT) return DP_MAX_OUT + 1;   // every waiting slice ends in >= 1 vertex
            if (lane == 0) {
                stack[top] = right;
                stack[top + 1] = slice;
            }
            top += 2;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    // clean-up of nearly collinear vertices on the closed ring (wave-uniform, <= 8 vertices, all lanes write the same values)
    const int n = new_count;
    int r = n - 1;
    sx = dst[2 * r];
    sy = dst[2 * r + 1];
    if (++r >= n) r = 0;
    int wpos = r;
    int px = dst[2 * r], py = dst[2 * r + 1];
    if (++r >= n) r = 0;
    for (int i = 0; i < n && new_count > 2; i++) {
        const int ex = dst[2 * r], ey = dst[2 * r + 1];
        if (++r >= n) r = 0;
        const int dx = ex - sx, dy = ey - sy;
        int dist = (px - sx) * dy - (py - sy) * dx;
        dist = dist < 0 ? -dist : dist;
        if ((double)dist * dist <= 0.5 * eps * (dx * dx + dy * dy) && dx != 0 && dy != 0) {
            new_count--;
            dst[2 * wpos] = sx = ex;
            dst[2 * wpos + 1] = sy = ey;
            if (++wpos >= n) wpos = 0;
            px = dst[2 * r];
            py = dst[2 * r + 1];
            if (++r >= n) r = 0;
            i++;
            continue;
        }
        dst[2 * wpos] = sx = px;
        dst[2 * wpos + 1] = sy = py;
        if (++wpos >= n) wpos = 0;
        px = ex;
        py = ey;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    return new_count;
}

// Statistics + approximation + filter + publication of one stored border (npts packed points at gsrc in global memory)
// by the whole wave.  STAGE: the points fit the wave's LDS staging area lpts (npts <= LDS_PTS).  Returns whether a quad
// was published.
template <bool CROP, bool STAGE>
__device__ __forceinline__ bool wave_finish_packed(const Workspace& ws, const StartCand c, const PlaneRef& pl, const unsigned* gsrc, int npts,
                                                   unsigned* lpts, WaveScratch* sc) {
    const int lane = threadIdx.x & 63;
    if (npts < 4) return false;
    unsigned lo = 0xffffffffu, hi = 0u;
    if (STAGE) {
        // all loads of the border are issued before the first one is consumed: one memory latency for the staging instead
        // of one per 64 points (the walks of the wave's other lanes stand still meanwhile)
        unsigned v[LDS_PTS / 64];
#pragma unroll
        for (int k = 0; k < LDS_PTS / 64; k++) {
            const int i = lane + 64 * k;
            v[k] = i < npts ? gsrc[i] : 0u;
        }
#pragma unroll
        for (int k = 0; k < LDS_PTS / 64; k++) {
            const int i = lane + 64 * k;
            if (i < npts) {
                lpts[i] = v[k];
                lo = pk_min(lo, v[k]);
                hi = pk_max(hi, v[k]);
            }
        }
    } else {
        for (int i = lane; i < npts; i += 64) {
            const unsigned p = gsrc[i];
            lo = pk_min(lo, p);
            hi = pk_max(hi, p);
        }
    }
    lo = wave_pk_min(lo);
    hi = wave_pk_max(hi);
    const int bw = px_of(hi) - px_of(lo), bh = py_of(hi) - py_of(lo);
    if (!((long long)bw * bh > 500)) return false;   // worth_approximating: the quad filter needs |area| > 500
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    const unsigned* src = STAGE ? lpts : gsrc;
    double per = 0.0;
    for (int i = lane; i < npts; i += 64) {
        const unsigned p = src[i], q = src[i == 0 ? npts - 1 : i - 1];
        const float dx = (float)px_of(p) - (float)px_of(q), dy = (float)py_of(p) - (float)py_of(q);
        per += (double)sqrt_rn(dx * dx + dy * dy);
    }
    per = wave_sum_f64(per);
#ifdef OCVAR_PROF
    if (lane == 0) atomicAdd(reinterpret_cast<unsigned long long*>(ws.counters + CNT_PROF) + (CROP ? 16 : 0) + 7, 1ull);
#endif
    const bool k32 = npts <= 1024 && (long long)bw * bw + (long long)bh * bh < (1ll << 21);
    const int m = k32 ? wave_approx_dp<true>(src, npts, per * 0.02, sc->dst, sc->stack) : wave_approx_dp<false>(src, npts, per * 0.02, sc->dst, sc->stack);
    if (m != 4) return false;
    int q[8];
#pragma unroll
    for (int k = 0; k < 8; k++) q[k] = uni(sc->dst[k]);
    if (!quad_filter(q, pl.img_w, pl.img_h)) return false;
    if (lane == 0) emit_quad<CROP>(ws, c, q);
    return true;
}

// trace_core.h::run_has_earlier_pixel on the tiled mask plane: the masks of the run's pixels lie next to each other in a
// tile row, so the 16 pixels from the start on come with two 16-byte loads (the tile row the start is in and the same
// row of the next tile) instead of one byte load per pixel -- a quarter of the requests for twice the look-ahead, which
// matters on shallow staircases, where the row above reaches over the run only after more than 8 pixels.  The per-pixel
// tests become byte-parallel bit scans.  Same decision rule: pixel k's row above is looked at before its E neighbour; a
// run that is still going on after the look-ahead (or at the plane's edge) is given the benefit of the doubt.
__device__ __forceinline__ bool run_has_earlier_pixel_rows(const uint8_t* nbr, int ns, int cpos, int is_hole) {
    const int x = cpos % ns, y = cpos / ns;
    const int xa = x & ~15;
    const uint8_t* row = nbr + nbr_addr(xa, y, ns);   // 16 masks of tile row (xa .. xa+15, y): 16-byte aligned
    const uint4 A = *reinterpret_cast<const uint4*>(row);
    uint4 B = make_uint4(0u, 0u, 0u, 0u);
    if (xa + 16 < ns) B = *reinterpret_cast<const uint4*>(row + 128);   // the next tile of the same tile row
    // the 16 masks from pixel x on: bytes (x & 15) .. of A:B
    const unsigned w[8] = {A.x, A.y, A.z, A.w, B.x, B.y, B.z, B.w};
    const int ds = (x & 15) >> 2;
    const unsigned bs = (unsigned)(x & 3);
    unsigned m[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        unsigned lo = 0, hi = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {   // ds is 0..3: select without dynamic indexing
            lo = ds == j ? w[i + j] : lo;
            hi = ds == j ? w[i + j + 1 < 8 ? i + j + 1 : 7] : hi;
        }
        m[i] = __builtin_amdgcn_alignbyte(hi, lo, bs);
    }
    int lim = ns - x < 16 ? ns - x : 16;   // pixels that exist
    unsigned long long bad, end;
    const unsigned long long m0 = ((unsigned long long)m[1] << 32) | m[0], m1 = ((unsigned long long)m[3] << 32) | m[2];
    for (int half = 0; half < 2; half++) {
        const unsigned long long v = half ? m1 : m0;
        const int n = lim - 8 * half;
        if (n <= 0) return false;
        const unsigned long long keep = n >= 8 ? ~0ull : ((1ull << (8 * n)) - 1ull);
        if (is_hole) {
            bad = ~v & 0x0404040404040404ull & keep;   // background directly above a pixel of the background run
            end = v & 0x0101010101010101ull & keep;    // E is foreground: the run ends here
        } else {
            bad = v & 0x0e0e0e0e0e0e0e0eull & keep;    // NE, N or NW of a pixel of the foreground run is foreground
            end = ~v & 0x0101010101010101ull & keep;   // E is background: the run ends here
        }
        if (bad | end) {
            const int kb = bad ? __builtin_ctzll(bad) >> 3 : 64, ke = end ? __builtin_ctzll(end) >> 3 : 64;
            return kb <= ke;
        }
    }
    return false;
}

// Tier 1, one lane per start, mask bytes read from global memory (one memory latency per step, 64 starts per wave in
// flight).  It sees every plausible start, and most of them drop out within a few steps (noise, staircase false
// starts).  Returns the route: 0 = dead, 1 = tier 2, 2 = straight to tier 3.
template <bool CROP>
__device__ __forceinline__ int follow_short(const Workspace& ws, const StartCand c) {
    const int BUDGET = SHORT_STEPS;
    const PlaneRef pl = plane_of<CROP>(ws, c.roi);
    if (c.pos <= 0 || c.pos >= pl.plane) return false;
    {
        // tier 1 only decides where a start goes: a look behind (outer starts), then SHORT_STEPS store-free steps.
        // Dead (not the border's first position, isolated pixel, fewer than 4 corner points): dropped.  Closed within the
        // budget or still running: tier 2, which follows again with point storage -- or straight to the wave tier when
        // the border barely turned in SHORT_STEPS steps (an image-sized straight border would only burn tier 2's whole
        // budget before getting there anyway).
        if (earlier_start_behind(pl.nbr, pl.ns, pl.plane, c.pos, c.is_hole, BACK_STEPS)) return 0;
        // crops: no border of a crop is longer than tier 2's budget by much (the crop's own frame border is ~4 sides of
        // <= 260 pixels), so the probe would only repeat what tier 2 does anyway.  A border that starts on the inner edge
        // of the crop's zeroed frame is (nearly always) the background's own outer border, ~800 steps around the crop: the
        // longest walk of its crop, so tier 2 is handed these first (route 3) -- its duration is its longest lane's chain,
        // and a lane that picked such a border up late was that chain.  (Handing them to the wave tier instead was slower:
        // 64 x 64 windows along four sides cost ~150 us of dependent window loads per border.)
        if (CROP) {
            const int x = c.pos % pl.ns, y = c.pos / pl.ns;
            return (x == 1 || y == 1) ? 3 : 1;
        }
        const LeanTrace lt = trace_flat(pl.nbr, pl.ns, pl.plane, c.pos, c.is_hole, nullptr, 0, BUDGET);
        if (lt.status == TRACE_OVERRUN) return lt.npts <= 2 ? 2 : 1;
        if (lt.status != TRACE_OK || lt.npts < 4) return 0;
        return 1;
    }
    return 0;
}

// Tier-2 borders with more corner points than a lane's slab holds (rare: > 1024 corners within the step budget):
// statistics first, then a storing follow into the pool, approximated by the lane itself.  Returns the route (1 = budget
// exhausted after all, 0 = done).
template <bool CROP>
__device__ int follow_overflow(const Workspace& ws, const StartCand c, const PlaneRef& pl) {
    const TraceStats st = trace_border<false, false>(pl.nbr, pl.ns, pl.plane, c.pos, c.is_hole, nullptr, 0, ws.mid_steps);
    if (st.status == TRACE_OVERRUN) return 1;
    if (!worth_approximating(st)) return 0;
    const int need = 2 * st.npts + 2 * (st.npts + 2);
    const long long off = atomicAdd(reinterpret_cast<unsigned long long*>(ws.counters + CNT_POOL_INTS), (unsigned long long)need);
    if (off + need > ws.cap_pool_ints) {
        atomicOr(ws.counters + CNT_ERR, ERR_POOL_OVERFLOW);
        return 0;
    }
    int* pts = ws.pool + off;
    trace_border<true, false>(pl.nbr, pl.ns, pl.plane, c.pos, c.is_hole, pts, st.npts, ws.mid_steps);
    approximate_and_emit<CROP>(ws, c, pl, pts, st.npts, st.perimeter, reinterpret_cast<DpSlice*>(pts + 2 * st.npts));
    return 0;
}

// A single global counter sustains only ~90 atomics per microsecond chip-wide (measured in round 1 on the start lists).
// One ticket per 64 starts plus one list append per 64-start batch was therefore this kernel's whole duration (5.7 M crop
// starts per 2048 frames = 89 k tickets = 1 ms).  A wave now takes T1_TICKET starts per ticket and collects the starts it
// routes on in LDS, appending them T1_OUT at a time with one atomic.
constexpr int T1_TICKET = 512;
constexpr int T1_OUT = 128;

template <bool CROP>
__global__ __launch_bounds__(FOLLOW_THREADS) void follow_kernel(Workspace ws) {
    const StartCand* cands = CROP ? ws.cands_crop : ws.cands_frame;
    int n = ws.counters[CROP ? CNT_CROP_CANDS : CNT_FRAME_CANDS];
    const int cap = CROP ? ws.cap_crop_cands : ws.cap_frame_cands;
    if (n > cap) n = cap;
    int* ticket = ws.counters + (CROP ? CNT_TICKET_C : CNT_TICKET_F);
    const int lane = threadIdx.x & 63;
    const int wave = uni((int)(threadIdx.x >> 6));
    const unsigned long long below = (1ull << lane) - 1ull;
    // Tier 1 sees every plausible start, and most of them are dead at once (run_has_earlier_pixel) or after a handful of
    // steps (a staircase pixel of a slanted edge, image noise) while a few run the whole budget: a wave of 64 fresh starts
    // would spend the full budget with almost all lanes idle.  So fresh starts first get the run test and PRE_STEPS steps
    // each, the survivors are queued per wave in LDS, and the full-budget follow only ever runs on (nearly) full waves of
    // survivors.
    __shared__ StartCand wqueue[FW][128];
    __shared__ StartCand obuf[FW][3][T1_OUT];   // routed starts waiting for their append: to tier 2, to tier 3, to tier 2's "first" list
    int o_n0 = 0, o_n1 = 0, o_n2 = 0;          // wave-uniform fill levels
    auto append = [&](int t, int& o_n) {       // one atomic for everything collected for target t
        if (o_n == 0) return;
        StartCand* list = t == 0 ? (CROP ? ws.mid_crop : ws.mid_frame) : t == 1 ? (CROP ? ws.long_crop : ws.long_frame) : ws.mid_first_crop;
        int* count = ws.counters + (t == 0 ? (CROP ? CNT_MID_C : CNT_MID_F) : t == 1 ? (CROP ? CNT_LONG_C : CNT_LONG_F) : CNT_MID_C_FIRST);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        int qbase = 0;
        if (lane == 0) qbase = atomicAdd(count, o_n);
        qbase = uni(qbase);
        for (int i = lane; i < o_n; i += 64) {
            if (qbase + i < ws.cap_long) list[qbase + i] = obuf[wave][t][i];
            else atomicOr(ws.counters + CNT_ERR, ERR_CAND_OVERFLOW);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        o_n = 0;
    };
    auto collect = [&](int t, int& o_n, bool mine, const StartCand& c) {
        const unsigned long long mask = __ballot(mine);
        if (!mask) return;
        const int cnt = __popcll(mask);
        if (o_n + cnt > T1_OUT) append(t, o_n);
        if (mine) obuf[wave][t][o_n + __popcll(mask & below)] = c;
        o_n += cnt;
    };
    int queued = 0;                // wave-uniform
    int tk_next = 0, tk_end = 0;   // wave-uniform: the part of the list this wave has taken a ticket for
    bool more = true;              // tickets left
    // every iteration consumes 64 starts of a ticket or up to 64 queued survivors: more iterations than that account for
    // means the loop's control flow is broken (see ticket_lane()); report instead of spinning
    for (int guard = 2 * (n / 64 + 2) + 2;; guard--) {
        if (guard < 0) {
            if (lane == 0) atomicOr(ws.counters + CNT_ERR, ERR_TICKET_RUNAWAY);
            break;
        }
        while (more && queued < 64) {
            if (tk_next >= tk_end) {
                int base = 0;
                if (ticket_lane() == 0) base = atomicAdd(ticket, T1_TICKET);
                base = uni(base);
                if (base >= n) {
                    more = false;
                    break;
                }
                tk_next = base;
                tk_end = base + T1_TICKET < n ? base + T1_TICKET : n;
            }
            const int idx = tk_next + lane;
            tk_next += 64;
            bool alive = false;
            StartCand cc;
            cc.roi = 0; cc.pos = 0; cc.is_hole = 0;
            if (idx < tk_end) {
                cc = cands[idx];
                const PlaneRef pl = plane_of<CROP>(ws, cc.roi);
                // (the ROI's own frame border -- the outer start at pixel (1, 1) -- is a rectangle when the ROI's ring of pixels next to
                // its zeroed frame is all set; ring_quads_kernel has then published what the walk would find: nothing to follow)
                const bool ring_done = !cc.is_hole && cc.pos == pl.ns + 1 && (CROP ? ws.ring_crop : ws.ring_frame)[cc.roi] != 0;
                if (!ring_done && cc.pos > 0 && cc.pos < pl.plane && !run_has_earlier_pixel_rows(pl.nbr, pl.ns, cc.pos, cc.is_hole))
                    alive = trace_flat(pl.nbr, pl.ns, pl.plane, cc.pos, cc.is_hole, nullptr, 0, PRE_STEPS).status == TRACE_OVERRUN;
            }
            const unsigned long long mask = __ballot(alive);
            if (alive) wqueue[wave][queued + __popcll(mask & below)] = cc;
            queued += __popcll(mask);
        }
        if (queued == 0) break;
        const int take = queued < 64 ? queued : 64;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        StartCand c;
        c.roi = 0; c.pos = 0; c.is_hole = 0;
        int route = 0;
        if (lane < take) {
            c = wqueue[wave][queued - take + lane];
            route = follow_short<CROP>(ws, c);
        }
        queued -= take;
        if (CROP && route == 1) atomicMin(ws.crop_min_rest + c.roi, c.pos);   // tier 2 walks a crop's earliest start first (see follow_mid_kernel)
        collect(0, o_n0, route == 1, c);
        collect(1, o_n1, route == 2, c);
        if (CROP) collect(2, o_n2, route == 3, c);
    }
    append(0, o_n0);
    append(1, o_n1);
    if (CROP) append(2, o_n2);
}

// ---- the ROI's own frame border without a walk --------------------------------------------------------------------------------
// cvFindContours zeroes the 1-pixel frame of the image it is given (a whole frame, or a crop), so the first border of almost every
// ROI is the outer border of the background region that touches that frame: found at pixel (1, 1), it runs once round the ROI -- 6000
// steps for a 1080p frame, ~700 for a crop: the longest walk of its ROI, a third of the crop pass's steps and, for one frame per call,
// the longest dependent chain of the whole call.  If the ring of pixels next to the zeroed frame (rows 1 and sh-2, columns 1 and sw-2)
// is all set, that border is exactly the rectangle (1,1) (1,sh-2) (sw-2,sh-2) (sw-2,1) -- the follower keeps the zeroed frame on one
// side and the next ring pixel is always there, whatever lies inside -- with these four corner points in this order (the walk runs
// down the left side first).  One wave per ROI checks the ring on the mask plane (bit 4 of a mask byte = the pixel to the west is set,
// bit 0 = the pixel to the east): a handful of independent loads instead of a dependent chain; lane 0 then runs the same
// approximation and filter on the four points as on any stored border and publishes the quad if it is one.  ring[roi] tells tier 1
// to drop the start.  A ring with a hole in it: ring[roi] = 0, the border is walked as before.
template <bool CROP>
__global__ __launch_bounds__(64) void ring_quads_kernel(Workspace ws) {
    int n = CROP ? ws.counters[CNT_CROP_ROIS] : ws.n_frames;
    if (CROP && n > ws.cap_crop_rois) n = ws.cap_crop_rois;
    int* ring = CROP ? ws.ring_crop : ws.ring_frame;
    const int lane = threadIdx.x;
    for (int roi = blockIdx.x; roi < n; roi += gridDim.x) {
        const PlaneRef pl = plane_of<CROP>(ws, roi);
        const int ns = uni(pl.ns), sh = uni(pl.sh), sw = uni(pl.img_w) & ~1;
        bool bad = sw < 8 || sh < 8;
        if (!bad) {
            // rows 1 and sh-2: pixel x in [1, sw-2] is bit 4 of mask byte x+1
            const int tiles = ns >> 4;
            for (int i = lane; i < 2 * tiles; i += 64) {
                const int y = i < tiles ? 1 : sh - 2, xt = i < tiles ? i : i - tiles;
                const uint4 v = *reinterpret_cast<const uint4*>(pl.nbr + nbr_addr(xt << 4, y, ns));
                const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const int xb = (xt << 4) + k;   // mask byte of pixel xb: its west neighbour is pixel xb-1
                    if (xb >= 2 && xb <= sw - 1 && !((w[k >> 2] >> (8 * (k & 3) + 4)) & 1u)) bad = true;
                }
            }
            // columns 1 and sw-2, rows 1 .. sh-2: bit 4 of mask(2, y), bit 0 of mask(sw-3, y)
            for (int y = 1 + lane; y <= sh - 2; y += 64) {
                const unsigned a = pl.nbr[nbr_addr(2, y, ns)], b = pl.nbr[nbr_addr(sw - 3, y, ns)];
                if (!((a >> 4) & 1u) || !(b & 1u)) bad = true;
            }
        }
        const bool clean = __ballot(bad) == 0;
        if (lane == 0) {
            ring[roi] = clean ? 1 : 0;
            if (clean && (long long)(sw - 3) * (sh - 3) > 500) {   // (worth_approximating: the quad filter needs |area| > 500)
                const int pts[8] = {1, 1, 1, sh - 2, sw - 2, sh - 2, sw - 2, 1};
                const TraceStats st = stats_of_points(pts, 4);
                StartCand c;
                c.roi = roi;
                c.pos = ns + 1;
                c.is_hole = 0;
                DpSlice stack[DP_STACK];
                approximate_and_emit<CROP>(ws, c, pl, pts, 4, st.perimeter, stack);
            }
        }
    }
}

// ---- the follower's step from a table (tier 2) ------------------------------------------------------------------------------
// trace_core.h::flat_step_t spends ~70 vector instructions per step: the rotate / count-trailing-zeros / rotate-back that turn
// (arrival direction, neighbour mask) into (exit direction, neighbours passed on the way), two table extractions for the
// direction's dx and dy, the scan position y * ns + x carried next to x and y, the tiled address from scratch, and a chain of
// selects.  The step is rebuilt around three ideas:
//  * the mask only matters through t = the number of zero neighbours examined before the next border pixel (a byte replicate,
//    a funnel shift and a find-first-bit), and everything the step decides is a function of (first examined direction, t): 64
//    states, ONE lookup in a 256-byte LDS table: entry = (dx + 1) | (dy + 1) << 16 | next first direction << 5 | exit
//    direction << 10 | passed-E << 30 | passed-W << 31.  (A table over (direction, mask) itself -- 8 KB per workgroup -- saved
//    four more instructions and cost a workgroup of occupancy next to the binarise kernels: slower.)
//  * the position is the packed point x | y << 16 itself (what is parked and stored anyway): one add3 moves it, and because y
//    is the major half, packed points compare like scan positions y * ns + x -- the "a position of this border before its
//    start" tests and the closing test need no scan position at all (passed-W / passed-E sit in the sign bits: an AND with the
//    signed distance to the start position decides).
//  * a walk that ends is never stepped again, so nothing is guarded.
// ~42 vector instructions per step instead of ~70; same states, same points, same status as flat_step_t, step for step
// (tests/test_gpu_parity.py compares every quad and candidate with the oracle; tools/fuzz_parity.py sweeps).
constexpr unsigned LW_EXIT = 0x1c00u;        // exit direction << 10
constexpr unsigned LW_DELTA = 0x00030003u;   // (dx + 1) | (dy + 1) << 16

__device__ __forceinline__ unsigned lean_table_entry(unsigned idx) {   // idx = t << 3 | first examined direction
    const unsigned from = idx & 7u, t = idx >> 3;
    const unsigned e = (from + t) & 7u;
    const unsigned dx = (unsigned)(step_dx((int)e) + 1), dy = (unsigned)(step_dy((int)e) + 1);
    // the t neighbours from, from + 1, ... were examined and found zero: was W (direction 4) / E (direction 0) among them?
    const unsigned pw = ((12u - from) & 7u) < t, pe = ((8u - from) & 7u) < t;
    return dx | (dy << 16) | (((e + 5u) & 7u) << 5) | (e << 10) | (pe << 30) | (pw << 31);
}

struct LeanWalk {
    unsigned xy;        // current pixel, x | y << 16
    unsigned m;         // its neighbour mask
    unsigned from;      // first direction examined from it = arrival direction + 1 (mod 8)
    unsigned pe;        // previous exit direction << 10: a corner point wherever the exit direction changes
    unsigned xy0, xy1;  // the border's first pixel and the pixel the closing step comes from
    unsigned cxy;       // the start's scan position as a packed point
    int npts, step, status;   // status: -1 while running, then a TraceStatus
};

__device__ __forceinline__ void lean_begin(LeanWalk& w, const uint8_t* nbr, int ns, int cpos, int is_hole) {
    const int i0 = cpos - is_hole;
    const int x = i0 % ns, y = i0 / ns;
    w.xy = w.xy0 = (unsigned)x | ((unsigned)y << 16);
    w.cxy = w.xy0 + (unsigned)is_hole;   // (a hole's start position is the pixel east of its first pixel, same row)
    w.npts = 0;
    w.step = 0;
    w.status = -1;
    w.xy1 = 0;
    w.pe = 0;
    w.from = 0;
    w.m = nbr[nbr_addr(x, y, ns)];
    if (w.m == 0) {   // single-pixel domain (only reachable for outer borders)
        w.status = TRACE_SINGLE;
        w.npts = 1;
        return;
    }
    const int s = first_cw(w.m, (is_hole ? 0 : 4) - 1);   // direction of the border's last pixel seen from its first
    w.xy1 = w.xy0 + (unsigned)(step_dx(s) + 65536 * step_dy(s));
    w.pe = (unsigned)(s ^ 4) << 10;                        // prev_s = s ^ 4
    w.from = (unsigned)(s + 1) & 7u;
}

// One step of a running walk.  tab: the 64-entry table in LDS; base / nt / last: the walk's plane (its first byte, tiles per
// tile row = ns >> 4, its last byte offset); park(xy): the caller parks the pixel being left (every step; only a corner point
// advances npts afterwards).
template <class Park>
__device__ __forceinline__ void lean_step(LeanWalk& w, const unsigned* tab, const uint8_t* base, unsigned nt, unsigned last, Park&& park) {
    const unsigned mm = __builtin_amdgcn_perm(w.m, w.m, 0u);                       // the mask in all four bytes
    const unsigned t = (unsigned)__builtin_ctz(__builtin_amdgcn_alignbit(mm, mm, w.from));   // zero neighbours before the next border pixel
    const unsigned ent = tab[(t << 3) + w.from];
    const int d = (int)(w.xy - w.cxy);                             // < 0: this pixel's west side precedes the start
    const unsigned nf = ((ent << 1) & (unsigned)(d + 1)) | (ent & (unsigned)d);   // sign: an earlier position of this border was passed
    const unsigned nxy = w.xy + (ent & LW_DELTA) - 0x10001u;
    const bool closes = (nxy == w.xy0) & (w.xy == w.xy1);
    const unsigned e = ent & LW_EXIT;
    const bool emit = e != w.pe;
    // tiled address of the next pixel (hd.h::nbr_addr on the packed point).  A consistent plane never sends a walk outside
    // itself; should one be inconsistent the offset is clamped (no access outside the plane), the walk runs into its budget and
    // the wave tier, which checks every coordinate, reports it.
    unsigned off = __umul24(nxy >> 19, nt) + ((nxy >> 4) & 0xfffu);
    off = (off << 7) | (nxy & 15u);
    off |= (nxy >> 12) & 0x70u;
    off = off < last ? off : last;
    const unsigned m4 = base[off];
    park(w.xy);
    w.npts += emit ? 1 : 0;
    w.status = (int)nf < 0 ? (int)TRACE_NOT_FIRST : closes ? (int)TRACE_OK : m4 == 0u ? (int)TRACE_OVERRUN : -1;
    w.step += 1;
    w.xy = nxy;
    w.pe = e;
    w.from = (ent >> 5) & 7u;
    w.m = m4;
}

// ---- Tier 2: one lane per surviving border, 64 independent walks per wave, lanes refilled as they finish -------------
// A walk is a chain of dependent byte loads, so a wave's time is its longest walk; with one batch of 64 starts per wave
// most lanes would sit idle behind the longest border (a crop holds a ~900-step border next to 100-step ones).  Here a
// lane whose walk has ended is retired every MID_BLOCK steps -- routed on, or approximated by the whole wave from its
// slab -- and takes the next start from the list, so the lanes stay busy until the list is empty.
constexpr int MID_BLOCK = 32;
// A walk's corner points are not stored to its slab step by step -- 64 lanes x one dword in 64 different cache lines per
// step made the stores, not the dependent mask loads, the bulk of this kernel's time (1.7 of 2.8 ms per 2048 frames) --
// but parked in LDS (row of POINT_ROW dwords per lane: up to MID_BLOCK points + the dummy slot of steps without a point;
// the odd row length spreads the lanes over the banks) and appended to the slabs after every block of steps, two lanes'
// rows per store instruction, each a contiguous run of up to 128 bytes.
constexpr int POINT_ROW = MID_BLOCK + 1;
constexpr int FLUSH_W = MID_BLOCK <= 16 ? 16 : 32;   // lanes that carry one walk's parked points in a flush (>= MID_BLOCK)
static_assert(FLUSH_W >= MID_BLOCK && 64 % FLUSH_W == 0, "a group of lanes covers a whole row of parked points");

// Crops, exact pruning: of a crop's quads only the one that starts earliest is used (cvarGetSquare keeps the sequence's last
// quad, opencvar.cpp:401-430), so a border that starts after a quad the crop already has can never matter -- it is not
// walked, and a walked one is not approximated.  To have that quad early, a crop's borders are walked in two launches:
// phase 1 takes the starts on the crop's frame (nearly always the background's own border: the earliest start, rarely a
// valid quad) and each crop's earliest other start (nearly always the marker's outline: the quad that wins); phase 2 takes
// what is left and skips every start behind its crop's best quad -- the outline's other side, the code cells, and the
// staircase starts inside them: about half of a crop's steps.  Frames keep all their quads: one launch, phase 0.
template <bool CROP>
__global__ __launch_bounds__(FOLLOW_THREADS) void follow_mid_kernel(Workspace ws, int phase) {
    __shared__ WaveScratch scratch[FW];
    __shared__ unsigned parked[FW][64 * POINT_ROW];
    __shared__ unsigned step_tab[64];
    if (threadIdx.x < 64u) step_tab[threadIdx.x] = lean_table_entry(threadIdx.x);
    __syncthreads();
    const StartCand* cands = CROP ? ws.mid_crop : ws.mid_frame;
    StartCand* longs = CROP ? ws.long_crop : ws.long_frame;
    int n = ws.counters[CROP ? CNT_MID_C : CNT_MID_F];
    if (n > ws.cap_long) n = ws.cap_long;
    // crops, phase 1: the starts on the crop frames (the longest walks) come first in the ticket order
    int n_first = (CROP && phase != 2) ? ws.counters[CNT_MID_C_FIRST] : 0;
    if (n_first > ws.cap_long) n_first = ws.cap_long;
    n += n_first;
    int* ticket = ws.counters + (CROP ? (phase == 2 ? CNT_TICKET_MC2 : CNT_TICKET_MC) : CNT_TICKET_MF);
    int* n_long = ws.counters + (CROP ? CNT_LONG_C : CNT_LONG_F);
    const int lane = threadIdx.x & 63;
    const int wave = uni((int)(threadIdx.x >> 6));
    const unsigned long long below = (1ull << lane) - 1ull;
    unsigned* wave_slabs = reinterpret_cast<unsigned*>(ws.slab) + ((size_t)blockIdx.x * blockDim.x + uni((int)(threadIdx.x & ~63u))) * SLAB_STRIDE;
    static_assert(64 * POINT_ROW >= LDS_PTS, "the staging area of the approximation aliases the (flushed) point rows");
    unsigned* my_row = parked[wave] + lane * POINT_ROW;
    const unsigned* wave_rows = parked[wave];
    int flushed = 0;     // points of this lane's walk already in its slab
    const int budget = ws.mid_steps;
    bool have = false;   // this lane holds a start whose walk has not been retired
    bool more = true;    // wave-uniform: the list may still hold starts
    StartCand c;
    c.roi = 0; c.pos = 0; c.is_hole = 0;
    PlaneRef pl = plane_of<CROP>(ws, 0);
    LeanWalk w;
    w.xy = w.m = w.from = w.pe = w.xy0 = w.xy1 = w.cxy = 0u;
    w.npts = w.step = 0;
    w.status = TRACE_NOT_FIRST;
    unsigned pl_nt = 1u, pl_last = 0u;   // tiles per tile row and last byte offset of the lane's plane
    // every round hands out at least one start or retires at least one walk after <= budget / MID_BLOCK rounds of stepping
    long long guard = ((long long)n + 64) * (budget / MID_BLOCK + 5) + 64;
    PROF_DECL();
    for (;;) {
        if (--guard < 0) {
            if (lane == 0) atomicOr(ws.counters + CNT_ERR, ERR_TICKET_RUNAWAY);
            break;
        }
        // hand idle lanes the next starts of the list (one ticket fetch for all of them; taking list entries in bulk per wave
        // instead was slower: lanes wait for the next round whenever the wave's range runs out, and the ranges unbalance the tail)
        const unsigned long long pt0 = PROF_NOW();
        // (crops: a start may be skipped -- not this phase's, or behind its crop's best quad -- so idle lanes get up to four
        // hand-outs per round; every hand-out consumes list entries, which the guard counts)
        for (int tries = 0; tries < (CROP ? 4 : 1); tries++) {
            const unsigned long long idle = __ballot(!have);
            if (!(more && idle)) break;
            if (tries > 0) --guard;
            const int cnt = __popcll(idle);
            int base = 0;
            if (ticket_lane() == 0) base = atomicAdd(ticket, cnt);
            base = uni(base);
            if (base + cnt >= n) more = false;
            if (!have) {
                const int idx = base + __popcll(idle & below);
                if (idx < n) {
                    c = (CROP && idx < n_first) ? ws.mid_first_crop[idx] : cands[idx - n_first];
                    bool take = true;
                    if (CROP) {
                        const int earliest = ws.crop_min_rest[c.roi];
                        const bool beaten = (unsigned)(ws.best_crop[c.roi] >> 32) < (unsigned)c.pos;
                        if (phase == 1) take = idx < n_first || c.pos == earliest;
                        else if (phase == 2) take = c.pos != earliest && !beaten;
                        else take = !beaten;   // phase 0 on crops (Workspace::crop_phases == 1): one launch, pruning only by what happens to be finished
                    }
                    pl = plane_of<CROP>(ws, c.roi);
                    if (take && c.pos > 0 && c.pos < pl.plane) {
                        lean_begin(w, pl.nbr, pl.ns, c.pos, c.is_hole);
                        pl_nt = (unsigned)(pl.ns >> 4);
                        pl_last = (unsigned)nbr_plane_bytes(pl.ns, pl.sh) - 1u;
                        have = true;
                        flushed = 0;
                    }
                }
            }
        }
        if (__ballot(have) == 0) break;
        const unsigned long long pt1 = PROF_NOW();
        PROF_ADD(0, pt1 - pt0);
        for (int k = 0; k < MID_BLOCK; k++) {
            const bool running = have && w.status < 0;
            if (__ballot(running) == 0) break;
            PROF_ADD(4, 1);
            PROF_ADD(5, __popcll(__ballot(running)));
            if (running)
                lean_step(w, step_tab, pl.nbr, pl_nt, pl_last, [&](unsigned xy) {
                    // every step writes its pixel at the running count; only a corner point advances the count, so a step
                    // without one is overwritten by the next (beyond the slab's capacity: the row's spare slot)
                    const int slot = w.npts - flushed;
                    my_row[slot < MID_BLOCK ? slot : MID_BLOCK] = xy;
                });
        }
        // the step budget is checked here, once per block, instead of in every step (a walk may overshoot it by up to
        // MID_BLOCK - 1 steps; it is handed to the wave tier either way, which follows the border again from its start)
        if (have && w.status < 0 && w.step >= budget) w.status = TRACE_OVERRUN;
        const unsigned long long pt2 = PROF_NOW();
        PROF_ADD(1, pt2 - pt1);
        {   // append the parked points to the slabs: every group of FLUSH_W lanes carries one walk's row per store
            const int stored = w.npts < SLAB_PTS ? w.npts : SLAB_PTS;
            const int pend = have ? stored - flushed : 0;
            unsigned long long fm = __ballot(pend > 0);
            if (fm) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            const int grp = lane / FLUSH_W, k = lane % FLUSH_W;
            while (fm) {
                int L = 0, cnt = 0, off = 0;
#pragma unroll
                for (int g = 0; g < 64 / FLUSH_W; g++) {
                    const int Lg = fm ? __ffsll((long long)fm) - 1 : 0;
                    const int cg = fm ? __builtin_amdgcn_readlane(pend, Lg) : 0;
                    const int og = __builtin_amdgcn_readlane(flushed, Lg);
                    fm &= fm - 1;   // (0 & -1 stays 0)
                    L = grp == g ? Lg : L;
                    cnt = grp == g ? cg : cnt;
                    off = grp == g ? og : off;
                }
                if (k < cnt) wave_slabs[(size_t)L * SLAB_STRIDE + off + k] = wave_rows[L * POINT_ROW + k];
            }
            flushed += pend;
        }
        const unsigned long long pt3 = PROF_NOW();
        PROF_ADD(2, pt3 - pt2);
        // retire the walks that have ended
        int route = 0, slab_npts = 0;
#ifdef OCVAR_PROF
        {   // where the steps went: walks that were not their border's first position / closed borders / budget exhausted
            const bool done = have && w.status >= 0;
            for (int st_ = 0; st_ < 3; st_++) {
                const int want = st_ == 0 ? (int)TRACE_NOT_FIRST : st_ == 1 ? (int)TRACE_OK : (int)TRACE_OVERRUN;
                const bool mine = done && w.status == want;
                PROF_ADD(9 + 2 * st_, __popcll(__ballot(mine)));
                unsigned steps_ = mine ? (unsigned)w.step : 0u;
                for (int o_ = 32; o_ > 0; o_ >>= 1) steps_ += (unsigned)__shfl_xor((int)steps_, o_, 64);
                PROF_ADD(8 + 2 * st_, steps_);
            }
            bool useless = false;   // crops: a closed border that starts after the crop's best quad so far
            if (CROP && done && w.status == TRACE_OK) useless = (unsigned)(ws.best_crop[c.roi] >> 32) < (unsigned)c.pos;
            unsigned us_ = useless ? (unsigned)w.step : 0u;
            for (int o_ = 32; o_ > 0; o_ >>= 1) us_ += (unsigned)__shfl_xor((int)us_, o_, 64);
            PROF_ADD(14, us_);
            PROF_ADD(15, __popcll(__ballot(done && w.status == TRACE_OK && w.npts < 4)));
        }
#endif
        if (have && w.status >= 0) {
            if (w.status == TRACE_OVERRUN) route = 1;   // budget exhausted: a longer border (tier 3)
            else if (w.status == TRACE_OK && w.npts >= 4) {
                if (w.npts <= SLAB_PTS) {
                    route = 3;   // finished by the whole wave below
                    slab_npts = w.npts;
                } else {
                    route = follow_overflow<CROP>(ws, c, pl);
                }
            }
            have = false;
        }
        // borders stored in the lanes' slabs: statistics, approximation and filter by the whole wave, one at a time.  While a
        // border is finished the wave's other walks stand still (at the benchmark's launch size finishing is half of this kernel's
        // cycles on frames and a third on crops), so nothing is fetched from memory per border that a lane already holds: the
        // ROI's size comes from the walking lane's registers, and every retiring lane loads its crop's best start once, all of
        // them together, before the loop (a value a moment old only prunes a little less: the winner is chosen by atomicMin).
        unsigned long long todo = __ballot(route == 3);
        unsigned my_best = 0xffffffffu;
        if (CROP && route == 3) my_best = (unsigned)(ws.best_crop[c.roi] >> 32);
        if (todo) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");   // each lane stored its own slab; now every lane reads them
        while (todo) {
            const int L = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            StartCand cl;
            cl.roi = __builtin_amdgcn_readlane(c.roi, L);
            cl.pos = __builtin_amdgcn_readlane(c.pos, L);
            cl.is_hole = __builtin_amdgcn_readlane(c.is_hole, L);
            const int nl = __builtin_amdgcn_readlane(slab_npts, L);
            const unsigned* sl = wave_slabs + (size_t)L * SLAB_STRIDE;
            PlaneRef pll = pl;   // (of the finish only the ROI's size is used: the quad filter's border rule)
            pll.img_w = __builtin_amdgcn_readlane(pl.img_w, L);
            pll.img_h = __builtin_amdgcn_readlane(pl.img_h, L);
            // crops: a border that starts behind the crop's best quad cannot replace it
            const bool beaten = CROP && (unsigned)__builtin_amdgcn_readlane((int)my_best, L) < (unsigned)cl.pos;
            if (!beaten) wave_finish_packed<CROP, true>(ws, cl, pll, sl, nl, parked[wave], &scratch[wave]);
            PROF_ADD(6, 1);
        }
        PROF_ADD(3, PROF_NOW() - pt3);
        // budget exhausted: queue for the wave tier
        const unsigned long long mask = __ballot(route == 1);
        if (mask) {
            int qbase = 0;
            const int leader = __ffsll((long long)mask) - 1;
            if (lane == leader) qbase = atomicAdd(n_long, __popcll(mask));
            qbase = __builtin_amdgcn_readlane(qbase, leader);
            if (route == 1) {
                const int slot = qbase + __popcll(mask & below);
                if (slot < ws.cap_long) longs[slot] = c;
                else atomicOr(ws.counters + CNT_ERR, ERR_CAND_OVERFLOW);
            }
        }
    }
    PROF_FLUSH(CROP ? 16 : 0);
}

// ---- Phase B: one wave per long border, walking inside an LDS tile cache -------------------------------------
// The follower's step needs the neighbour mask of the pixel it just moved to: a dependent byte load, i.e. one
// HBM/L2 latency per step when done from global memory.  Here the wave cooperatively copies a TILE x TILE window
// of the mask plane around the current pixel into LDS (each lane one row, 16-byte loads) and all lanes walk the
// same border redundantly (wave-uniform control flow), so a step costs an LDS read; the window is re-centred when
// the walk leaves it.  Same stepping rules as trace_core.h::trace_border.
struct TileCache {
    uint8_t* lds;          // TILE*TILE bytes of this wave
    const uint8_t* nbr;
    int ns, sh;            // ns is a multiple of 16, the plane base is 16-byte aligned
    int tx0, ty0;          // window origin (tx0 multiple of 16, may be negative); wave-uniform
};

// (bx,by): direction of travel; the window is pushed ahead so that a straight run re-centres as rarely as possible
__device__ __forceinline__ void tile_load(TileCache& t, int x, int y, int bx = 0, int by = 0) {
    const int lane = threadIdx.x & 63;
    t.tx0 = (x + 20 * bx - TILE / 2 + 8) & ~15;   // keeps x-20..x+27 inside for bx = 0, up to 55 pixels ahead otherwise
    t.ty0 = y + 28 * by - TILE / 2;
    const int row = t.ty0 + lane;
    uint4* dst = reinterpret_cast<uint4*>(t.lds + lane * TILE);
#pragma unroll
    for (int cch = 0; cch < TILE / 16; cch++) {
        const int cx = t.tx0 + 16 * cch;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (row >= 0 && row < t.sh && cx >= 0 && cx + 16 <= t.ns) v = *reinterpret_cast<const uint4*>(t.nbr + nbr_addr(cx, row, t.ns));
        dst[cch] = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
}

__device__ __forceinline__ unsigned tile_get(TileCache& t, int x, int y) {
    if ((unsigned)(x - t.tx0) >= (unsigned)TILE || (unsigned)(y - t.ty0) >= (unsigned)TILE) tile_load(t, x, y);
    return uni((unsigned)t.lds[(y - t.ty0) * TILE + (x - t.tx0)]);
}

// Lean follower on the tile cache.  All 64 lanes walk the same border and the walker's state (position, direction,
// mask, counters) is wave-uniform -- kept in SGPRs, stepped by the scalar ALU, branches scalar.  Corner points are
// parked packed (x | y << 16) in a lane of hp (lane = point index mod 64) and written 64 at a time, one coalesced
// 256-byte store, up to max_pts (the count keeps running beyond it); statistics are recomputed from the points afterwards.
__device__ LeanTrace trace_lean_tiled(TileCache& t, int cpos, int is_hole, int* out, int max_pts, int max_steps) {
    const int lane = threadIdx.x & 63;
    LeanTrace r;
    r.status = TRACE_OK;
    r.npts = 0;
    r.steps = 0;
    const int ns = t.ns;
    const int i0 = cpos - is_hole;
    const int x0 = i0 % ns, y0 = i0 / ns;
    int x = x0, y = y0;
    unsigned m = tile_get(t, x, y);
    if (m == 0) {
        r.status = TRACE_SINGLE;
        r.npts = 1;
        return r;
    }
    int s = first_cw(m, (is_hole ? 0 : 4) - 1);
    const int x1 = x0 + step_dx(s), y1 = y0 + step_dy(s);
    int prev_s = s ^ 4;
    int straight = 0;  // consecutive straight steps through identical masks (a long run is likely once this is 2)
    int step = 0;
    unsigned hp = 0;
    unsigned* out1 = reinterpret_cast<unsigned*>(out);
    for (;; step++) {
        if (step >= max_steps) {
            r.status = TRACE_OVERRUN;
            break;
        }
        const int from = (s + 1) & 7;
        const int tz = __builtin_ctz(((m * 0x101u) >> from) & 0xffu);
        const int e = (from + tz) & 7;
        const unsigned passed = ((((1u << tz) - 1u) * 0x101u) << from) >> 8;
        const int idx = y * ns + x;
        if (((passed & 0x10u) && idx < cpos) || ((passed & 1u) && idx + 1 < cpos)) {
            r.status = TRACE_NOT_FIRST;
            break;
        }
        if (e != prev_s) {
            if (r.npts < max_pts) {
                const int slot = r.npts & 63;
                hp = lane == slot ? ((unsigned)x | ((unsigned)y << 16)) : hp;   // v_cndmask: no EXEC change
                if (slot == 63) out1[r.npts - 63 + lane] = hp;
            }
            r.npts++;
            prev_s = e;
        }
        const int ddx = step_dx(e), ddy = step_dy(e);
        const int px = x, py = y;
        x += ddx;
        y += ddy;
        if (x == x0 && y == y0 && px == x1 && py == y1) break;
        if ((unsigned)x >= (unsigned)ns || (unsigned)y >= (unsigned)t.sh) {
            r.status = TRACE_OVERRUN;
            break;
        }
        unsigned m4 = tile_get(t, x, y);
        if (m4 == 0) {
            r.status = TRACE_OVERRUN;
            break;
        }
        if (e == (s ^ 4) && m4 == m) {
            // straight through a pixel whose successor has the same mask: the state repeats (trace_core.h)
            if (++straight >= 2) {
                // a run: the 64 lanes inspect the next 64 pixels of the line at once; the first lane that sees another
                // mask, the end of the window or the closing step decides how far the walk jumps
                bool closed = false, bad = false;
                for (;;) {
                    const int qx = x + (lane + 1) * ddx, qy = y + (lane + 1) * ddy;
                    const bool in_tile = (unsigned)(qx - t.tx0) < (unsigned)TILE && (unsigned)(qy - t.ty0) < (unsigned)TILE;
                    const unsigned mq = in_tile ? t.lds[(qy - t.ty0) * TILE + (qx - t.tx0)] : 256u;
                    const bool closes = qx == x0 && qy == y0 && qx - ddx == x1 && qy - ddy == y1;
                    const unsigned long long close_mask = __ballot(closes);
                    const unsigned long long stop = close_mask | __ballot(mq != m);
                    const int k = stop ? __ffsll((long long)stop) - 1 : 64;   // pixels (x,y)+1..k carry mask m
                    // scan positions are monotonic along a line: the two ends decide
                    const int pa = y * ns + x, pb = (y + k * ddy) * ns + x + k * ddx;
                    if (((passed & 0x10u) && (pa < cpos || pb < cpos)) || ((passed & 1u) && (pa + 1 < cpos || pb + 1 < cpos))) {
                        r.status = TRACE_NOT_FIRST;
                        bad = true;
                        break;
                    }
                    x += k * ddx;
                    y += k * ddy;
                    step += k;
                    if (step >= max_steps) {
                        r.status = TRACE_OVERRUN;
                        bad = true;
                        break;
                    }
                    if (k == 64) continue;
                    if ((close_mask >> k) & 1ull) {
                        closed = true;
                        break;
                    }
                    const unsigned stop_m = (unsigned)__builtin_amdgcn_readlane((int)mq, k);
                    if (stop_m == 256u) {  // end of the window: re-centre ahead and keep running
                        if ((unsigned)(x + ddx) >= (unsigned)ns || (unsigned)(y + ddy) >= (unsigned)t.sh) {
                            r.status = TRACE_OVERRUN;
                            bad = true;
                            break;
                        }
                        tile_load(t, x + ddx, y + ddy, ddx, ddy);
                        continue;
                    }
                    x += ddx;   // land on the first pixel with a different mask
                    y += ddy;
                    step++;
                    m4 = stop_m;
                    break;
                }
                if (bad || closed) break;
                if (m4 == 0) {
                    r.status = TRACE_OVERRUN;
                    break;
                }
                straight = 0;
            }
        } else {
            straight = 0;
        }
        m = m4;
        s = e ^ 4;
    }
    if (r.status == TRACE_OK) {   // the points still parked in lanes
        const int stored = r.npts < max_pts ? r.npts : max_pts;
        const int rem = stored & 63;
        if (lane < rem) out1[stored - rem + lane] = hp;
    }
    r.steps = step;
    return r;
}

template <bool CROP>
__global__ __launch_bounds__(FOLLOW_THREADS) void follow_long_kernel(Workspace ws) {
    __shared__ __attribute__((aligned(16))) uint8_t tiles[FW][TILE * TILE];
    __shared__ WaveScratch scratch[FW];
    __shared__ unsigned staged[FW][LDS_PTS];
    const StartCand* longs = CROP ? ws.long_crop : ws.long_frame;
    int n = ws.counters[CROP ? CNT_LONG_C : CNT_LONG_F];
    if (n > ws.cap_long) n = ws.cap_long;
    int* ticket = ws.counters + (CROP ? CNT_TICKET_LC : CNT_TICKET_LF);
    const int lane = threadIdx.x & 63;
    const int wave = uni((int)(threadIdx.x >> 6));
    // this wave's point space in global memory
    int* slab = ws.slab3 + ((size_t)blockIdx.x * FW + wave) * SLAB3_STRIDE;
    // a ticket is issued at most n + (waves of the grid) times: more iterations than that means the loop's control flow is
    // broken (see ticket_lane()); report instead of spinning
    for (int guard = n + 2;; guard--) {
        if (guard < 0) {
            if (lane == 0) atomicOr(ws.counters + CNT_ERR, ERR_TICKET_RUNAWAY);
            break;
        }
        int idx = 0;
        if (ticket_lane() == 0) idx = atomicAdd(ticket, 1);
        idx = uni(idx);
        if (idx >= n) break;
        StartCand c = longs[idx];
        c.roi = uni(c.roi);
        c.pos = uni(c.pos);
        c.is_hole = uni(c.is_hole);
        const PlaneRef pl = plane_of<CROP>(ws, c.roi);
        // crops: a border that starts behind the crop's best quad cannot replace it (see follow_mid_kernel); nothing below
        // depends on lt then (status TRACE_NOT_FIRST: dropped)
        const bool beaten = CROP && uni((unsigned)(ws.best_crop[c.roi] >> 32)) < (unsigned)c.pos;
        TileCache t;
        t.lds = tiles[wave];
        t.nbr = pl.nbr;
        t.ns = uni(pl.ns);
        t.sh = uni(pl.sh);
        t.tx0 = t.ty0 = -(1 << 28);
        LeanTrace lt;
        lt.status = TRACE_NOT_FIRST;
        lt.npts = 0;
        lt.steps = 0;
        if (!beaten) lt = trace_lean_tiled(t, c.pos, c.is_hole, slab, SLAB3_PTS, 4 * uni(pl.plane) + 16);
        // (no `continue` below: one back-edge, scalar conditions -- see uni())
        if (lt.status == TRACE_OVERRUN) {
            if (lane == 0) atomicOr(ws.counters + CNT_ERR, ERR_TRACE_OVERRUN);
        } else if (lt.status == TRACE_OK && lt.npts >= 4) {
            if (lt.npts > SLAB3_PTS) {
                // more corner points than the wave's slab holds (a percolating noise cluster, a sawtooth the size of the
                // frame): the count is known now, so take the points' space from the pool and follow once more, storing
                // all.  Out of pool: report, never truncate silently.
                const long long need = (long long)lt.npts + 64;
                unsigned long long off = 0;
                if (lane == 0) off = atomicAdd(reinterpret_cast<unsigned long long*>(ws.counters + CNT_POOL_INTS), (unsigned long long)need);
                off = uni(off);
                if ((long long)off + need > ws.cap_pool_ints) {
                    if (lane == 0) atomicOr(ws.counters + CNT_ERR, ERR_POOL_OVERFLOW);
                } else {
                    int* big = ws.pool + off;
                    t.tx0 = t.ty0 = -(1 << 28);
                    const LeanTrace lt2 = trace_lean_tiled(t, c.pos, c.is_hole, big, lt.npts, 4 * uni(pl.plane) + 16);
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                    if (lt2.status == TRACE_OK && lt2.npts == lt.npts) {
                        wave_finish_packed<CROP, false>(ws, c, pl, reinterpret_cast<const unsigned*>(big), lt2.npts, staged[wave], &scratch[wave]);
                    } else if (lane == 0) {
                        atomicOr(ws.counters + CNT_ERR, ERR_TRACE_OVERRUN);
                    }
                }
            } else {
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");   // points stored by other lanes of this wave (same CU: L1 is coherent)
                if (lt.npts <= LDS_PTS) wave_finish_packed<CROP, true>(ws, c, pl, reinterpret_cast<const unsigned*>(slab), lt.npts, staged[wave], &scratch[wave]);
                else wave_finish_packed<CROP, false>(ws, c, pl, reinterpret_cast<const unsigned*>(slab), lt.npts, staged[wave], &scratch[wave]);
            }
        }
    }
}

__global__ __launch_bounds__(256) void order_and_crops_kernel(Workspace ws) {
    extern __shared__ int order_lds[];              // [maxq] discovery positions, then [maxq][8] ordered corners
    int* s_start = order_lds;
    float (*s_sq)[8] = reinterpret_cast<float (*)[8]>(order_lds + ws.maxq);
    __shared__ int s_n;
    const int f = blockIdx.x;
    const int tid = threadIdx.x;
    int n = ws.n_quads_frame[f];
    if (n > ws.maxq) n = ws.maxq;
    const QuadRec* q = ws.quads_frame + (size_t)f * ws.maxq;
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
        float* out = ws.squares + ((size_t)f * ws.maxq + i) * 8;
        int quad[8];
        for (int k = 0; k < 8; k++) {
            out[k] = s_sq[i][k];
            quad[k] = (int)s_sq[i][k];
        }
        int x0, y0, cw, ch, roi_index = -1;
        crop_rect(quad, ws.W, ws.H, &x0, &y0, &cw, &ch);
        const int sw = cw & ~1, sh = ch & ~1, ns = (sw + 15) & ~15;
        if (sw >= 2 && sh >= 2) {
            const int r = atomicAdd(ws.counters + CNT_CROP_ROIS, 1);
            const long long plane = nbr_plane_bytes(ns, sh);
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
                    ws.crop_min_rest[r] = 0x7fffffff;
                    ws.ring_crop[r] = 0;   // (ring_quads_kernel sets it where the crop's own frame border needs no walk)
                    for (int t = 0; t < ntx * nty; t++) {
                        TileDesc td;
                        td.roi = r; td.x0 = t % ntx; td.y0 = (t / ntx) * MARCH_CROP_ROWS;
                        ws.tiles_crop[tbase + t] = td;
                    }
                    roi_index = r;
                }
            }
        }
        ws.crop_of[(size_t)f * ws.maxq + i] = roi_index;
    }
}

void launch_ring_quads_frames(const Workspace& ws, hipStream_t stream) {
    hipLaunchKernelGGL(ring_quads_kernel<false>, dim3(ws.n_frames < 4096 ? ws.n_frames : 4096), dim3(64), 0, stream, ws);
}
void launch_ring_quads_crops(const Workspace& ws, hipStream_t stream) {
    hipLaunchKernelGGL(ring_quads_kernel<true>, dim3(ws.n_frames >= 128 ? 4096 : ws.n_frames * 32), dim3(64), 0, stream, ws);
}
void launch_follow_frames(const Workspace& ws, hipStream_t stream) {
    hipLaunchKernelGGL(follow_kernel<false>, dim3(ws.short_blocks * (4 / FW)), dim3(FOLLOW_THREADS), 0, stream, ws);
}
void launch_follow_crops(const Workspace& ws, hipStream_t stream) {
    hipLaunchKernelGGL(follow_kernel<true>, dim3(ws.short_blocks * (4 / FW)), dim3(FOLLOW_THREADS), 0, stream, ws);
}
void launch_follow_mid_frames(const Workspace& ws, hipStream_t stream) {
    hipLaunchKernelGGL(follow_mid_kernel<false>, dim3(ws.mid_blocks * (4 / FW)), dim3(FOLLOW_THREADS), 0, stream, ws, 0);
}
void launch_follow_mid_crops(const Workspace& ws, hipStream_t stream) {
    if (ws.crop_phases == 1) {
        hipLaunchKernelGGL(follow_mid_kernel<true>, dim3(ws.mid_blocks * (4 / FW)), dim3(FOLLOW_THREADS), 0, stream, ws, 0);
        return;
    }
    hipLaunchKernelGGL(follow_mid_kernel<true>, dim3(ws.mid_blocks * (4 / FW)), dim3(FOLLOW_THREADS), 0, stream, ws, 1);
    hipLaunchKernelGGL(follow_mid_kernel<true>, dim3(ws.mid_blocks * (4 / FW)), dim3(FOLLOW_THREADS), 0, stream, ws, 2);
}
void launch_follow_long_frames(const Workspace& ws, hipStream_t stream) {
    hipLaunchKernelGGL(follow_long_kernel<false>, dim3(ws.long_blocks * (4 / FW)), dim3(FOLLOW_THREADS), 0, stream, ws);
}
void launch_follow_long_crops(const Workspace& ws, hipStream_t stream) {
    hipLaunchKernelGGL(follow_long_kernel<true>, dim3(ws.long_blocks * (4 / FW)), dim3(FOLLOW_THREADS), 0, stream, ws);
}
void launch_order_and_crops(const Workspace& ws, hipStream_t stream) {
    if (ws.n_frames > 0) hipLaunchKernelGGL(order_and_crops_kernel, dim3(ws.n_frames), dim3(ws.maxq <= 256 ? 64 : 256) /* one wave: starts wherever a SIMD has room */, (size_t)ws.maxq * 9 * sizeof(int), stream, ws);
}

}  // namespace ocvar
