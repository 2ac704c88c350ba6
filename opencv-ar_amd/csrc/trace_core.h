// trace_core.h -- border following, polygon approximation and the quad filter of the square finder.
//
// Replaces, for one border, what the reference obtains from OpenCV inside cvarFindSquares
// (/root/reference/src/opencvar.cpp:183-214): cvFindContours(RETR_LIST, CHAIN_APPROX_SIMPLE),
// cvContourPerimeter, cvApproxPoly(DP, 2 % perimeter), cvContourArea, cvCheckContourConvexity and
// the first-vertex border rule.
//
// MI355X formulation (not a translation of the sequential scan):
//  * the binarise kernel stores, per pixel, the 8-bit mask of non-zero neighbours, so one step of the
//    follower is one byte load + a rotate + a count-trailing-zeros;
//  * cvFindContours discovers each border at the raster-first scan position among the positions where
//    that border passes the west side (0->1 transition) or the east side (1->0 transition) of one of
//    its pixels.  The follower therefore starts from every locally plausible start in parallel and
//    drops out as soon as it meets a position of its own border that precedes its start; exactly one
//    start per border survives, and the surviving starts sorted by position reproduce OpenCV's
//    sequence order.  No labels are written into the image, so borders are independent work items.
#pragma once
#include "hd.h"
#include <math.h>

namespace ocvar {

OCVAR_HD float sqrt_rn(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __fsqrt_rn(v);
#else
    return sqrtf(v);
#endif
}

// first set bit of the 8-bit mask m (m != 0) at direction from, from+1, ... (counter-clockwise)
OCVAR_HD int first_ccw(unsigned m, int from) {
    from &= 7;
    unsigned r = ((m >> from) | (m << (8 - from))) & 0xFFu;
    return (from + __builtin_ctz(r)) & 7;
}
// first set bit at direction from, from-1, ... (clockwise)
OCVAR_HD int first_cw(unsigned m, int from) {
    from &= 7;
    unsigned r = ((m << (7 - from)) | (m >> (from + 1))) & 0xFFu;  // direction `from` -> bit 7
    return (from - (__builtin_clz(r) - 24)) & 7;
}

enum TraceStatus { TRACE_OK = 0, TRACE_NOT_FIRST = 1, TRACE_SINGLE = 2, TRACE_OVERRUN = 3 };
// (TRACE_OVERRUN doubles as "step budget exhausted" when the caller passes a small max_steps on purpose)

struct TraceStats {
    int status;
    int npts;
    int minx, maxx, miny, maxy;
    double perimeter;  // cvArcLength(closed): float32 segment lengths summed in double
    int steps;         // pixels visited (instrumentation)
};

// Follows the border that cvFindContours would start at scan position cpos (is_hole: 1->0 transition,
// the border's first pixel is cpos-1).  nbr: neighbour masks of an sw-wide plane.  With STORE the
// CHAIN_APPROX_SIMPLE points are written to out (x,y pairs).  Returns TRACE_NOT_FIRST as soon as a
// scan position of this border smaller than cpos is met.
// RUN: jump over straight runs with 8 mask loads in flight (pays off for a lone long border; in a wave whose
// lanes follow different borders the extra code path only adds divergence, so the lane tiers switch it off).
template <bool STORE, bool RUN = true>
OCVAR_HD TraceStats trace_border(const uint8_t* nbr, int sw, int plane, int cpos, int is_hole, int* out, int max_pts,
                                 int max_steps) {
    TraceStats st;
    st.status = TRACE_OK;
    st.npts = 0;
    st.minx = st.miny = 0x7fffffff;
    st.maxx = st.maxy = -0x7fffffff;
    st.perimeter = 0.0;
    st.steps = 0;
    const int i0 = cpos - is_hole;
    int x = i0 % sw, y = i0 / sw;
    unsigned m = nbr[nbr_addr(x, y, sw)];
    if (m == 0) {  // single-pixel domain (only reachable for outer borders)
        st.status = TRACE_SINGLE;
        st.npts = 1;
        st.minx = st.maxx = x;
        st.miny = st.maxy = y;
        if (STORE && max_pts > 0) {
            out[0] = x;
            out[1] = y;
        }
        return st;
    }
    int s = first_cw(m, (is_hole ? 0 : 4) - 1);
    const int i1 = i0 + dir_dy(s) * sw + dir_dx(s);
    int i3 = i0;
    int prev_s = s ^ 4;
    int fx = 0, fy = 0, lx = 0, ly = 0;
    int step = 0;
    for (;; step++) {
        if (step >= max_steps) {
            st.status = TRACE_OVERRUN;
            return st;
        }
        const int s_end = s;
        s = first_ccw(m, s_end + 1);
        const int examined = (s - (s_end + 1)) & 7;  // zero neighbours passed on the way
        // scan positions of this border at this pixel: west side passed -> i3, east side passed -> i3+1
        if ((((4 - (s_end + 1)) & 7) < examined && i3 < cpos) || (((0 - (s_end + 1)) & 7) < examined && i3 + 1 < cpos)) {
            st.status = TRACE_NOT_FIRST;
            return st;
        }
        if (s != prev_s) {
            if (st.npts == 0) {
                fx = x;
                fy = y;
            } else {
                float dx = (float)x - (float)lx, dy = (float)y - (float)ly;
                st.perimeter += (double)sqrt_rn(dx * dx + dy * dy);
            }
            lx = x;
            ly = y;
            st.minx = x < st.minx ? x : st.minx;
            st.maxx = x > st.maxx ? x : st.maxx;
            st.miny = y < st.miny ? y : st.miny;
            st.maxy = y > st.maxy ? y : st.maxy;
            if (STORE && st.npts < max_pts) {
                out[2 * st.npts] = x;
                out[2 * st.npts + 1] = y;
            }
            st.npts++;
            prev_s = s;
        }
        const int ddx = dir_dx(s), ddy = dir_dy(s), d = ddy * sw + ddx;
        int i4 = i3 + d;
        x += ddx;
        y += ddy;
        if (i4 == i0 && i3 == i1) break;
        if ((unsigned)i4 >= (unsigned)plane) {  // cannot happen on a consistent neighbour plane; never read outside it
            st.status = TRACE_OVERRUN;
            return st;
        }
        unsigned m4 = nbr[nbr_addr(x, y, sw)];
        if (m4 == 0) {
            st.status = TRACE_OVERRUN;
            return st;
        }
        if (RUN && s == ((s_end + 4) & 7) && m4 == m) {
            // Straight run.  The follower's state is (arrival direction, neighbour mask); the border went straight
            // through i3 and i4 has the same mask, so i4 -- and every further pixel with this mask -- leaves in
            // direction s again: no direction change (no point), same examined neighbours.  Fetch 8 pixels ahead
            // at once (independent loads: one memory latency per 8 steps instead of one per step).
            const bool w_ex = ((4 - (s_end + 1)) & 7) < examined, e_ex = ((0 - (s_end + 1)) & 7) < examined;
            int p = i4;  // last pixel known to carry mask m
            bool landed = false, closed = false;
            while (!landed && !closed) {
                unsigned long long ahead = 0;
                for (int t = 1; t <= 8; t++) {
                    const int qi = p + t * d;
                    const bool ok = qi >= 0 && qi < plane;   // (x,y) is the position of p here
                    ahead |= (unsigned long long)(ok ? nbr[nbr_addr(x + t * ddx, y + t * ddy, sw)] : 0) << (8 * (t - 1));
                }
                for (int t = 1; t <= 8; t++) {
                    // p carries mask m and leaves in direction s: its scan positions, then the step p -> q
                    if ((w_ex && p < cpos) || (e_ex && p + 1 < cpos)) {
                        st.status = TRACE_NOT_FIRST;
                        return st;
                    }
                    const int q = p + d;
                    x += ddx;
                    y += ddy;
                    if (q == i0 && p == i1) {
                        closed = true;
                        break;
                    }
                    const unsigned mq = (unsigned)(ahead >> (8 * (t - 1))) & 255u;
                    if ((unsigned)q >= (unsigned)plane || mq == 0 || ++step >= max_steps) {
                        st.status = TRACE_OVERRUN;
                        return st;
                    }
                    p = q;
                    if (mq != m) {
                        m4 = mq;
                        landed = true;
                        break;
                    }
                }
            }
            if (closed) break;
            i4 = p;
        }
        i3 = i4;
        m = m4;
        s = (s + 4) & 7;
    }
    if (st.npts > 1) {
        float dx = (float)fx - (float)lx, dy = (float)fy - (float)ly;
        st.perimeter += (double)sqrt_rn(dx * dx + dy * dy);
    }
    st.steps = step;
    return st;
}

// ---- lean follower -------------------------------------------------------------------------------------------
// Same stepping rules as trace_border, stripped to the dependent chain: no statistics while walking (they are
// recomputed from the stored points afterwards), direction deltas from packed 2-bit tables, the examined-neighbour
// test as one rotated bit mask.  Stores up to max_pts points (out must hold max_pts + 1: one scratch slot); npts keeps
// counting beyond that.
struct LeanTrace {
    int status;
    int npts;
    int steps;
};

// dy * ns for dy in {-1,0,1}: the 24-bit multiplier on the device (a full 32-bit multiply is quarter rate)
OCVAR_HD int mul_small(int d, int ns) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __mul24(d, ns);
#else
    return d * ns;
#endif
}
OCVAR_HD int step_dx(int s) { return (int)((0x901Au >> (2 * s)) & 3u) - 1; }   // 1,1,0,-1,-1,-1,0,1
OCVAR_HD int step_dy(int s) { return (int)((0xA901u >> (2 * s)) & 3u) - 1; }   // 0,-1,-1,-1,0,1,1,1

OCVAR_HD LeanTrace trace_lean(const uint8_t* nbr, int ns, int plane, int cpos, int is_hole, int* out, int max_pts, int max_steps) {
    LeanTrace r;
    r.status = TRACE_OK;
    r.npts = 0;
    r.steps = 0;
    const int i0 = cpos - is_hole;
    int x = i0 % ns, y = i0 / ns;
    unsigned m = nbr[nbr_addr(x, y, ns)];
    if (m == 0) {
        r.status = TRACE_SINGLE;
        r.npts = 1;
        return r;
    }
    int s = first_cw(m, (is_hole ? 0 : 4) - 1);       // direction of the border's last pixel seen from its first
    const int i1 = i0 + step_dy(s) * ns + step_dx(s);
    int idx = i0;
    int prev_s = s ^ 4;
    int step = 0;
    for (;; step++) {
        if (step >= max_steps) {
            r.status = TRACE_OVERRUN;
            break;
        }
        const int from = (s + 1) & 7;
        const int t = __builtin_ctz(((m * 0x101u) >> from) & 0xffu);   // zero neighbours passed before the next border pixel
        const int e = (from + t) & 7;                                   // exit direction
        const unsigned passed = ((((1u << t) - 1u) * 0x101u) << from) >> 8;   // 8-bit rotate of t ones to position `from`
        if (((passed & 0x10u) && idx < cpos) || ((passed & 1u) && idx + 1 < cpos)) {
            r.status = TRACE_NOT_FIRST;
            break;
        }
        const bool emit = e != prev_s;   // CHAIN_APPROX_SIMPLE: a point wherever the direction changes
        const int ex = x, ey = y;
        prev_s = e;
        const int dx = step_dx(e), dy = step_dy(e);
        const int nidx = idx + mul_small(dy, ns) + dx;
        x += dx;
        y += dy;
        if (nidx == i0 && idx == i1) {
            if (emit) {
                if (r.npts < max_pts) {
                    out[2 * r.npts] = ex;
                    out[2 * r.npts + 1] = ey;
                }
                r.npts++;
            }
            break;
        }
        if ((unsigned)nidx >= (unsigned)plane) {
            r.status = TRACE_OVERRUN;
            break;
        }
        idx = nidx;
        // The next step's mask is requested BEFORE this step's point is stored, and the store is unconditional
        // (non-points go to a scratch slot behind the last point): memory operations retire in order, so the wait for
        // the mask must be able to leave exactly these two younger stores in flight -- which the compiler can only
        // count when they are on the straight-line path.
        m = nbr[nbr_addr(x, y, ns)];
        if (max_pts > 0) {
            const int slot = (emit && r.npts < max_pts) ? r.npts : max_pts;
            out[2 * slot] = ex;
            out[2 * slot + 1] = ey;
        }
        r.npts += emit ? 1 : 0;
        if (m == 0) {
            r.status = TRACE_OVERRUN;
            break;
        }
        s = e ^ 4;
    }
    r.steps = step;
    return r;
}

// trace_lean with a straight-line loop body.  On the GPU every `if` inside a lane-divergent loop costs an EXEC-mask
// save/restore and a branch per wave and step, and a memory operation inside a branch makes the compiler wait for all
// outstanding ones; here the three ways a walk ends (not the border's first position, closed, left the plane) and the
// emission of a corner point are predicates, the load of the next mask and the store of the point are unconditional
// (a lane that ends re-reads its own pixel; a step without a point stores into the scratch slot behind the last
// point), and the only control flow is the loop itself.  Same results as trace_lean, step for step.
// out must hold max_pts + 1 points; with max_pts == 0 nothing is stored (out may be null).
// The walk as a resumable state (tier 2 on the GPU steps 64 independent walks per wave and hands a lane a new start as
// soon as its walk has ended).  status: -1 while running, then a TraceStatus.
struct FlatWalk {
    int x, y, idx, s, prev_s, npts, step, status;
    unsigned m;
    int i0, i1;
};

OCVAR_HD void flat_begin(FlatWalk& w, const uint8_t* nbr, int ns, int cpos, int is_hole) {
    w.i0 = cpos - is_hole;
    w.x = w.i0 % ns;
    w.y = w.i0 / ns;
    w.idx = w.i0;
    w.npts = 0;
    w.step = 0;
    w.status = -1;
    w.m = nbr[nbr_addr(w.x, w.y, ns)];
    w.s = 0;
    w.prev_s = 0;
    w.i1 = 0;
    if (w.m == 0) {   // single-pixel domain (only reachable for outer borders)
        w.status = TRACE_SINGLE;
        w.npts = 1;
        return;
    }
    w.s = first_cw(w.m, (is_hole ? 0 : 4) - 1);       // direction of the border's last pixel seen from its first
    w.i1 = w.i0 + mul_small(step_dy(w.s), ns) + step_dx(w.s);
    w.prev_s = w.s ^ 4;
}

// one step of a running walk (w.status < 0).  store(emit, x, y) is called once per step between the request for
// the next mask and the state update: emit = the pixel being left is a corner point, x/y its position (the caller decides
// where a point -- or the dummy of a step without one -- goes; see flat_step and follow.hip's tier 2).
// BUDGET = false: the step budget is not looked at (the caller checks w.step between blocks of steps instead).
template <bool BUDGET = true, class Store>
OCVAR_HD void flat_step_t(FlatWalk& w, const uint8_t* nbr, int ns, int plane, int cpos, int max_steps, Store&& store) {
    const int from = (w.s + 1) & 7;
    const int t = __builtin_ctz(((w.m * 0x101u) >> from) & 0xffu);   // zero neighbours passed before the next border pixel
    const int e = (from + t) & 7;                                     // exit direction
    const unsigned passed = ((((1u << t) - 1u) * 0x101u) << from) >> 8;   // 8-bit rotate of t ones to position `from`
    // (bitwise & and | on purpose: with && and || the compiler turns these predicates into nested divergent branches --
    // EXEC save/restore and a scalar branch per condition and step)
    const bool budget = BUDGET & (w.step >= max_steps);
    // the scan positions of this border at this pixel -- west side passed: idx, east side passed: idx + 1 -- against cpos, as
    // sign bits lined up with `passed` (bit 4 = W, bit 0 = E)
    const unsigned early = (((unsigned)(w.idx - cpos) >> 31) << 4) | ((unsigned)(w.idx + 1 - cpos) >> 31);
    const bool nf = !budget & ((passed & early) != 0u);
    const int dx = step_dx(e), dy = step_dy(e);
    const int nidx = w.idx + mul_small(dy, ns) + dx;
    const bool closes = !budget & !nf & (nidx == w.i0) & (w.idx == w.i1);
    const bool oob = (unsigned)nidx >= (unsigned)plane;   // cannot happen on a consistent plane; never read outside it
    // A walk that ends in this step is not stepped again (the callers loop while status < 0), so nothing below is guarded by
    // "the walk goes on": the position, the directions and the point count of an ended walk may be off by this one step --
    // except for a border that closes, whose last corner point is exactly this pixel (emit as for any other step).
    const bool emit = !budget & (e != w.prev_s);          // CHAIN_APPROX_SIMPLE: a point wherever the direction changes
    const int lx = w.x + dx, ly = w.y + dy;
    const unsigned m4 = nbr[oob ? 0u : nbr_addr(lx, ly, ns)];
    store(emit, w.x, w.y);
    w.npts += emit ? 1 : 0;
    w.status = budget ? (int)TRACE_OVERRUN : nf ? (int)TRACE_NOT_FIRST : closes ? (int)TRACE_OK : (oob | (m4 == 0u)) ? (int)TRACE_OVERRUN : -1;
    w.step += w.status < 0 ? 1 : 0;
    w.x = lx;
    w.y = ly;
    w.idx = nidx;
    w.prev_s = e;
    w.m = m4;
    w.s = e ^ 4;
}

// points as x,y int pairs in out (max_pts + 1 pairs: a step without a point stores into the scratch slot behind the last)
OCVAR_HD void flat_step(FlatWalk& w, const uint8_t* nbr, int ns, int plane, int cpos, int* out, int max_pts, int max_steps) {
    flat_step_t(w, nbr, ns, plane, cpos, max_steps, [&](bool emit, int x, int y) {
        if (max_pts > 0) {
            const int slot = (emit && w.npts < max_pts) ? w.npts : max_pts;
            out[2 * slot] = x;
            out[2 * slot + 1] = y;
        }
    });
}

OCVAR_HD LeanTrace trace_flat(const uint8_t* nbr, int ns, int plane, int cpos, int is_hole, int* out, int max_pts, int max_steps) {
    FlatWalk w;
    flat_begin(w, nbr, ns, cpos, is_hole);
    while (w.status < 0) flat_step(w, nbr, ns, plane, cpos, out, max_pts, max_steps);
    LeanTrace r;
    r.status = w.status;
    r.npts = w.npts;
    r.steps = w.step;
    return r;
}

// cvArcLength(closed) and the bounding box of stored contour points -> the TraceStats the filters expect
OCVAR_HD TraceStats stats_of_points(const int* pts, int n) {
    TraceStats st;
    st.status = TRACE_OK;
    st.npts = n;
    st.steps = 0;
    st.minx = st.miny = 0x7fffffff;
    st.maxx = st.maxy = -0x7fffffff;
    st.perimeter = 0.0;
    int px = pts[2 * (n - 1)], py = pts[2 * (n - 1) + 1];
    for (int i0 = 0; i0 < n; i0 += 8) {
        int xs[8], ys[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int q = i0 + k < n ? i0 + k : n - 1;
            xs[k] = pts[2 * q];
            ys[k] = pts[2 * q + 1];
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (i0 + k >= n) break;
            const int x = xs[k], y = ys[k];
            st.minx = x < st.minx ? x : st.minx;
            st.maxx = x > st.maxx ? x : st.maxx;
            st.miny = y < st.miny ? y : st.miny;
            st.maxy = y > st.maxy ? y : st.maxy;
            if (n > 1) {
                const float dx = (float)x - (float)px, dy = (float)y - (float)py;
                st.perimeter += (double)sqrt_rn(dx * dx + dy * dy);
            }
            px = x;
            py = y;
        }
    }
    return st;
}

// Walks the border backwards (against the follower) from the start for at most max_back pixels and
// reports whether a scan position of this border that precedes cpos lies there.  The follower runs
// outer borders down their left side first, so a plausible-but-late outer start on an upper-left
// staircase finds its predecessor one or two pixels behind it instead of a whole lap ahead.
// (pixel, exit direction) -> (previous pixel, its exit direction) is the inverse of the follower's step:
// the arrival direction is the first non-zero neighbour clockwise from exit-1.
OCVAR_HD bool earlier_start_behind(const uint8_t* nbr, int sw, int plane, int cpos, int is_hole, int max_back) {
    const int i0 = cpos - is_hole;
    int x = i0 % sw, y = i0 / sw;
    unsigned m = nbr[nbr_addr(x, y, sw)];
    if (m == 0) return false;
    const int b0 = first_cw(m, (is_hole ? 0 : 4) - 1);
    int q = i0 + dir_dy(b0) * sw + dir_dx(b0);
    x += dir_dx(b0);
    y += dir_dy(b0);
    int s = (b0 + 4) & 7;  // exit direction at q (towards the pixel we came from)
    for (int k = 0; k < max_back; k++) {
        if ((unsigned)q >= (unsigned)plane) return false;
        m = nbr[nbr_addr(x, y, sw)];
        if (m == 0) return false;
        const int b = first_cw(m, s - 1);
        if (q == i0 && b == b0) return false;  // back at the start visit: whole lap, nothing earlier
        const int examined = (s - (b + 1)) & 7;
        if ((((4 - (b + 1)) & 7) < examined && q < cpos) || (((0 - (b + 1)) & 7) < examined && q + 1 < cpos)) return true;
        q += dir_dy(b) * sw + dir_dx(b);
        x += dir_dx(b);
        y += dir_dy(b);
        s = (b + 4) & 7;
    }
    return false;
}

// Cheap exact rejection of a plausible start before any border step.  A start is real only if its pixel is the raster-first
// pixel of its region: of an 8-connected foreground component (outer start at pixel (x,y)) or of a 4-connected background
// region (hole start at the background pixel (x,y)).  The region contains the whole horizontal run that begins at (x,y);
// if any pixel of row y-1 touches that run -- 8-adjacent for a foreground run, directly above for a background run -- it
// belongs to the same region and precedes (x,y) in the scan, so no border can be discovered at (x,y).  The neighbour masks
// of the run's pixels hold exactly these bits (NW N NE of each run pixel / N of each run pixel; E says whether the run
// goes on), the loads are independent (one memory latency, usually one cache line), and on marker frames ~80 % of the
// plausible starts (stair corners of slanted edges) end here.  The binarise kernel's local test is the first pixel of
// this one.  max_run (<= 16) bounds the look-ahead; a run that is still going on after max_run pixels is given the benefit
// of the doubt.  (The device reads the 16 masks with two 16-byte loads: follow.hip::run_has_earlier_pixel_rows.)
OCVAR_HD bool run_has_earlier_pixel(const uint8_t* nbr, int ns, int cpos, int is_hole, int max_run) {
    const int x = cpos % ns, y = cpos / ns;
    unsigned m[16];
    const int n = max_run < 16 ? max_run : 16;
    for (int k = 0; k < 16; k++) m[k] = (k < n && x + k < ns) ? nbr[nbr_addr(x + k, y, ns)] : 0u;
    for (int k = 0; k < n && x + k < ns; k++) {
        if (is_hole) {
            if (!(m[k] & 0x04u)) return true;       // background directly above a pixel of the background run
            if (m[k] & 0x01u) return false;         // E is foreground: the run ends here
        } else {
            if (m[k] & 0x0eu) return true;          // NE, N or NW of a pixel of the foreground run is foreground
            if (!(m[k] & 0x01u)) return false;      // E is background: the run ends here
        }
    }
    return false;
}

struct DpSlice { int start, end; };

// cvApproxPoly(CV_POLY_APPROX_DP) on a closed integer contour of count >= 1 points (x,y pairs in src).
// dst must hold DP_MAX_OUT+1 points, stack DP_STACK slices.  Returns the vertex count after the
// collinear clean-up, or DP_MAX_OUT+1 as soon as the result is known to exceed 4 vertices (the
// clean-up removes at most every second vertex, so more than 8 raw vertices can never become 4).
// Every slice waiting on the stack ends in at least one vertex, so "vertices so far + slices waiting > 8" already
// decides that -- which also bounds the stack: it never holds more than DP_MAX_OUT + 1 slices.
constexpr int DP_MAX_OUT = 8;
constexpr int DP_STACK = DP_MAX_OUT + 4;

OCVAR_HD int approx_poly_dp(const int* src, int count, double parameter, int* dst, DpSlice* stack) {
    float eps = (float)parameter;  // cvApproxPoly passes the accuracy as float
    eps *= eps;
    int top = 0, new_count = 0;
    DpSlice slice = {0, 0}, right = {0, 0};
    int sx = 0, sy = 0;
    bool le_eps = false;
    int pos = 0;
    // two approximately farthest points
    for (int it = 0; it < 3; it++) {
        int max_dist = 0;
        pos = (pos + right.start) % count;
        sx = src[2 * pos];
        sy = src[2 * pos + 1];
        // points are fetched 8 at a time (independent loads first, then the dependent max-search): one memory
        // latency per 8 points instead of one per point when the contour lives in global memory
        for (int j0 = 1; j0 < count; j0 += 8) {
            int px[8], py[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                int q = pos + j0 + k;
                q = q >= count ? q - count : q;
                q = q >= count ? 0 : q;   // lanes past the end read a valid slot and are ignored below
                px[k] = src[2 * q];
                py[k] = src[2 * q + 1];
            }
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (j0 + k >= count) break;
                const int dx = px[k] - sx, dy = py[k] - sy;
                const int dist = dx * dx + dy * dy;
                if (dist > max_dist) {
                    max_dist = dist;
                    right.start = j0 + k;
                }
            }
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
        stack[top++] = right;
        stack[top++] = slice;
    }
    for (int guard = 4 * count + 16; top > 0;) {  // every iteration emits a vertex or splits a slice: <= 2*count rounds
        if (--guard < 0) return DP_MAX_OUT + 1;
        slice = stack[--top];
        int e = slice.end >= count ? slice.end - count : slice.end;
        int b = slice.start >= count ? slice.start - count : slice.start;
        const int ex = src[2 * e], ey = src[2 * e + 1];
        sx = src[2 * b];
        sy = src[2 * b + 1];
        if (slice.end > slice.start + 1) {
            const int dx = ex - sx, dy = ey - sy;
            int max_dist = 0;
            for (int i0 = slice.start + 1; i0 < slice.end; i0 += 8) {
                int px[8], py[8];
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    int q = i0 + k;                      // indices run up to < 2*count
                    q = q >= count ? q - count : q;
                    q = q >= count ? 0 : q;
                    px[k] = src[2 * q];
                    py[k] = src[2 * q + 1];
                }
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    if (i0 + k >= slice.end) break;
                    int d = (py[k] - sy) * dx - (px[k] - sx) * dy;
                    d = d < 0 ? -d : d;
                    if (d > max_dist) {
                        max_dist = d;
                        right.start = i0 + k;
                    }
                }
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
            if (new_count + top + 2 > DP_MAX_OUT) return DP_MAX_OUT + 1;
            stack[top++] = right;
            stack[top++] = slice;
        }
    }
    // clean-up of nearly collinear vertices on the closed ring
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
    return new_count;
}

// opencvar.cpp:199-206 on a 4-vertex result: |area| > 500, convex, first vertex inside the 2-px inset.
OCVAR_HD bool quad_filter(const int* q, int img_w, int img_h) {
    double a00 = 0;
    double xi_1 = q[6], yi_1 = q[7];
    for (int i = 0; i < 4; i++) {
        double xi = q[2 * i], yi = q[2 * i + 1];
        a00 += xi_1 * yi - xi * yi_1;
        xi_1 = xi;
        yi_1 = yi;
    }
    double area = a00 * 0.5;
    if (!((area < 0 ? -area : area) > 500)) return false;
    int orientation = 0;
    int cx = q[0], cy = q[1];
    int dx0 = cx - q[6], dy0 = cy - q[7];
    for (int i = 0; i < 4; i++) {
        int nx = q[2 * ((i + 1) & 3)], ny = q[2 * ((i + 1) & 3) + 1];
        int dx = nx - cx, dy = ny - cy;
        int dxdy0 = dx * dy0, dydx0 = dy * dx0;
        orientation |= (dydx0 > dxdy0) ? 1 : ((dydx0 < dxdy0) ? 2 : 3);
        if (orientation == 3) return false;
        dx0 = dx;
        dy0 = dy;
        cx = nx;
        cy = ny;
    }
    return q[0] > 2 && q[0] < img_w - 2 && q[1] > 2 && q[1] < img_h - 2;
}

// Cheap exact rejections before any point is stored: a 4-vertex result needs >= 4 contour points, and
// a quad inscribed in the contour's bounding box cannot have more area than the box.
OCVAR_HD bool worth_approximating(const TraceStats& st) {
    if (st.status != TRACE_OK || st.npts < 4) return false;
    long long bw = st.maxx - st.minx, bh = st.maxy - st.miny;
    return bw * bh > 500;
}

}  // namespace ocvar
