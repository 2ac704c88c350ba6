// emul.cpp -- host build of the device algorithm cores (opencv-ar_amd/csrc/*_core.h) for CPU-side
// logic tests.  TEST ONLY: nothing in the product links this; it exists so the formulation the HIP
// kernels use (parallel border starts, neighbour-mask follower, ...) can be checked against the oracle
// in the GPU-less container.
#include "trace_core.h"
#include <algorithm>
#include <cstring>
#include <vector>

using namespace ocvar;
static int g_back = 32;
static int g_run_filter = 1;   // follow.hip tier 1: run_has_earlier_pixel before any step
extern "C" void emul_set_run_filter(int v) { g_run_filter = v; }
static int g_run = 1;
static int g_lean = 0;
extern "C" void emul_set_lean(int v) { g_lean = v; }
extern "C" void emul_set_run(int r) { g_run = r; }
template <bool STORE> static TraceStats tb(const uint8_t* nbr, int sw, int plane, int cpos, int hole, int* out, int mp, int ms) {
    return g_run ? trace_border<STORE, true>(nbr, sw, plane, cpos, hole, out, mp, ms) : trace_border<STORE, false>(nbr, sw, plane, cpos, hole, out, mp, ms);
}
static long long g_back_drops = 0;
extern "C" void emul_set_back(int b) { g_back = b; }
extern "C" long long emul_back_drops() { return g_back_drops; }

extern "C" void emul_nbr_plane(const uint8_t* bin, int w, int h, uint8_t* nbr) {
    auto b = [&](int x, int y) -> int {
        if (x < 1 || y < 1 || x > w - 2 || y > h - 2) return 0;
        return bin[(size_t)y * w + x] != 0;
    };
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            unsigned m = 0;
            for (int s = 0; s < 8; s++) m |= (unsigned)b(x + dir_dx(s), y + dir_dy(s)) << s;
            nbr[(size_t)y * w + x] = (uint8_t)m;
        }
}

// candidate rule of the binarise kernel + follower; output in cvFindContours list order
extern "C" int emul_find_contours(const uint8_t* bin, int w, int h, int* pts, int max_pts, int* offs, int* starts,
                                  int* holes, int max_contours, long long* work_steps) {
    std::vector<uint8_t> nbr((size_t)w * h);
    emul_nbr_plane(bin, w, h, nbr.data());
    auto b = [&](int x, int y) -> int {
        if (x < 1 || y < 1 || x > w - 2 || y > h - 2) return 0;
        return bin[(size_t)y * w + x] != 0;
    };
    struct C { int pos, hole; std::vector<int> p; };
    std::vector<C> found;
    std::vector<int> buf(2 * (size_t)w * h * 4 + 16);
    for (int y = 1; y < h - 1; y++)
        for (int x = 1; x < w - 1; x++) {
            int c = b(x, y);
            bool outer = c && !b(x - 1, y) && !b(x - 1, y - 1) && !b(x, y - 1) && !b(x + 1, y - 1) && !(b(x + 1, y) && b(x + 2, y - 1));
            bool hole = !c && b(x - 1, y) && b(x, y - 1) && (b(x + 1, y) || b(x + 1, y - 1));
            if (!outer && !hole) continue;
            if (g_run_filter && run_has_earlier_pixel(nbr.data(), w, y * w + x, hole ? 1 : 0, 16)) continue;
            if (g_back > 0 && earlier_start_behind(nbr.data(), w, w * h, y * w + x, hole ? 1 : 0, g_back)) { g_back_drops++; continue; }
            TraceStats st;
            if (g_lean) {
                LeanTrace lt = g_lean == 2 ? trace_flat(nbr.data(), w, w * h, y * w + x, hole ? 1 : 0, buf.data(), (int)buf.size() / 2 - 1, 4 * w * h + 16)
                                           : trace_lean(nbr.data(), w, w * h, y * w + x, hole ? 1 : 0, buf.data(), (int)buf.size() / 2 - 1, 4 * w * h + 16);
                st.status = lt.status;
                st.npts = lt.npts;
                if (lt.status == TRACE_SINGLE) { buf[0] = x; buf[1] = y; }
            } else {
                st = tb<true>(nbr.data(), w, w * h, y * w + x, hole ? 1 : 0, buf.data(), (int)buf.size() / 2, 4 * w * h + 16);
            }
            if (st.status == TRACE_OVERRUN) return -2;
            if (st.status == TRACE_NOT_FIRST) continue;
            C cc;
            cc.pos = y * w + x;
            cc.hole = hole;
            cc.p.assign(buf.begin(), buf.begin() + 2 * st.npts);
            found.push_back(std::move(cc));
        }
    std::sort(found.begin(), found.end(), [](const C& a, const C& b) { return a.pos > b.pos; });
    if ((int)found.size() > max_contours) return -1;
    int np = 0;
    for (size_t i = 0; i < found.size(); i++) {
        offs[i] = np;
        starts[i] = found[i].pos;
        holes[i] = found[i].hole;
        int n = (int)found[i].p.size() / 2;
        if (np + n > max_pts) return -1;
        memcpy(pts + 2 * np, found[i].p.data(), sizeof(int) * 2 * n);
        np += n;
    }
    offs[found.size()] = np;
    return (int)found.size();
}

extern "C" int emul_approx_poly(const int* src, int n, double eps, int* dst) {
    std::vector<DpSlice> stack(n + 2);
    int tmp[2 * (DP_MAX_OUT + 1)];
    int m = approx_poly_dp(src, n, eps, tmp, stack.data());
    int k = m > DP_MAX_OUT ? DP_MAX_OUT : m;
    memcpy(dst, tmp, sizeof(int) * 2 * k);
    return m;
}

extern "C" int emul_quad_filter(const int* q, int w, int h) { return quad_filter(q, w, h) ? 1 : 0; }

// trace statistics (perimeter, bbox) for one start
extern "C" int emul_trace_stats(const uint8_t* nbr, int sw, int cpos, int is_hole, double* perimeter, int* bbox, int* npts) {
    TraceStats st = tb<false>(nbr, sw, 1 << 30, cpos, is_hole, nullptr, 0, 1 << 30);
    *perimeter = st.perimeter;
    bbox[0] = st.minx; bbox[1] = st.maxx; bbox[2] = st.miny; bbox[3] = st.maxy;
    *npts = st.npts;
    return st.status;
}

// The K3 flow on one binary plane: candidates -> pass 1 (stats, first-check) -> prune -> pass 2 -> DP -> filter.
// quads come out in the reference's push order (descending start).  stats: [0] candidates, [1] steps spent by
// candidates that dropped out, [2] steps of surviving borders (pass 1), [3] borders, [4] borders approximated.
extern "C" int emul_find_squares_bin(const uint8_t* bin, int sw, int sh, int img_w, int img_h, int* quads, int max_quads,
                                     long long* stats) {
    std::vector<uint8_t> nbr((size_t)sw * sh);
    emul_nbr_plane(bin, sw, sh, nbr.data());
    auto b = [&](int x, int y) -> int {
        if (x < 1 || y < 1 || x > sw - 2 || y > sh - 2) return 0;
        return bin[(size_t)y * sw + x] != 0;
    };
    struct Q { int start; int p[8]; };
    std::vector<Q> out;
    long long st_[5] = {0, 0, 0, 0, 0};
    std::vector<int> buf;
    for (int y = 1; y < sh - 1; y++)
        for (int x = 1; x < sw - 1; x++) {
            int c = b(x, y);
            bool outer = c && !b(x - 1, y) && !b(x - 1, y - 1) && !b(x, y - 1) && !b(x + 1, y - 1) && !(b(x + 1, y) && b(x + 2, y - 1));
            bool hole = !c && b(x - 1, y) && b(x, y - 1) && (b(x + 1, y) || b(x + 1, y - 1));
            if (!outer && !hole) continue;
            if (g_run_filter && run_has_earlier_pixel(nbr.data(), sw, y * sw + x, hole ? 1 : 0, 16)) continue;
            if (g_back > 0 && earlier_start_behind(nbr.data(), sw, sw * sh, y * sw + x, hole ? 1 : 0, g_back)) { g_back_drops++; continue; }
            st_[0]++;
            if (g_lean) {   // the tier-2/3 flow: store while following, statistics from the stored points
                buf.resize(2 * 4097);
                LeanTrace lt = trace_lean(nbr.data(), sw, sw * sh, y * sw + x, hole, buf.data(), 4096, 4 * sw * sh + 16);
                if (lt.status == TRACE_OVERRUN || lt.npts > 4096) return -3;
                if (lt.status != TRACE_OK) continue;
                st_[3]++;
                TraceStats st = stats_of_points(buf.data(), lt.npts);
                if (!worth_approximating(st)) continue;
                st_[4]++;
                std::vector<DpSlice> stack(st.npts + 2);
                int dst[2 * (DP_MAX_OUT + 1)];
                int m = approx_poly_dp(buf.data(), st.npts, st.perimeter * 0.02, dst, stack.data());
                if (m == 4 && quad_filter(dst, img_w, img_h)) {
                    Q q;
                    q.start = y * sw + x;
                    memcpy(q.p, dst, sizeof q.p);
                    out.push_back(q);
                }
                continue;
            }
            // count steps by re-running with a step cap search (cheap instrumentation)
            TraceStats st = tb<false>(nbr.data(), sw, sw * sh, y * sw + x, hole, nullptr, 0, 4 * sw * sh + 16);
            if (st.status == TRACE_NOT_FIRST) {
                int lo = 0, hi = 4 * sw * sh + 16;  // smallest cap that still reports NOT_FIRST = steps taken
                while (lo < hi) {
                    int mid = (lo + hi) / 2;
                    TraceStats t2 = tb<false>(nbr.data(), sw, sw * sh, y * sw + x, hole, nullptr, 0, mid);
                    if (t2.status == TRACE_NOT_FIRST) hi = mid; else lo = mid + 1;
                }
                st_[1] += lo + 1;
                continue;
            }
            st_[3]++;
            if (!worth_approximating(st)) continue;
            st_[4]++;
            buf.resize(2 * (size_t)st.npts);
            tb<true>(nbr.data(), sw, sw * sh, y * sw + x, hole, buf.data(), st.npts, 4 * sw * sh + 16);
            std::vector<DpSlice> stack(st.npts + 2);
            int dst[2 * (DP_MAX_OUT + 1)];
            int m = approx_poly_dp(buf.data(), st.npts, st.perimeter * 0.02, dst, stack.data());
            if (m == 4 && quad_filter(dst, img_w, img_h)) {
                Q q;
                q.start = y * sw + x;
                memcpy(q.p, dst, sizeof q.p);
                out.push_back(q);
            }
        }
    std::sort(out.begin(), out.end(), [](const Q& a, const Q& b) { return a.start > b.start; });
    if ((int)out.size() > max_quads) return -1;
    for (size_t i = 0; i < out.size(); i++) memcpy(quads + 8 * i, out[i].p, sizeof out[i].p);
    if (stats) memcpy(stats, st_, sizeof st_);
    return (int)out.size();
}

// instrumentation: per-candidate (type, steps until drop or -1 if survivor)
extern "C" int emul_candidate_steps(const uint8_t* bin, int sw, int sh, int* type, int* steps, int maxn) {
    std::vector<uint8_t> nbr((size_t)sw * sh);
    emul_nbr_plane(bin, sw, sh, nbr.data());
    auto b = [&](int x, int y) -> int {
        if (x < 1 || y < 1 || x > sw - 2 || y > sh - 2) return 0;
        return bin[(size_t)y * sw + x] != 0;
    };
    int n = 0;
    for (int y = 1; y < sh - 1; y++)
        for (int x = 1; x < sw - 1; x++) {
            int c = b(x, y);
            bool outer = c && !b(x - 1, y) && !b(x - 1, y - 1) && !b(x, y - 1) && !b(x + 1, y - 1) && !(b(x + 1, y) && b(x + 2, y - 1));
            bool hole = !c && b(x - 1, y) && b(x, y - 1) && (b(x + 1, y) || b(x + 1, y - 1));
            if (!outer && !hole) continue;
            if (g_back > 0 && earlier_start_behind(nbr.data(), sw, sw * sh, y * sw + x, hole ? 1 : 0, g_back)) { g_back_drops++; continue; }
            TraceStats st = tb<false>(nbr.data(), sw, sw * sh, y * sw + x, hole, nullptr, 0, 4 * sw * sh + 16);
            int s = -1;
            if (st.status == TRACE_NOT_FIRST) {
                int lo = 0, hi = 4 * sw * sh + 16;
                while (lo < hi) {
                    int mid = (lo + hi) / 2;
                    TraceStats t2 = tb<false>(nbr.data(), sw, sw * sh, y * sw + x, hole, nullptr, 0, mid);
                    if (t2.status == TRACE_NOT_FIRST) hi = mid; else lo = mid + 1;
                }
                s = lo + 1;
            }
            if (n < maxn) { type[n] = hole; steps[n] = s; }
            n++;
        }
    return n;
}

#include "decode_core.h"
#include "pose_core.h"
#include "tail_core.h"

extern "C" long long emul_read_code(const uint8_t* crop, int cw, int ch, int stride, const float* patPoint, int tw, int th) {
    return read_code(crop, cw, ch, stride, patPoint, tw, th);
}
extern "C" void emul_square_to_glmatrix(const float* sq, const void* cam, double ratio, double* gl) {
    square_to_glmatrix(sq, *(const CameraRec*)cam, ratio, gl);
}
extern "C" void emul_dedupe(int* markerId, const int* templateId, const double* score, int n) { dedupe(markerId, templateId, score, n); }
extern "C" int emul_sizes(int* out) {
    out[0] = sizeof(TemplateRec); out[1] = sizeof(CameraRec); out[2] = sizeof(MarkerRec); out[3] = sizeof(Roi);
    return 4;
}

// ---- host emulation of follow.hip::trace_lean_tiled (tier 3): the wave's 64 lanes are a loop, LDS is an array ----
namespace tiled_emul {
constexpr int TILE = 64;
struct TileCache {
    uint8_t lds[TILE * TILE];
    const uint8_t* nbr;
    int ns, sh, tx0, ty0;
    long long loads;
};
static void tile_load(TileCache& t, int x, int y, int bx = 0, int by = 0) {
    t.tx0 = (x + 20 * bx - TILE / 2 + 8) & ~15;
    t.ty0 = y + 28 * by - TILE / 2;
    t.loads++;
    for (int lane = 0; lane < 64; lane++) {
        const int row = t.ty0 + lane;
        for (int cch = 0; cch < TILE / 16; cch++) {
            const int cx = t.tx0 + 16 * cch;
            for (int b = 0; b < 16; b++) {
                uint8_t v = 0;
                if (row >= 0 && row < t.sh && cx >= 0 && cx + 16 <= t.ns) v = t.nbr[(long long)row * t.ns + cx + b];
                t.lds[lane * TILE + 16 * cch + b] = v;
            }
        }
    }
}
static unsigned tile_get(TileCache& t, int x, int y) {
    if ((unsigned)(x - t.tx0) >= (unsigned)TILE || (unsigned)(y - t.ty0) >= (unsigned)TILE) tile_load(t, x, y);
    return t.lds[(y - t.ty0) * TILE + (x - t.tx0)];
}
static LeanTrace trace(TileCache& t, int cpos, int is_hole, int* out, int max_pts, int max_steps) {
    LeanTrace r;
    r.status = TRACE_OK; r.npts = 0; r.steps = 0;
    const int ns = t.ns;
    const int i0 = cpos - is_hole;
    const int x0 = i0 % ns, y0 = i0 / ns;
    int x = x0, y = y0;
    unsigned m = tile_get(t, x, y);
    if (m == 0) { r.status = TRACE_SINGLE; r.npts = 1; return r; }
    int s = first_cw(m, (is_hole ? 0 : 4) - 1);
    const int x1 = x0 + step_dx(s), y1 = y0 + step_dy(s);
    int prev_s = s ^ 4;
    int straight = 0;
    int step = 0;
    for (;; step++) {
        if (step >= max_steps) { r.status = TRACE_OVERRUN; break; }
        const int from = (s + 1) & 7;
        const int tz = __builtin_ctz(((m * 0x101u) >> from) & 0xffu);
        const int e = (from + tz) & 7;
        const unsigned passed = ((((1u << tz) - 1u) * 0x101u) << from) >> 8;
        const int idx = y * ns + x;
        if (((passed & 0x10u) && idx < cpos) || ((passed & 1u) && idx + 1 < cpos)) { r.status = TRACE_NOT_FIRST; break; }
        if (e != prev_s) {
            if (r.npts < max_pts) { out[2 * r.npts] = x; out[2 * r.npts + 1] = y; }
            r.npts++;
            prev_s = e;
        }
        const int ddx = step_dx(e), ddy = step_dy(e);
        const int px = x, py = y;
        x += ddx; y += ddy;
        if (x == x0 && y == y0 && px == x1 && py == y1) break;
        if ((unsigned)x >= (unsigned)ns || (unsigned)y >= (unsigned)t.sh) { r.status = TRACE_OVERRUN; break; }
        unsigned m4 = tile_get(t, x, y);
        if (m4 == 0) { r.status = TRACE_OVERRUN; break; }
        if (e == (s ^ 4) && m4 == m) {
            if (++straight >= 2) {
                bool closed = false, bad = false;
                for (long long spin = 0;; spin++) {
                    if (spin > 100000000ll) { r.status = 77; bad = true; break; }   // emulation only: report a spin
                    int k = 64, stop_closes = 0; unsigned stop_m = 0;
                    for (int lane = 0; lane < 64; lane++) {
                        const int qx = x + (lane + 1) * ddx, qy = y + (lane + 1) * ddy;
                        const bool in_tile = (unsigned)(qx - t.tx0) < (unsigned)TILE && (unsigned)(qy - t.ty0) < (unsigned)TILE;
                        const unsigned mq = in_tile ? t.lds[(qy - t.ty0) * TILE + (qx - t.tx0)] : 256u;
                        const bool closes = qx == x0 && qy == y0 && qx - ddx == x1 && qy - ddy == y1;
                        if (closes || mq != m) { k = lane; stop_closes = closes; stop_m = mq; break; }
                    }
                    const int pa = y * ns + x, pb = (y + k * ddy) * ns + x + k * ddx;
                    if (((passed & 0x10u) && (pa < cpos || pb < cpos)) || ((passed & 1u) && (pa + 1 < cpos || pb + 1 < cpos))) {
                        r.status = TRACE_NOT_FIRST; bad = true; break;
                    }
                    x += k * ddx; y += k * ddy; step += k;
                    if (step >= max_steps) { r.status = TRACE_OVERRUN; bad = true; break; }
                    if (k == 64) continue;
                    if (stop_closes) { closed = true; break; }
                    if (stop_m == 256u) {
                        if ((unsigned)(x + ddx) >= (unsigned)ns || (unsigned)(y + ddy) >= (unsigned)t.sh) { r.status = TRACE_OVERRUN; bad = true; break; }
                        tile_load(t, x + ddx, y + ddy, ddx, ddy);
                        continue;
                    }
                    x += ddx; y += ddy; step++;
                    m4 = stop_m;
                    break;
                }
                if (bad || closed) break;
                if (m4 == 0) { r.status = TRACE_OVERRUN; break; }
                straight = 0;
            }
        } else {
            straight = 0;
        }
        m = m4;
        s = e ^ 4;
    }
    r.steps = step;
    return r;
}
}  // namespace tiled_emul

// follows the border starting at cpos with the tier-3 emulation; returns status, fills npts/steps/loads
extern "C" int emul_trace_tiled(const uint8_t* nbr, int ns, int sh, int cpos, int is_hole, int* out, int max_pts, int max_steps, int* info) {
    tiled_emul::TileCache t;
    t.nbr = nbr; t.ns = ns; t.sh = sh; t.tx0 = t.ty0 = -(1 << 28); t.loads = 0;
    LeanTrace r = tiled_emul::trace(t, cpos, is_hole, out, max_pts, max_steps);
    info[0] = r.npts; info[1] = r.steps; info[2] = (int)t.loads;
    return r.status;
}

// ---- layout of the panelled grey plane (hd.h) ---------------------------------------------------------------------------
extern "C" int emul_gray_pitch(int W) { return ocvar::gray_pitch(W); }
extern "C" unsigned emul_gray_col(int x) { return ocvar::gray_col(x); }
