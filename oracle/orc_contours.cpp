// orc_contours.cpp -- oracle: contour stages of cvarFindSquares (TEST INFRASTRUCTURE, see oracle.h).
//
// Restates the OpenCV 2.4.x algorithms behind these reference call sites:
//   opencvar.cpp:183-184  cvFindContours(RETR_LIST, CHAIN_APPROX_SIMPLE)   
//   opencvar.cpp:192      cvContourPerimeter                               
//   opencvar.cpp:190-192  cvApproxPoly(CV_POLY_APPROX_DP)                  
//   opencvar.cpp:199-202  cvContourArea, cvCheckContourConvexity           
//   opencvar.cpp:156-223  cvarFindSquares (glue)
// The scan below is the literal sequential Suzuki-Abe raster scan with in-image labels (2 / -126);
// it is deliberately NOT the geometric formulation the HIP path uses, so the two check each other.
// Parity unpinned at the OpenCV boundary.
#include "oracle.h"
#include <vector>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace {

const int DX[8] = {1, 1, 0, -1, -1, -1, 0, 1};
const int DY[8] = {0, -1, -1, -1, 0, 1, 1, 1};

struct Contour {
    std::vector<int> pts;  // x,y
    int start;             // scan position y*w + x
    int hole;
};

// icvFetchContour, CHAIN_APPROX_SIMPLE branch.  img: signed labels, step = w.
void fetch_contour(signed char* img, int step, int i0, int ptx, int pty, bool is_hole, std::vector<int>& out) {
    int deltas[16];
    for (int s = 0; s < 8; s++) deltas[s] = deltas[s + 8] = DY[s] * step + DX[s];
    const signed char nbd = 2;
    int s_end, s;
    s_end = s = is_hole ? 0 : 4;
    int i1;
    do {
        s = (s - 1) & 7;
        i1 = i0 + deltas[s];
        if (img[i1] != 0) break;
    } while (s != s_end);

    if (s == s_end) {  // single pixel domain
        img[i0] = (signed char)(nbd | -128);
        out.push_back(ptx);
        out.push_back(pty);
        return;
    }
    int i3 = i0, i4;
    int prev_s = s ^ 4;
    for (;;) {
        s_end = s;
        for (;;) {
            i4 = i3 + deltas[++s];
            if (img[i4] != 0) break;
        }
        s &= 7;
        if ((unsigned)(s - 1) < (unsigned)s_end)
            img[i3] = (signed char)(nbd | -128);
        else if (img[i3] == 1)
            img[i3] = nbd;
        if (s != prev_s) {
            out.push_back(ptx);
            out.push_back(pty);
            prev_s = s;
        }
        ptx += DX[s];
        pty += DY[s];
        if (i4 == i0 && i3 == i1) break;
        i3 = i4;
        s = (s + 4) & 7;
    }
}

// cvFindContours with RETR_LIST: returns contours in list order (last discovered first).
void find_contours(const uint8_t* bin, int w, int h, std::vector<Contour>& list) {
    std::vector<signed char> img((size_t)w * h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            bool border = (x == 0 || y == 0 || x == w - 1 || y == h - 1);
            img[(size_t)y * w + x] = (!border && bin[(size_t)y * w + x]) ? 1 : 0;
        }
    std::vector<Contour> found;
    for (int y = 1; y < h - 1; y++) {
        signed char* row = &img[(size_t)y * w];
        int prev = 0;
        for (int x = 1; x < w - 1; x++) {
            int p = row[x];
            if (p == prev) continue;
            bool is_hole = false;
            bool start = true;
            if (!(prev == 0 && p == 1)) {
                if (p != 0 || prev < 1)
                    start = false;
                else
                    is_hole = true;
            }
            if (start) {
                Contour c;
                c.start = y * w + x;
                c.hole = is_hole;
                int ox = x - (is_hole ? 1 : 0);
                fetch_contour(&img[0], w, y * w + ox, ox, y, is_hole, c.pts);
                found.push_back(std::move(c));
                p = row[x];  // scanner resumes at x+1 with prev = img[x]
            }
            prev = p;
        }
    }
    list.clear();
    for (size_t i = found.size(); i-- > 0;) list.push_back(std::move(found[i]));
}

}  // namespace

extern "C" int orc_find_contours(const uint8_t* bin, int w, int h, int* pts, int max_pts, int* offs, int* starts,
                                 int* holes, int max_contours) {
    std::vector<Contour> list;
    find_contours(bin, w, h, list);
    if ((int)list.size() > max_contours) return -1;
    int np = 0;
    for (size_t i = 0; i < list.size(); i++) {
        offs[i] = np;
        starts[i] = list[i].start;
        holes[i] = list[i].hole;
        int n = (int)list[i].pts.size() / 2;
        if (np + n > max_pts) return -1;
        memcpy(pts + 2 * np, list[i].pts.data(), sizeof(int) * 2 * n);
        np += n;
    }
    offs[list.size()] = np;
    return (int)list.size();
}

extern "C" double orc_arc_length_closed(const int* pts, int n) {
    // cvArcLength(closed): float32 segment lengths, double accumulation.
    double perimeter = 0;
    if (n <= 1) return 0;
    int px = pts[2 * (n - 1)], py = pts[2 * (n - 1) + 1];
    for (int i = 0; i < n; i++) {
        float dx = (float)pts[2 * i] - (float)px, dy = (float)pts[2 * i + 1] - (float)py;
        float d2 = dx * dx + dy * dy;
        perimeter += (double)sqrtf(d2);
        px = pts[2 * i];
        py = pts[2 * i + 1];
    }
    return perimeter;
}

// ---- which cvApproxPoly?  (parity is unpinned here: no OpenCV in the image; the variants size the uncertainty) ------------------
// The reference says "OpenCV 2.3 or newer" (README.md:18-19) and calls cvApproxPoly(CV_POLY_APPROX_DP) (opencvar.cpp:190-192).
//   variant 0 (default; what the HIP path reproduces): the legacy C routine icvApproxPolyDP_32s of OpenCV <= 2.4.3 -- the
//             accuracy arrives as FLOAT, integer distances, `(float)max_dist <= eps` in the seeding loop.
//   variant 1: approxPolyDP_<int> of OpenCV 2.4.4+ (cvApproxPoly became a wrapper of the templated C++ routine): the accuracy
//             stays DOUBLE, distances are computed in double, no float cast anywhere.
//   variant 2: the same with the extra clean-up condition `successive_inner_product >= 0` of later 2.4.x / 3.x releases.
// All from recollection of those sources.  tools/approx_variants.py counts over >= 10 k synthetic frames how often a variant
// changes a contour's vertices, a frame's quads, or a final marker (DESIGN.md section 5 has the numbers).
static int g_approx_variant = 0;
extern "C" void orc_set_approx_variant(int v) { g_approx_variant = v < 0 || v > 2 ? 0 : v; }
extern "C" int orc_get_approx_variant(void) { return g_approx_variant; }

static int approx_poly_double(const int* src, int count, double eps, int* dst, bool inner_product) {
    struct Slice { int start, end; };
    if (count == 0) return 0;
    eps *= eps;
    std::vector<Slice> stack;
    Slice slice = {0, 0}, right_slice = {0, 0};
    int new_count = 0;
    int sx = 0, sy = 0, ex = 0, ey = 0, px = 0, py = 0;
    bool le_eps = false;
    int pos = 0;
    // READ_PT(pt, pos): pt = src[pos]; if (++pos >= count) pos = 0
#define RD(X, Y) do { X = src[2 * pos]; Y = src[2 * pos + 1]; if (++pos >= count) pos = 0; } while (0)
    right_slice.start = 0;
    for (int i = 0; i < 3; i++) {
        double max_dist = 0;
        pos = (pos + right_slice.start) % count;
        RD(sx, sy);
        for (int j = 1; j < count; j++) {
            RD(px, py);
            const double dx = px - sx, dy = py - sy;
            const double dist = dx * dx + dy * dy;
            if (dist > max_dist) {
                max_dist = dist;
                right_slice.start = j;
            }
        }
        le_eps = max_dist <= eps;
    }
    if (!le_eps) {
        right_slice.end = slice.start = pos % count;
        slice.end = right_slice.start = (right_slice.start + slice.start) % count;
        stack.push_back(right_slice);
        stack.push_back(slice);
    } else {
        dst[0] = sx;
        dst[1] = sy;
        new_count = 1;
    }
    while (!stack.empty()) {
        slice = stack.back();
        stack.pop_back();
        ex = src[2 * slice.end];
        ey = src[2 * slice.end + 1];
        pos = slice.start;
        RD(sx, sy);
        if (pos != slice.end) {
            const double dx = ex - sx, dy = ey - sy;
            double max_dist = 0;
            while (pos != slice.end) {
                RD(px, py);
                const double dist = std::fabs((py - sy) * dx - (px - sx) * dy);
                if (dist > max_dist) {
                    max_dist = dist;
                    right_slice.start = (pos + count - 1) % count;
                }
            }
            le_eps = max_dist * max_dist <= eps * (dx * dx + dy * dy);
        } else {
            le_eps = true;
            sx = src[2 * slice.start];
            sy = src[2 * slice.start + 1];
        }
        if (le_eps) {
            dst[2 * new_count] = sx;
            dst[2 * new_count + 1] = sy;
            new_count++;
        } else {
            right_slice.end = slice.end;
            slice.end = right_slice.start;
            stack.push_back(right_slice);
            stack.push_back(slice);
        }
    }
#undef RD
    count = new_count;
    if (count == 0) return 0;
    int r = count - 1;
#define RDD(X, Y) do { X = dst[2 * r]; Y = dst[2 * r + 1]; if (++r >= count) r = 0; } while (0)
    RDD(sx, sy);
    int wpos = r;
    RDD(px, py);
    for (int i = 0; i < count && new_count > 2; i++) {
        RDD(ex, ey);
        const double dx = ex - sx, dy = ey - sy;
        const double dist = std::fabs((px - sx) * dy - (py - sy) * dx);
        const double sip = (double)(px - sx) * (ex - px) + (double)(py - sy) * (ey - py);
        if (dist * dist <= 0.5 * eps * (dx * dx + dy * dy) && dx != 0 && dy != 0 && (!inner_product || sip >= 0)) {
            new_count--;
            dst[2 * wpos] = sx = ex;
            dst[2 * wpos + 1] = sy = ey;
            if (++wpos >= count) wpos = 0;
            RDD(px, py);
            i++;
            continue;
        }
        dst[2 * wpos] = sx = px;
        dst[2 * wpos + 1] = sy = py;
        if (++wpos >= count) wpos = 0;
        px = ex;
        py = ey;
    }
#undef RDD
    return new_count;
}

extern "C" int orc_approx_poly(const int* src, int count, double parameter, int* dst) {
    if (g_approx_variant) return approx_poly_double(src, count, parameter, dst, g_approx_variant == 2);
    // icvApproxPolyDP_32s, closed contour; eps is passed as float by cvApproxPoly.
    struct Slice { int start, end; };
    float eps = (float)parameter;
    if (count == 0) return 0;
    eps *= eps;
    std::vector<Slice> stack;
    Slice slice = {0, 0}, right_slice = {0, 0};
    int new_count = 0;
    int sx = 0, sy = 0, ex = 0, ey = 0, px = 0, py = 0;
    bool le_eps = false;
    int pos = 0;
#define RD(X, Y, P) do { X = src[2 * ((P) % count)]; Y = src[2 * ((P) % count) + 1]; } while (0)

    // 1. approximately two farthest points
    right_slice.start = 0;
    for (int i = 0; i < 3; i++) {
        int max_dist = 0;
        pos = (pos + right_slice.start) % count;
        RD(sx, sy, pos);
        for (int j = 1; j < count; j++) {
            RD(px, py, pos + j);
            int dx = px - sx, dy = py - sy;
            int dist = dx * dx + dy * dy;
            if (dist > max_dist) {
                max_dist = dist;
                right_slice.start = j;
            }
        }
        le_eps = (float)max_dist <= eps;
    }
    // 2. initialise the stack
    if (!le_eps) {
        slice.start = pos;
        slice.end = right_slice.start += slice.start;
        right_slice.start -= right_slice.start >= count ? count : 0;
        right_slice.end = slice.start;
        if (right_slice.end < right_slice.start) right_slice.end += count;
        stack.push_back(right_slice);
        stack.push_back(slice);
    } else {
        dst[0] = sx;
        dst[1] = sy;
        new_count = 1;
    }
    // 3. recursive process
    while (!stack.empty()) {
        slice = stack.back();
        stack.pop_back();
        RD(ex, ey, slice.end);
        RD(sx, sy, slice.start);
        if (slice.end > slice.start + 1) {
            int dx = ex - sx, dy = ey - sy;
            int max_dist = 0;
            for (int i = slice.start + 1; i < slice.end; i++) {
                RD(px, py, i);
                int dist = abs((py - sy) * dx - (px - sx) * dy);
                if (dist > max_dist) {
                    max_dist = dist;
                    right_slice.start = i;
                }
            }
            le_eps = (double)max_dist * max_dist <= (double)eps * ((double)dx * dx + (double)dy * dy);
        } else {
            le_eps = true;
        }
        if (le_eps) {
            dst[2 * new_count] = sx;
            dst[2 * new_count + 1] = sy;
            new_count++;
        } else {
            right_slice.end = slice.end;
            slice.end = right_slice.start;
            stack.push_back(right_slice);
            stack.push_back(slice);
        }
    }
#undef RD
    // 4. clean-up of nearly collinear vertices on the closed output ring
    count = new_count;
    if (count == 0) return 0;
    int r = count - 1;
#define RDD(X, Y) do { X = dst[2 * r]; Y = dst[2 * r + 1]; if (++r >= count) r = 0; } while (0)
    RDD(sx, sy);
    int wpos = r;
    RDD(px, py);
    for (int i = 0; i < count && new_count > 2; i++) {
        RDD(ex, ey);
        int dx = ex - sx, dy = ey - sy;
        int dist = abs((px - sx) * dy - (py - sy) * dx);
        if ((double)dist * dist <= 0.5 * eps * (dx * dx + dy * dy) && dx != 0 && dy != 0) {
            new_count--;
            dst[2 * wpos] = sx = ex;
            dst[2 * wpos + 1] = sy = ey;
            if (++wpos >= count) wpos = 0;
            RDD(px, py);
            i++;
            continue;
        }
        dst[2 * wpos] = sx = px;
        dst[2 * wpos + 1] = sy = py;
        if (++wpos >= count) wpos = 0;
        px = ex;
        py = ey;
    }
#undef RDD
    return new_count;
}

extern "C" double orc_contour_area(const int* pts, int n) {
    if (n == 0) return 0;
    double a00 = 0;
    double xi_1 = pts[2 * (n - 1)], yi_1 = pts[2 * (n - 1) + 1];
    for (int i = 0; i < n; i++) {
        double xi = pts[2 * i], yi = pts[2 * i + 1];
        a00 += xi_1 * yi - xi * yi_1;
        xi_1 = xi;
        yi_1 = yi;
    }
    return fabs(a00 * 0.5);
}

extern "C" int orc_is_convex(const int* pts, int n) {
    if (n == 0) return 0;
    int orientation = 0;
    int px = pts[2 * (n - 1)], py = pts[2 * (n - 1) + 1];
    int cx = pts[0], cy = pts[1];
    int dx0 = cx - px, dy0 = cy - py;
    for (int i = 0; i < n; i++) {
        int nx = pts[2 * ((i + 1) % n)], ny = pts[2 * ((i + 1) % n) + 1];
        int dx = nx - cx, dy = ny - cy;
        int dxdy0 = dx * dy0, dydx0 = dy * dx0;
        orientation |= (dydx0 > dxdy0) ? 1 : ((dydx0 < dxdy0) ? 2 : 3);
        if (orientation == 3) return 0;
        dx0 = dx;
        dy0 = dy;
        cx = nx;
        cy = ny;
    }
    return 1;
}

extern "C" int orc_find_squares(const uint8_t* gray, int w, int h, int stride, int* quads, int max_quads) {
    int sw = w & -2, sh = h & -2;
    if (sw < 2 || sh < 2) return 0;
    std::vector<uint8_t> bin((size_t)sw * sh);
    orc_binarise(gray, w, h, stride, bin.data(), nullptr);
    std::vector<Contour> list;
    find_contours(bin.data(), sw, sh, list);
    int nq = 0;
    std::vector<int> out;
    for (size_t ci = 0; ci < list.size(); ci++) {
        const Contour& c = list[ci];
        int n = (int)c.pts.size() / 2;
        out.resize(2 * (size_t)n + 2);
        double perim = orc_arc_length_closed(c.pts.data(), n);
        int m = orc_approx_poly(c.pts.data(), n, perim * 0.02, out.data());
        bool check = orc_contour_area(out.data(), m) > 500;
        if (m == 4 && check && orc_is_convex(out.data(), 4)) {
            int px = out[0], py = out[1];
            if (px > 2 && px < w - 2 && py > 2 && py < h - 2) {  // img->width/height, not sz (D7)
                if (nq >= max_quads) return -1;
                memcpy(quads + 8 * nq, out.data(), 8 * sizeof(int));
                nq++;
            }
        }
    }
    return nq;
}
