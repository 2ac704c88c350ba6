// orc_registration.cpp -- oracle: cvarArMultRegistration glue (TEST INFRASTRUCTURE, see oracle.h).
//
// Follows opencvar.cpp:619-807 step by step, including the behaviour-defining quirks of SURVEY
// Appendix D: D2 (|| dedupe), D4 (orient 2/4 rotation leaks across templates), D5 (score-0
// candidates), D6 (last quad of the crop pass), D9 (erase-while-iterating in tracking), D10.
// Helper restatements: cvarSquare2Rect 546-562, cvarTrack 592-617, cvarRotSquare 464-501.
#include "oracle.h"
#include <cstring>
#include <vector>

namespace {

struct Rect { int x, y, w, h; };

Rect square2rect(const OrcPoint2f* pt) {  // opencvar.cpp:546-562 (float -> int truncation on assignment)
    int x = 50000, y = 50000, x2 = -50000, y2 = -50000;
    for (int i = 0; i < 4; i++) {
        if (pt[i].x < x) x = (int)pt[i].x;
        if (pt[i].x > x2) x2 = (int)pt[i].x;
        if (pt[i].y < y) y = (int)pt[i].y;
        if (pt[i].y > y2) y2 = (int)pt[i].y;
    }
    return Rect{x, y, x2 - x, y2 - y};
}

void rot_square(OrcPoint2f* src, int rot) {  // opencvar.cpp:464-501
    OrcPoint2f t[4];
    memcpy(t, src, sizeof t);
    for (int i = 0; i < 4; i++) src[(rot - 1 + i) % 4] = t[i];
}

int track(OrcPoint2f* pt1, const OrcPoint2f* pt2) {  // opencvar.cpp:592-617
    for (int j = 0; j < 4; j++) {
        int res = 0;
        for (int i = 0; i < 4; i++)
            if (orc_acCalcLength(pt1[i].x, pt1[i].y, pt2[(i + j) % 4].x, pt2[(i + j) % 4].y) < 20) res++;
        if (res == 4) {
            for (int i = 0; i < 4; i++) pt1[i] = pt2[(i + j) % 4];
            return 1;
        }
    }
    return 0;
}

}  // namespace

extern "C" int orc_registration(uint8_t* bgr, int w, int h, int stride, OrcMarker* markers_io, int n_in,
                                int max_markers, const OrcTemplate* templates, int n_templates, const OrcCamera* cam,
                                OrcCandidate* cands_out, int max_cands, int* n_cands_out) {
    // A. grey in place (624-627)
    std::vector<uint8_t> gray((size_t)w * h);
    orc_bgr2gray(bgr, w, h, stride, gray.data(), w);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            uint8_t g = gray[(size_t)y * w + x];
            uint8_t* p = bgr + (size_t)y * stride + 3 * x;
            p[0] = p[1] = p[2] = g;
        }
    // B. frame pass (630-632)
    std::vector<int> quads(8 * 4096);
    int nq = orc_find_squares(gray.data(), w, h, w, quads.data(), 4096);
    if (nq < 0) nq = 0;
    std::vector<OrcPoint2f> squares((size_t)nq * 4);
    for (int i = 0; i < nq * 4; i++) {
        squares[i].x = (float)quads[2 * i];
        squares[i].y = (float)quads[2 * i + 1];
    }
    // C. tracking (635-668)
    std::vector<OrcMarker> markers(markers_io, markers_io + n_in);
    std::vector<int> reserve;
    for (size_t i = 0; i < markers.size(); i++) {
        for (size_t j = 0; j < squares.size(); j += 4) {
            OrcPoint2f points[4];
            for (int k = 0; k < 4; k++) points[k] = squares[j + k];
            if (track(markers[i].square, points)) {
                reserve.push_back((int)i);
                squares.erase(squares.begin() + j, squares.begin() + j + 4);
                orc_square_to_matrix(&markers[i].square[0].x, cam, markers[i].aspectRatio, markers[i].glMatrix);
            }
        }
    }
    std::vector<OrcMarker> copy = markers;
    markers.clear();
    for (size_t i = 0; i < reserve.size(); i++) markers.push_back(copy[reserve[i]]);

    // D. candidate loop (675-777)
    OrcPoint2f points[4];
    std::vector<OrcMarker> candidates;
    std::vector<OrcCandidate> dbg;
    for (int i = 0; i < (int)(squares.size() / 4); i++) {
        for (int j = 0; j < 4; j++) points[j] = squares[(size_t)i * 4 + j];
        Rect r = square2rect(points);
        r.x -= 5;
        r.y -= 5;
        r.w += 10;
        r.h += 10;
        // cvSetImageROI clips the rectangle to the image (A.9)
        int x0 = r.x < 0 ? 0 : r.x, y0 = r.y < 0 ? 0 : r.y;
        int x1 = r.x + r.w > w ? w : r.x + r.w, y1 = r.y + r.h > h ? h : r.y + r.h;
        int cw = x1 - x0, ch = y1 - y0;
        if (cw <= 0 || ch <= 0) continue;
        const uint8_t* crop = gray.data() + (size_t)y0 * w + x0;
        OrcPoint2f patPoint[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
        for (int j = 0; j < n_templates; j++) {
            // cvarGetSquare(cvarFindSquares(crop)) -- last quad wins (401-430)
            std::vector<int> cq(8 * 1024);
            int ncq = orc_find_squares(crop, cw, ch, w, cq.data(), 1024);
            if (ncq <= 0) continue;
            for (int k = 0; k < 4; k++) {
                patPoint[k].x = (float)cq[8 * (ncq - 1) + 2 * k];
                patPoint[k].y = (float)cq[8 * (ncq - 1) + 2 * k + 1];
            }
            OrcMarker marker;
            memset(&marker, 0, sizeof marker);
            marker.markerId = i;
            int tw = templates[j].width, th = templates[j].height;
            long long bit = orc_readout_bits(crop, cw, ch, w, &patPoint[0].x, tw, th);
            int orient = 0;
            for (int k = 0; k < 4; k++)
                if (bit == templates[j].code[k]) {
                    orient = k + 1;
                    break;
                }
            marker.score = orient ? 1.0 : 0.0;
            marker.templateId = j;
            if (orient == 4) rot_square(points, 2);
            else if (orient == 2) rot_square(points, 4);
            memcpy(marker.square, points, sizeof points);
            marker.aspectRatio = (double)tw / th;
            candidates.push_back(marker);
            OrcCandidate d;
            memset(&d, 0, sizeof d);
            d.markerId = i;
            d.templateId = j;
            d.orient = orient;
            d.bit = bit;
            memcpy(d.square, points, sizeof points);
            memcpy(d.patPoint, patPoint, sizeof patPoint);
            dbg.push_back(d);
        }
    }
    // E. dedupe (780-792)
    for (size_t i = 0; i < candidates.size(); i++)
        for (size_t j = 0; j < i; j++)
            if (candidates[i].markerId == candidates[j].markerId || candidates[i].templateId == candidates[j].templateId) {
                if (candidates[i].score > candidates[j].score)
                    candidates[j].markerId = -1;
                else
                    candidates[i].markerId = -1;
            }
    // F. output (795-801)
    for (size_t i = 0; i < candidates.size(); i++)
        if (candidates[i].templateId >= 0 && candidates[i].markerId >= 0) {
            orc_square_to_matrix(&candidates[i].square[0].x, cam, candidates[i].aspectRatio, candidates[i].glMatrix);
            markers.push_back(candidates[i]);
        }
    if (n_cands_out) *n_cands_out = (int)dbg.size();
    if (cands_out)
        for (size_t i = 0; i < dbg.size() && (int)i < max_cands; i++) cands_out[i] = dbg[i];
    int n = (int)markers.size() < max_markers ? (int)markers.size() : max_markers;
    for (int i = 0; i < n; i++) markers_io[i] = markers[i];
    return (int)markers.size();
}
