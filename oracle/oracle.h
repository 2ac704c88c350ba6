/*
 * oracle.h -- CPU restatement of the youtalk/opencv-ar hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This directory is the parity oracle.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it.  Nothing under opencv-ar_amd/ links, includes
 * or dlopens anything from here; the product path fails loudly without its HIP library.
 *
 * What it restates (reference file:line, relative to /root/reference):
 *   src/opencvar.cpp:156-223   cvarFindSquares           -> orc_find_squares
 *   src/opencvar.cpp:619-807   cvarArMultRegistration    -> orc_registration
 *   src/opencvar.cpp:524-540   cvarSquareToMatrix (+261-278, 133-152, 229-245) -> orc_square_to_matrix
 *   src/opencvar.cpp:284-321   cvarLoadTemplateTag / cvarLoadTag -> orc_load_template_pixels / orc_load_tag
 *   src/acmath.cpp:201-276,293-298,486-580  (bit codec, quaternion tail) -> orc_ac*
 * and, because every pixel-touching step of those functions is a call into OpenCV's legacy C API
 * (not vendored, not pinned, not installed here), the published OpenCV 2.4.x algorithms behind
 * each call site: cvCvtColor, cvPyrDown, cvPyrUp, cvAdaptiveThreshold,
 * cvFindContours, cvContourPerimeter, cvApproxPoly, cvContourArea, cvCheckContourConvexity,
 * cvGetPerspectiveTransform, cvWarpPerspective, cvThreshold, cvFindExtrinsicCameraParams2,
 * cvRodrigues2.
 *
 * PINNING STATUS
 *   - acmath subset: PINNED against oracle/_ref/libacmath_ref.so (the reference's own
 *     src/acmath.cpp compiled in place) and the golden values in tests/golden/acmath_golden.json.
 *   - everything at the OpenCV boundary: PARITY UNPINNED.  The reference ships no tests, no golden
 *     vectors, and OpenCV is not installable here; the restatement follows OpenCV 2.4.x as recalled
 *     and is only checked for internal consistency.
 */
#ifndef ORC_ORACLE_H
#define ORC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Layout-compatible with include/opencvar structs (opencvar.h:54-82). */
typedef struct { int width, height; double cameraMatrix[9]; double distCoeffs[5]; double glProjection[16]; } OrcCamera;
typedef struct { int width, height; double scale; long long code[4]; } OrcTemplate;
typedef struct { float x, y; } OrcPoint2f;
typedef struct {
    double glMatrix[16]; int templateId; int markerId; double score; OrcPoint2f square[4]; double aspectRatio;
} OrcMarker;

/* Pre-dedupe candidate record (debug hook). */
typedef struct {
    int markerId, templateId, orient, pad;
    long long bit;
    float square[8];      /* after the orient 2/4 rotation quirk */
    float patPoint[8];    /* last quad of the crop pass, crop coordinates */
} OrcCandidate;

/* ---- image stages (A.1-A.4) ---- */
void orc_bgr2gray(const uint8_t* bgr, int w, int h, int stride, uint8_t* gray, int gstride);
/* pyrDown+pyrUp+adaptive threshold of a single-channel image (all three channels of the reference's
 * working image are equal after the in-place grey, opencvar.cpp:625-626).  Output is (w&~1)x(h&~1),
 * values 0/255, row stride = w&~1.  up (optional) receives the pyrUp result (same size). */
void orc_binarise(const uint8_t* gray, int w, int h, int stride, uint8_t* bin, uint8_t* up);

/* ---- contours (A.5-A.8) ---- */
/* Literal sequential Suzuki-Abe scan, RETR_LIST + CHAIN_APPROX_SIMPLE.  bin is w x h 0/nonzero and is
 * NOT modified.  Contours are returned in OpenCV list order (reverse discovery).  pts: x,y pairs
 * concatenated; offs[i]..offs[i+1] index pts (in points); starts[i] = scan position y*w+x at which
 * contour i was discovered; holes[i] = 1 for hole borders.  Returns number of contours, or -1 if a
 * capacity was exceeded. */
int orc_find_contours(const uint8_t* bin, int w, int h, int* pts, int max_pts, int* offs, int* starts,
                      int* holes, int max_contours);
double orc_arc_length_closed(const int* pts, int n);
/* Douglas-Peucker as cvApproxPoly(CV_POLY_APPROX_DP) on a closed int contour; returns output count. */
int orc_approx_poly(const int* pts, int n, double eps, int* out);
/* Which OpenCV release's cvApproxPoly the oracle restates (process-wide; orc_contours.cpp): 0 = <= 2.4.3 legacy integer routine,
 * float accuracy (default, the parity definition); 1 = 2.4.4+ approxPolyDP_<int>, double accuracy; 2 = 1 + the inner-product
 * clean-up condition of later releases.  Only tools/approx_variants.py switches it. */
void orc_set_approx_variant(int v);
int orc_get_approx_variant(void);
double orc_contour_area(const int* pts, int n);
int orc_is_convex(const int* pts, int n);

/* cvarFindSquares on a single-channel image: returns number of quads, quads[8*i..] = 4 int points,
 * in the order the reference pushes them (opencvar.cpp:187-214). */
int orc_find_squares(const uint8_t* gray, int w, int h, int stride, int* quads, int max_quads);

/* ---- warp + readout (A.9-A.11) ---- */
void orc_get_perspective_transform(const float* src8, const float* dst8, float* m9);
void orc_warp_perspective_gray(const uint8_t* src, int sw, int sh, int sstride, const float* m9, uint8_t* dst,
                               int dw, int dh);
long long orc_readout_bits(const uint8_t* gray, int w, int h, int stride, const float* patPoint8, int tw, int th);

/* ---- acmath subset ---- */
void orc_acArray2DToBit(const unsigned char* arr, int w, int h, long long* bit);
void orc_acBitToArray2D(long long bit, unsigned char* arr, int w, int h);
void orc_acArray2DRotateub(unsigned char* arr, int w, int h, int rot);
void orc_acBitRotate(long long* bit, int rot, int w, int h);
void orc_acMatrixToQuaternion(const double* m, double* q);
void orc_acQuaternionToMatrix(const double* q, double* m);
void orc_acMatrixTranspose(double* m);
double orc_acCalcLength(double x1, double y1, double x2, double y2);

/* ---- templates / camera (setup side) ---- */
void orc_load_tag(OrcTemplate* tpl, long long bit, int width, int height, double scale);
/* pixels: full template image incl. the 1-px frame, row-major, top row first, (w x h), 8-bit gray. */
void orc_load_template_pixels(OrcTemplate* tpl, const uint8_t* pixels, int w, int h, double scale);
void orc_camera_default(OrcCamera* cam);
void orc_camera_scale(OrcCamera* cam, int width, int height);

/* ---- pose (A.12 + opencvar.cpp:133-152) ---- */
void orc_rodrigues_vec2mat(const double* r, double* R9, double* J27);
void orc_rodrigues_mat2vec(const double* R9, double* r);
/* dist5: (k1, k2, p1, p2, k3) or NULL */
void orc_find_extrinsic(const double* obj12, const double* img8, const double* K9, const double* dist5, double* rvec, double* tvec);
void orc_gl_matrix(const double* R9, const double* t3, double* m16);
void orc_square_to_matrix(const float* pts8, const OrcCamera* cam, double ratio, double* m16);

/* ---- registration ---- */
/* bgr is greyed in place (opencvar.cpp:624-627).  markers: in = previous markers (n_in), out = result.
 * cands (optional): pre-dedupe candidate list.  Returns markers->size(). */
int orc_registration(uint8_t* bgr, int w, int h, int stride, OrcMarker* markers, int n_in, int max_markers,
                     const OrcTemplate* templates, int n_templates, const OrcCamera* cam, OrcCandidate* cands,
                     int max_cands, int* n_cands);

/* Measurement aid (bench.py cpu_baseline): orc_registration frame-parallel over `threads` host threads (<= 0: all hardware
 * threads) for about budget_s seconds; returns the number of frames processed, *seconds the wall time. */
long long orc_registration_throughput(const uint8_t* frames, int n_frames, int w, int h, int stride, size_t frame_stride,
                                      const OrcTemplate* templates, int n_templates, const OrcCamera* cam, int threads,
                                      double budget_s, long long max_calls, double* seconds, int* threads_used);

#ifdef __cplusplus
}
#endif
#endif
