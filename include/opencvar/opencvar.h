/*
 * opencvar/opencvar.h -- public API of the opencvar AR-marker library, MI355X edition.
 *
 * Source- and link-compatible with the reference header (/root/reference/include/opencvar/opencvar.h:54-262):
 * the three POD structs, the function names, argument order, default arguments and C linkage are the same, so a
 * caller such as samples/ARTest.cpp builds and links unchanged.  What differs is underneath:
 * cvarArMultRegistration runs on an MI355X through the C ABI of include/ocvar_hip.h.
 *
 * OpenCV types appear only in signatures.  With real OpenCV installed its headers are used; otherwise the
 * layout-compatible stand-ins of include/shim/opencv are picked up through the same include path.
 */
#pragma once

#ifdef ACDLL
#define AC_DLL __declspec(dllexport)
#else
#define AC_DLL
#endif

#include <vector>
#include <opencv/cv.h>
#include <opencv/highgui.h>

using namespace std;  // the reference header does this at global scope; callers rely on it

/* Pinhole camera: OpenCV-style intrinsics plus the matching OpenGL projection. */
struct CvarCamera {
    int width;
    int height;
    double cameraMatrix[9];
    double distCoeffs[5];
    double glProjection[16];
};

/* A marker template: inner grid size and its code in the four 90-degree rotations. */
struct CvarTemplate {
    int width;
    int height;
    double scale;
    long long int code[4];
};

/* One detected marker. */
struct CvarMarker {
    double glMatrix[16];      /* OpenGL modelview */
    int templateId;           /* index of the matching template */
    int markerId;             /* index of the quad in the frame's square list */
    double score;             /* 1 = code matched, 0 = not */
    CvPoint2D32f square[4];   /* quad corners in the frame */
    double aspectRatio;       /* template width : height */
};

extern "C" {

/* filename == NULL: 640x480, f = 500 defaults.  Returns 1 on success, 0 on failure. */
AC_DLL int cvarReadCamera(const char* filename, CvarCamera* pCam);
AC_DLL void cvarCameraScale(CvarCamera* pCam, int width, int height);
AC_DLL void cvarCameraProjection(CvarCamera* pCam, double* projection, int glstyle = 0);
AC_DLL void cvarGlMatrix(double* modelview, CvMat* rotate3, CvMat* translate);

/* Square finder on an 8-bit 3-channel image.  Returns an opaque sequence for cvarGetSquare/cvarGetAllSquares. */
AC_DLL CvSeq* cvarFindSquares(IplImage* img, CvMemStorage* storage);
AC_DLL void cvarSquareInit(CvMat* mat, double ratio = 1);
AC_DLL void cvarReverseSquare(CvPoint2D32f sq[4]);
AC_DLL void cvarFindCamera(CvarCamera* cam, CvMat* objPts, CvMat* imgPts, double* modelview);

AC_DLL int cvarLoadTemplateTag(CvarTemplate* tpl, const char* filename, double scale = 0.01);
AC_DLL void cvarLoadTag(CvarTemplate* tpl, long long int bit, int width = 8, int height = 8, double scale = 0.01);

AC_DLL int cvarCompareSquare(IplImage* img, CvPoint2D32f* points);
AC_DLL int cvarDrawSquares(IplImage* img, CvSeq* squares);
AC_DLL int cvarGetSquare(CvSeq* squares, CvPoint2D32f* points);
AC_DLL void cvarSquare(CvPoint2D32f* src, int width, int height, int ccw = 1);
AC_DLL void cvarRotSquare(CvPoint2D32f* src, int rot);
AC_DLL void cvarInvertPerspective(IplImage* input, IplImage* output, CvPoint2D32f* src, CvPoint2D32f* dst);
AC_DLL void cvarSquareToMatrix(CvPoint2D32f* points, CvarCamera* cam, double* modelview, double ratio = 1);

CvRect cvarSquare2Rect(CvPoint2D32f pt[4]);
int cvarGetAllSquares(CvSeq* squares, vector<CvPoint2D32f>* pts);
int cvarTrack(CvPoint2D32f pt1[4], CvPoint2D32f pt2[4]);

/* Detects and registers all markers of one frame.  `image` (8UC3 BGR) is greyed in place; `markers` carries
 * the previous frame's markers in and the tracked + new markers out.  Returns markers->size(). */
int cvarArMultRegistration(IplImage* image, vector<CvarMarker>* markers, vector<CvarTemplate> templates,
                           CvarCamera* camera);

/* EXTENSION (not in the reference's header; nothing in the reference calls it).  The sequences cvarFindSquares returns live in
 * the caller's CvMemStorage in the reference, i.e. until cvClearMemStorage / cvReleaseMemStorage (opencvar.cpp:168).  This
 * library is built without OpenCV, so it keeps them itself, keyed by the storage pointer that was passed (NULL is a key too):
 * a storage's sequences stay valid until this call, or until that same storage has collected more than 4096 of them (then
 * its oldest goes).  Call it where the reference's caller clears or releases the storage. */
AC_DLL void cvarReleaseSquares(CvMemStorage* storage);

}  // extern "C"
