/*
 * Minimal, layout-compatible stand-in for the OpenCV legacy C types that the opencvar public header mentions
 * (IplImage, CvMat, CvSeq, CvMemStorage, CvPoint, CvPoint2D32f, CvRect, CvSize).  Used ONLY when real OpenCV
 * headers are not installed (this image has none): callers built against real OpenCV 2.x/3.x pass the same
 * struct layouts (LP64: sizeof(IplImage) == 144, width@40, imageData@88, widthStep@96).
 * No OpenCV function is declared or implemented here; the detection path does not use any.
 */
#ifndef OCVAR_SHIM_OPENCV_CV_H
#define OCVAR_SHIM_OPENCV_CV_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct CvPoint { int x, y; } CvPoint;
typedef struct CvPoint2D32f { float x, y; } CvPoint2D32f;
typedef struct CvSize { int width, height; } CvSize;
typedef struct CvRect { int x, y, width, height; } CvRect;

typedef struct _IplROI { int coi, xOffset, yOffset, width, height; } IplROI;

typedef struct _IplImage {
    int nSize;
    int ID;
    int nChannels;
    int alphaChannel;
    int depth;
    char colorModel[4];
    char channelSeq[4];
    int dataOrder;
    int origin;
    int align;
    int width;
    int height;
    struct _IplROI* roi;
    struct _IplImage* maskROI;
    void* imageId;
    void* tileInfo;
    int imageSize;
    char* imageData;
    int widthStep;
    int BorderMode[4];
    int BorderConst[4];
    char* imageDataOrigin;
} IplImage;

typedef struct CvMat {
    int type;
    int step;
    int* refcount;
    int hdr_refcount;
    union { unsigned char* ptr; short* s; int* i; float* fl; double* db; } data;
    int rows;
    int cols;
} CvMat;

typedef struct CvMemStorage CvMemStorage; /* opaque: only passed through */

/* CvSeq as OpenCV 2.x/3.x lay it out (CV_TREE_NODE_FIELDS + CV_SEQUENCE_FIELDS, types_c.h): a caller compiled against the
 * real headers finds `total` at offset 40, `elem_size` at 44 and the element blocks behind `first`, so reading
 * seq->total or walking the blocks (what cvGetSeqElem does) works on the sequences cvarFindSquares returns. */
typedef struct CvSeqBlock {
    struct CvSeqBlock* prev;
    struct CvSeqBlock* next;
    int start_index;
    int count;
    signed char* data;
} CvSeqBlock;

typedef struct CvSeq {
    int flags;
    int header_size;
    struct CvSeq* h_prev;
    struct CvSeq* h_next;
    struct CvSeq* v_prev;
    struct CvSeq* v_next;
    int total;
    int elem_size;
    signed char* block_max;
    signed char* ptr;
    int delta_elems;
    CvMemStorage* storage;
    CvSeqBlock* free_blocks;
    CvSeqBlock* first;
} CvSeq;

#define IPL_DEPTH_8U 8

#ifdef __cplusplus
}
#endif
#endif
