// hd.h -- shared definitions for the HIP kernels (gfx950) of the opencv-ar hot path.
//
// The algorithm cores in *_core.h are written as plain functions so that tests/emul can compile them
// with g++ and check their logic against the oracle without a GPU.  That host build is test-only;
// the product library (libocvar_hip.so) contains the gfx950 code objects and nothing else computes.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define OCVAR_HD __host__ __device__ __forceinline__
#define OCVAR_D __device__ __forceinline__
#define OCVAR_UNROLL _Pragma("unroll")   // small fixed-size loops over arrays that must become registers (constant indices)
#else
#define OCVAR_HD inline
#define OCVAR_D inline
#define OCVAR_UNROLL
#endif

namespace ocvar {

// Pixel-neighbour directions of the border follower, s = 0..7 = E,NE,N,NW,W,SW,S,SE
// (increasing s is counter-clockwise on screen; cvFindContours' icvCodeDeltas order).
OCVAR_HD int dir_dx(int s) { return (s == 0 || s == 1 || s == 7) ? 1 : ((s >= 3 && s <= 5) ? -1 : 0); }
OCVAR_HD int dir_dy(int s) { return (s >= 1 && s <= 3) ? -1 : ((s >= 5 && s <= 7) ? 1 : 0); }

struct Pt { int x, y; };

// Byte offset of pixel (x,y) in a neighbour-mask plane of row stride ns (a multiple of 16).
// Product build (OCVAR_NBR_TILED): the plane is stored as 16x8-pixel tiles of 128 contiguous bytes -- one cache
// line -- because the border follower walks locally in 2-D: with raster rows every vertical step of a border is
// a new 128-byte line (a fresh HBM/MALL miss on a dependent chain), with tiles a walk stays in a line for ~10 steps.
// The host-side test build of the cores keeps plain raster planes.
// (A plane is far smaller than 4 GB; the offset is 32-bit and, on the device, built with the 24-bit multiplier: the
// followers compute it once per step on a dependent chain.)
OCVAR_HD unsigned nbr_addr(int x, int y, int ns) {
#if defined(OCVAR_NBR_TILED)
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned tile = __umul24((unsigned)(y >> 3), (unsigned)(ns >> 4)) + (unsigned)(x >> 4);
#else
    const unsigned tile = (unsigned)(y >> 3) * (unsigned)(ns >> 4) + (unsigned)(x >> 4);
#endif
    return (tile << 7) | ((unsigned)(y & 7) << 4) | (unsigned)(x & 15);
#else
    return (unsigned)(y * ns + x);
#endif
}
// the same offset split into its row part (wave-uniform in the row-marching kernels) and its column part (constant per lane)
OCVAR_HD long long nbr_row_off(int y, int ns) {
#if defined(OCVAR_NBR_TILED)
    return ((long long)((y >> 3) * (ns >> 4)) << 7) + ((y & 7) << 4);
#else
    return (long long)y * ns;
#endif
}
OCVAR_HD unsigned nbr_col_off(int x) {
#if defined(OCVAR_NBR_TILED)
    return ((unsigned)(x >> 4) << 7) + (unsigned)(x & 15);
#else
    return (unsigned)x;
#endif
}
OCVAR_HD long long nbr_plane_bytes(int ns, int sh) { return (long long)ns * ((sh + 7) & ~7); }

// The grey plane is stored in PANELS: panel p of a row holds columns 240 p - 8 .. 240 p + 247 in 256 bytes -- the 256 columns one
// wave of the frame binarise kernel converts (its 240 output columns and the 8-column halo on either side), so that every grey
// store of that kernel is 256 contiguous bytes at a 256-byte boundary: whole 64-byte sectors.  (Stored row-major at the image's
// own pitch, a wave's 240 bytes began at 240 s: five sectors touched, two of them shared with the neighbouring strips' waves; the
// partial sectors made the grey byte the dearest of the five the kernel moves per pixel -- tools/micro/strip_stream.hip: the same
// loads and stores take 2.9 ms per 1024 frames with rows stored whole and 2.2 ms with sector-aligned strips.)  Neighbouring panels
// overlap by 16 columns, so any run of up to 9 consecutive columns lies inside one panel: a reader takes the panel of the run's
// first column.  Row pitch = panels x 256 bytes; the bytes of columns outside the image are never read.
constexpr int GRAY_PANEL_COLS = 240, GRAY_PANEL_BYTES = 256, GRAY_PANEL_LEAD = 8;
OCVAR_HD int gray_pitch(int W) { return ((W + GRAY_PANEL_COLS - 1) / GRAY_PANEL_COLS) * GRAY_PANEL_BYTES; }
OCVAR_HD long long gray_plane_bytes(int W, int H) { return (long long)gray_pitch(W) * H; }
// byte offset of column x (>= 0) in its row, in the panel x belongs to (consecutive columns up to x + 8 follow contiguously)
OCVAR_HD unsigned gray_col(int x) {
    const unsigned p = (unsigned)x / (unsigned)GRAY_PANEL_COLS;
    return p * (unsigned)(GRAY_PANEL_BYTES - GRAY_PANEL_COLS) + (unsigned)x + (unsigned)GRAY_PANEL_LEAD;
}

// One region of interest handed to the square finder: a whole frame (frame pass) or the clipped
// bounding box of a frame-pass quad (crop pass, opencvar.cpp:682-693).
struct Roi {
    int frame;        // frame index in the batch
    int x0, y0;       // origin inside the frame's gray plane
    int w, h;         // ROI size (img->width/height for the border rule, opencvar.cpp:204-206)
    int sw, sh;       // w & ~1, h & ~1 (opencvar.cpp:158): size of the binary / neighbour plane
    int ns;           // row stride of the neighbour plane (sw rounded up to a multiple of 4)
    int owner;        // frame pass: frame index; crop pass: index of the frame-pass quad it came from
    long long nbr_off;  // offset of this ROI's neighbour-mask plane in the pass's pool
};

// A candidate border start found by the binarise kernel.
struct StartCand {
    int roi;
    int pos;          // scan position y*ns + x (ns = plane row stride) at which cvFindContours would discover the border
    int is_hole;
};

// A quad that passed the cvarFindSquares filter.
struct QuadRec {
    int roi;
    int start;        // discovery scan position (orders the sequence: later discovery = earlier in list)
    int pt[8];
};

}  // namespace ocvar
