// kernels.h -- device workspace layout and launch entry points shared by the .hip translation units.
#pragma once
#include "hd.h"
#include "decode_core.h"
#include "pose_core.h"
#include "tail_core.h"
#include "ocvar_hip.h"

namespace ocvar {

constexpr int MAXQ_DEFAULT = OCVAR_MAX_QUADS;   // frame-pass quads kept per frame unless the context was created for more (Workspace::maxq)
constexpr int MAXM = OCVAR_MAX_MARKERS;    // markers kept per frame (tracked + new)
constexpr int MAXT = OCVAR_MAX_TEMPLATES;
constexpr int MARCH_HALO_L = 2, MARCH_HALO_R = 2;   // halo lanes (4 pixels each) left / right of a strip's output lanes
constexpr int MARCH_STRIP = 4 * (64 - MARCH_HALO_L - MARCH_HALO_R);   // 240 output columns of one wave's strip in the binarise kernel (256 loaded)
static_assert(MARCH_STRIP == GRAY_PANEL_COLS && 4 * MARCH_HALO_L == GRAY_PANEL_LEAD && 4 * 64 == GRAY_PANEL_BYTES, "a grey panel is what one wave of the frame kernel converts");
constexpr int MARCH_CROP_ROWS = 256;       // rows per work unit in the crop pass (even)
constexpr int MARCH_STAGE = 512;           // border starts a wave stages in LDS between two appends to the global list
constexpr int BACK_STEPS = 32;             // backward look of an outer start before it follows its border
constexpr int PRE_STEPS = 8;               // steps every plausible start gets before it may queue for tier 1's full budget
constexpr int SHORT_STEPS = 96;            // step budget of follower tier 1 (every plausible start, one lane each)
constexpr int MID_STEPS = 1536;            // step budget of tier 2 (borders that outlived tier 1, one lane each); the rest: tier 3, one wave each
constexpr int SLAB_PTS = 1024;             // points (packed x | y << 16) a tier-2 lane can keep in its private slab (no second follow needed below that)
constexpr int SLAB_STRIDE = SLAB_PTS + 4;  // dwords per tier-2 lane slab: the points + the scratch slot of flat_step
constexpr int MID_BLOCKS_MAX = 1024;       // tier-2 grid limit (x256 threads, one slab each)
constexpr int SLAB3_PTS = 8192;            // points a tier-3 wave can keep in its slab
constexpr int SLAB3_STRIDE = SLAB3_PTS + 64;   // dwords per tier-3 wave slab
constexpr int LONG_BLOCKS_MAX = 1024;      // tier-3 grid limit (x4 waves, one slab each)
constexpr int TILE = 64;                   // side of the LDS tile the wave-per-border follower walks in

// error bits accumulated in Workspace::err[0]
enum { ERR_CAND_OVERFLOW = 1, ERR_POOL_OVERFLOW = 2, ERR_QUAD_OVERFLOW = 4, ERR_TRACE_OVERRUN = 8, ERR_CROP_OVERFLOW = 16,
       ERR_TILE_OVERFLOW = 32,
       ERR_TICKET_RUNAWAY = 64,     // a work-queue loop ran more iterations than its list can account for (control flow broken)
       ERR_MARKER_OVERFLOW = 128 }; // more than OCVAR_MAX_MARKERS markers in one frame

struct TileDesc { int roi, x0, y0; };   // binarise work unit of the crop pass: x0 = strip index, y0 = first row

struct CandRec {   // pre-dedupe candidate, slot [frame][quad][template]
    int valid, orient;
    long long bit;
    float square[8];
    float patPoint[8];
};

// counters block (device ints), zeroed at the start of every batch
enum { CNT_FRAME_CANDS = 0, CNT_CROP_ROIS = 1, CNT_CROP_TILES = 2, CNT_CROP_CANDS = 3,
       CNT_POOL_INTS = 4 /* 64-bit, uses 4..5 */, CNT_CROP_QUADS = 6, CNT_TICKET_F = 7, CNT_TICKET_C = 8, CNT_ERR = 9,
       CNT_CROP_PIXELS = 10 /* 64-bit, uses 10..11 */, CNT_LONG_F = 12, CNT_LONG_C = 13, CNT_TICKET_LF = 14, CNT_TICKET_LC = 15,
       CNT_MID_F = 16, CNT_MID_C = 17, CNT_TICKET_MF = 18, CNT_TICKET_MC = 19, CNT_TICKET_BC = 20, CNT_MID_C_FIRST = 21, CNT_TICKET_MC2 = 22,
       CNT_POSE_JOBS = 23,
       CNT_PROF = 24 /* 32 64-bit profiling slots, written only by builds with -DOCVAR_PROF (tools/prof_tier2.py) */, CNT_COUNT = 88 };

struct Workspace {
    // limits
    int max_w, max_h, max_batch;
    int maxq;               // frame-pass quads kept per frame (ocvar_hip_create: OCVAR_MAX_QUADS; ocvar_hip_create_ex: caller's choice)
    int maxc;               // pre-dedupe candidates per frame the tail can replay = maxq * n_templates, bounded by its LDS
    int cap_frame_cands, cap_crop_cands, cap_crop_rois, cap_crop_tiles, cap_crop_quads;
    long long cap_pool_ints, cap_crop_pixels;
    // per batch geometry
    int W, H, sw, sh, ns, n_frames, n_templates;   // ns: row stride of a neighbour-mask plane = sw rounded up to 4
    int crop_phases;                         // 2: crop tier 2 in two launches (earliest starts first, then the rest behind exact pruning); 1: one launch (few frames: the shorter chain)
    int mid_steps, mid_blocks, long_blocks;  // tuning (env OCVAR_MID_STEPS / OCVAR_MID_BLOCKS / OCVAR_LONG_BLOCKS): tier-2 step budget and grid, tier-3 grid
    int max_mid_blocks, max_long_blocks;     // slabs allocated at create (scaled with max_batch)
    int short_blocks, crop_blocks;           // grids of follower tier 1 and of the crop binarise kernel (scaled with the batch)
    int frame_strips, frame_chunks, frame_chunk_rows;  // binarise work decomposition of a frame
    // device buffers
    uint8_t* gray;          // [B][H][gray_pitch(W)]: panels of 256 bytes (hd.h::gray_col)
    uint8_t* nbr_frame;     // [B][sh][sw]
    uint8_t* nbr_crop;      // crop pool
    StartCand* cands_frame;
    StartCand* cands_crop;
    StartCand* mid_frame;   // starts whose border exceeded tier 1's step budget
    StartCand* mid_crop;
    StartCand* mid_first_crop;   // crop starts tier 2 takes first (expected longest walks), counter CNT_MID_C_FIRST, capacity cap_long
    StartCand* long_frame;  // starts whose border exceeded tier 2's step budget
    StartCand* long_crop;
    int cap_long;
    int* pool;              // points + DP stacks
    int* slab;              // [max_mid_blocks*256][SLAB_STRIDE] private point space of the tier-2 lanes
    int* slab3;             // [max_long_blocks*4][SLAB3_STRIDE] point space of the tier-3 waves
    QuadRec* quads_frame;   // [B][maxq] unordered
    int* n_quads_frame;     // [B]
    float* squares;         // [B][maxq][8] ordered, after tracking
    int* n_squares;         // [B]
    int* crop_of;           // [B][maxq] crop ROI index of square i, or -1
    Roi* rois_crop;
    TileDesc* tiles_crop;
    unsigned long long* crop_pixels;   // = counters + CNT_CROP_PIXELS: running sum of crop plane sizes (pool cursor)
    QuadRec* quads_crop;    // pool
    unsigned long long* best_crop;     // [cap_crop_rois] (start<<32 | quad slot), ~0 = none
    int* ring_frame;        // [B] 1: the frame's own frame border is the rectangle ring_quads_kernel has published (tier 1 drops its start)
    int* ring_crop;         // [cap_crop_rois] the same for a crop
    int* crop_min_rest;     // [cap_crop_rois] smallest start position among a crop's tier-2 starts off the crop's frame (tier 2 walks these first)
    CandRec* cand_recs;     // [B][maxq][MAXT]
    MarkerRec* prev;        // [B][MAXM]
    int* n_prev;            // [B]
    int* reserve;           // [B][MAXM] tracked marker indices
    int* n_reserve;         // [B]
    int* pose_jobs;         // [B][MAXM] frame * MAXM + slot of every output marker (counter CNT_POSE_JOBS): pose_kernel's work list
    MarkerRec* markers;     // [B][MAXM] output
    int* n_markers;         // [B]
    TemplateRec* templates; // [MAXT]
    CameraRec* camera;
    int* counters;          // [CNT_COUNT]
};

// launchers (each enqueues on `stream`, no synchronisation)
void launch_binarise_frames(const Workspace& ws, const uint8_t* d_bgr, int row_stride, size_t frame_stride, int grey_in_place,
                            hipStream_t stream);
void launch_binarise_crops(const Workspace& ws, hipStream_t stream);
void launch_ring_quads_frames(const Workspace& ws, hipStream_t stream);   // the ROIs' own frame borders without a walk (before tier 1)
void launch_ring_quads_crops(const Workspace& ws, hipStream_t stream);
void launch_follow_frames(const Workspace& ws, hipStream_t stream);
void launch_follow_crops(const Workspace& ws, hipStream_t stream);
void launch_follow_mid_frames(const Workspace& ws, hipStream_t stream);
void launch_follow_mid_crops(const Workspace& ws, hipStream_t stream);   // both phases
void launch_follow_long_frames(const Workspace& ws, hipStream_t stream);
void launch_follow_long_crops(const Workspace& ws, hipStream_t stream);
void launch_order_and_crops(const Workspace& ws, hipStream_t stream);
void launch_decode(const Workspace& ws, hipStream_t stream);
void launch_finalise(const Workspace& ws, hipStream_t stream);

}  // namespace ocvar
