/*
 * ocvar_hip.h -- thin C ABI of the MI355X (gfx950) AR-marker detection path.
 *
 * This is the drop-in boundary for the reference's one hot path, cvarArMultRegistration
 * (/root/reference/include/opencvar/opencvar.h:259-260, /root/reference/src/opencvar.cpp:619-807) and the
 * functions under it.  Plain pointers and sizes only; no C++ or torch types.  The host-side C++ mirror of
 * the reference API (include/opencvar/opencvar.h in this repo) is a thin caller of these entry points.
 *
 * Which reference interface each entry point stands in for:
 *   ocvar_hip_set_templates  <- the vector<CvarTemplate> argument (opencvar.h:65-70, filled by
 *                               cvarLoadTemplateTag/cvarLoadTag, opencvar.cpp:284-321)
 *   ocvar_hip_set_camera     <- the CvarCamera* argument (opencvar.h:54-60; cvarReadCamera/cvarCameraScale,
 *                               opencvar.cpp:39-104)
 *   ocvar_hip_detect_*       <- cvarArMultRegistration for a batch of independent frames (stateless when
 *                               prev == NULL; with prev it replays the tracking stage, opencvar.cpp:635-668)
 *   ocvar_hip_find_squares   <- cvarFindSquares + cvarGetAllSquares (opencvar.h:131,240; opencvar.cpp:156-223,564-590)
 *   ocvar_hip_debug_*        <- no reference counterpart: parity hooks (binary image, pre-dedupe candidates)
 *
 * All functions return 0 on success and a negative OCVAR_E_* code on failure; none throws.  There is no
 * CPU fallback: without a gfx950 device ocvar_hip_create fails.
 */
#ifndef OCVAR_HIP_H
#define OCVAR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Layout-identical to the reference's PODs (opencvar.h:54-82); sizes 248 / 48 / 184 bytes. */
typedef struct { int width, height; double cameraMatrix[9]; double distCoeffs[5]; double glProjection[16]; } OcvarCamera;
typedef struct { int width, height; double scale; long long code[4]; } OcvarTemplate;
typedef struct {
    double glMatrix[16]; int templateId; int markerId; double score; float square[8]; double aspectRatio;
} OcvarMarker;

/* Pre-dedupe candidate (one per frame-pass quad with a crop-pass quad, per template); SURVEY.md 8(d). */
typedef struct {
    int markerId, templateId, orient, valid;
    long long bit;
    float square[8];
    float patPoint[8];
} OcvarCandidate;

typedef struct OcvarHip OcvarHip;

enum {
    OCVAR_OK = 0,
    OCVAR_E_NO_DEVICE = -1,   /* no HIP device / not gfx950 */
    OCVAR_E_ARG = -2,
    OCVAR_E_HIP = -3,         /* a HIP runtime call failed; see ocvar_hip_last_error */
    OCVAR_E_CAPACITY = -4     /* a device work list overflowed (frame too cluttered for the configured limits) */
};

enum { OCVAR_MAX_TEMPLATES = 16, OCVAR_MAX_QUADS = 256, OCVAR_MAX_MARKERS = 64,
       OCVAR_MAX_QUADS_EX = 1792 /* most squares per frame a context can be created for (ocvar_hip_create_ex) */ };

/* Creates a context on `device` with workspace for batches of up to max_batch frames of up to
 * max_width x max_height pixels. */
int ocvar_hip_create(OcvarHip** ctx, int device, int max_width, int max_height, int max_batch);
/* Same with room for max_quads (1 .. OCVAR_MAX_QUADS_EX) frame-pass squares per frame instead of OCVAR_MAX_QUADS: the
 * reference's square list is unbounded (opencvar.cpp:187-214); a caller that gets OCVAR_E_CAPACITY with flag 4 (squares)
 * repeats the frame on such a context (the host mirror of cvarArMultRegistration / cvarFindSquares does). */
int ocvar_hip_create_ex(OcvarHip** ctx, int device, int max_width, int max_height, int max_batch, int max_quads);
void ocvar_hip_destroy(OcvarHip* ctx);
const char* ocvar_hip_last_error(const OcvarHip* ctx);
/* Flag word of the last call that failed with OCVAR_E_CAPACITY: 1 start lists, 2 point pool, 4 squares / candidates per
 * frame, 8 trace overrun, 16 crops, 32 crop tiles, 64 work-queue runaway, 128 markers per frame.  0 after a good call. */
int ocvar_hip_capacity_flags(const OcvarHip* ctx);

/* Several contexts in flight on one GPU (each with its own stream) overlap the latency-bound border followers of one batch
 * with the streaming binarise kernels of another -- but left alone they drift: a quarter of the time none of them is in a
 * binarise kernel and a sixth of the time three are (rocprofv3 trace of bench.py).  A gate shared by the
 * contexts lets at most `width` binarise kernels run at once (stream-ordered: the n-th binarise launch waits for the
 * (n - width)-th to finish; nothing blocks on the host), so that more contexts can be kept in flight to always have one
 * ready.  No counterpart in the reference (it processes one frame per call). */
typedef struct OcvarGate OcvarGate;
int ocvar_hip_gate_create(OcvarGate** gate, int device, int width);
void ocvar_hip_gate_destroy(OcvarGate* gate);
/* gate may be NULL (no gate: the default).  The gate must outlive the batches enqueued under it; it keeps its order on the
 * host without locks, so the contexts that share one are enqueued from one host thread. */
int ocvar_hip_set_gate(OcvarHip* ctx, OcvarGate* gate);

int ocvar_hip_set_templates(OcvarHip* ctx, const OcvarTemplate* templates, int n);
int ocvar_hip_set_camera(OcvarHip* ctx, const OcvarCamera* camera);

/* Batch detection on frames already resident in device memory.
 *   d_bgr        8UC3 interleaved BGR, frame f starts at d_bgr + f*frame_stride, rows row_stride bytes apart
 *   grey_in_place non-zero: overwrite each frame with its grey version (the reference's side effect,
 *                opencvar.cpp:624-627)
 *   prev / prev_counts  host arrays [n_frames][OCVAR_MAX_MARKERS] and [n_frames] of the previous call's
 *                markers per stream, or NULL for stateless detection
 *   markers / counts    host outputs [n_frames][max_per_frame] and [n_frames]; counts[f] is the reference's
 *                return value for frame f (may exceed max_per_frame; only the first max_per_frame are stored)
 * ocvar_hip_detect_device = enqueue + wait + copy-out.  The enqueue/collect pair lets a caller time or
 * overlap the device work; `stream` is a hipStream_t (NULL = the context's own stream). */
int ocvar_hip_detect_device(OcvarHip* ctx, uint8_t* d_bgr, int width, int height, int row_stride, size_t frame_stride,
                            int n_frames, int grey_in_place, const OcvarMarker* prev, const int* prev_counts,
                            OcvarMarker* markers, int* counts, int max_per_frame);
int ocvar_hip_enqueue(OcvarHip* ctx, uint8_t* d_bgr, int width, int height, int row_stride, size_t frame_stride,
                      int n_frames, int grey_in_place, const OcvarMarker* prev, const int* prev_counts, void* stream);
int ocvar_hip_collect(OcvarHip* ctx, OcvarMarker* markers, int* counts, int max_per_frame);
/* ocvar_hip_enqueue with the previous markers in DEVICE memory: d_prev [n_frames][OCVAR_MAX_MARKERS], d_prev_counts [n_frames]
 * (counts above OCVAR_MAX_MARKERS are read as OCVAR_MAX_MARKERS), e.g. the block ocvar_hip_results_to_device wrote for the
 * same streams one time step earlier -- the tracking state of a block of video streams (the caller-owned `markers` vector of
 * cvarArMultRegistration, /root/reference/src/opencvar.cpp:635-668, samples/ARTest.cpp:57) then never leaves the device.
 * The copy into the context is stream-ordered: the arrays may be overwritten by ocvar_hip_results_to_device of the same
 * batch. */
int ocvar_hip_enqueue_tracked(OcvarHip* ctx, uint8_t* d_bgr, int width, int height, int row_stride, size_t frame_stride,
                              int n_frames, int grey_in_place, const OcvarMarker* d_prev, const int* d_prev_counts, void* stream);
/* 1 if the enqueued batch has finished (ocvar_hip_collect will not wait), 0 if it is still running, < 0 on error or when
 * nothing is enqueued.  A caller with several contexts in flight collects the one that is ready first. */
int ocvar_hip_ready(OcvarHip* ctx);
/* How many marker records per frame a batch brings to the host (1 .. OCVAR_MAX_MARKERS, the default): the copy-out of a
 * 2048-frame batch is 24 MB at 64 records per frame, 3 MB at 8.  ocvar_hip_collect returns at most this many per frame;
 * the counts are always the frames' full counts. */
int ocvar_hip_set_result_limit(OcvarHip* ctx, int max_per_frame);

/* Launch parameters that change how the work is cut, never what comes out (every value gives bit-identical results; the
 * defaults are the measured optimum for the batch size).  value 0 restores the default.  The product library reads no tuning
 * from the environment; only builds made with -DOCVAR_PROF (`make prof`) also honour OCVAR_* variables of the same names. */
enum { OCVAR_TUNE_CROP_PHASES = 1, /* crop-pass tier 2 in 1 launch or 2 (exact pruning behind the crop's best quad) */
       OCVAR_TUNE_MID_STEPS = 2,   /* step budget of follower tier 2 before a border goes to the wave tier (>= 32) */
       OCVAR_TUNE_MID_BLOCKS = 3, OCVAR_TUNE_LONG_BLOCKS = 4, OCVAR_TUNE_SHORT_BLOCKS = 5, /* grids of tiers 2, 3, 1 */
       OCVAR_TUNE_MIN_UNITS = 6,   /* binarise work units per launch below which row chunks are not made taller */
       OCVAR_TUNE_GATE_MODE = 8,   /* which of a context's two binarise kernels wait at its gate: 0 both (default), 1 the frames
                                    * kernel only, 2 the crops kernel only */
       OCVAR_TUNE_HP_MASK = 7 };   /* kernels launched on the context's high-priority stream, one bit per launch: 1 tier 1
                                    * (frames), 2 tier 2, 4 tier 3, 8 order/crops, 16 tier 1 (crops), 32 tier 2, 64 tier 3,
                                    * 128 decode, 256 dedupe+pose; ocvar_hip_set_tuning(ctx, OCVAR_TUNE_HP_MASK, m) sets mask m
                                    * (0 = none), there is no "back to default" for this knob short of a new context */
int ocvar_hip_set_tuning(OcvarHip* ctx, int knob, int value);
/* How the library was built: "... product(...)" or "... OCVAR_PROF(...)" -- bench.py prints it with its number. */
const char* ocvar_hip_build_info(void);

/* The context's own HIP stream (hipStream_t; what ocvar_hip_enqueue uses when its `stream` argument is NULL), for callers that
 * order their own work or events against a batch. */
void* ocvar_hip_stream(const OcvarHip* ctx);

/* After ocvar_hip_enqueue: stream-ordered device-to-device copy of the batch's results into caller-owned device
 * buffers, d_markers [n_frames][OCVAR_MAX_MARKERS] and d_counts [n_frames] -- for callers that gather results
 * across GPUs (RCCL) before any host copy.  Does not wait; ocvar_hip_collect must still be called. */
int ocvar_hip_results_to_device(OcvarHip* ctx, OcvarMarker* d_markers, int* d_counts, void* stream);
/* Same with a narrower block: d_markers [n_frames][max_per_frame] (1 <= max_per_frame <= OCVAR_MAX_MARKERS) holds the first
 * max_per_frame records of every frame; d_counts still carries the frames' full counts, so a frame with more markers than
 * the block keeps is visible to the receiver.  A gather of 8 records per frame moves 1/8 of the bytes over xGMI. */
int ocvar_hip_results_to_device_ex(OcvarHip* ctx, OcvarMarker* d_markers, int* d_counts, int max_per_frame, void* stream);

/* Several contexts on one GPU as one detector: n_contexts contexts of chunk_frames frames each (own streams, one shared gate
 * of gate_width binarise kernels, 0: none) detect a device-resident array of any number of frames chunk by chunk, every context
 * with one chunk in flight (csrc/pipe.hip: the schedule of bench.py behind the C ABI -- four or five contexts, gate 2: ~1.9x the
 * frames/s of one context).  Stateless; results in frame order like ocvar_hip_detect_device.  No counterpart in the reference
 * (one frame per call); the many-frames form of its per-frame loop, samples/ARTest.cpp:43-82. */
typedef struct OcvarPipe OcvarPipe;
int ocvar_hip_pipe_create(OcvarPipe** pipe, int device, int max_width, int max_height, int chunk_frames, int n_contexts, int gate_width);
void ocvar_hip_pipe_destroy(OcvarPipe* pipe);
const char* ocvar_hip_pipe_last_error(const OcvarPipe* pipe);
int ocvar_hip_pipe_set_templates(OcvarPipe* pipe, const OcvarTemplate* templates, int n);
int ocvar_hip_pipe_set_camera(OcvarPipe* pipe, const OcvarCamera* camera);
int ocvar_hip_pipe_detect_device(OcvarPipe* pipe, uint8_t* d_bgr, int width, int height, int row_stride, size_t frame_stride,
                                 long long n_frames, int grey_in_place, OcvarMarker* markers, int* counts, int max_per_frame);
/* Stateful form (the reference's behaviour over video: `markers` persists across calls, opencvar.cpp:635-668): one call = one
 * time step of n_streams video streams, frame s of d_bgr being stream s's current frame.  Every stream's markers of the
 * previous step are its tracking input; they are kept by the pipe in device memory between calls (never copied to the host
 * and back).  reset != 0 (or a different n_streams than the last call): the streams start with no markers, i.e. a stateless
 * first step.  Streams are independent, so chunks of streams go round the contexts as chunks of frames do above; results in
 * stream order. */
int ocvar_hip_pipe_track_device(OcvarPipe* pipe, uint8_t* d_bgr, int width, int height, int row_stride, size_t frame_stride,
                                long long n_streams, int grey_in_place, int reset, OcvarMarker* markers, int* counts, int max_per_frame);

/* Streaming form for a caller with an endless supply of frames (the many-cameras form of the reference's per-frame loop,
 * samples/ARTest.cpp:43-82): the pipeline of contexts is kept full instead of being filled and drained per call.  submit hands
 * a chunk of n_frames <= chunk_frames device-resident frames to the next context and returns at once (OCVAR_E_BUSY when every
 * context already has a chunk in flight); collect waits for the OLDEST chunk in flight, writes its markers [n][max_per_frame]
 * and counts [n], stores the tag it was submitted under and returns its frame count n (0: nothing in flight, < 0: error).
 * Stateless.  set_result_limit (only while nothing is in flight): how many marker records per frame the chunks bring to the
 * host, as ocvar_hip_set_result_limit. */
enum { OCVAR_E_BUSY = -6 };
int ocvar_hip_pipe_submit(OcvarPipe* pipe, uint8_t* d_bgr, int width, int height, int row_stride, size_t frame_stride, int n_frames,
                          int grey_in_place, long long tag);
int ocvar_hip_pipe_collect(OcvarPipe* pipe, long long* tag, OcvarMarker* markers, int* counts, int max_per_frame);
int ocvar_hip_pipe_in_flight(const OcvarPipe* pipe);
int ocvar_hip_pipe_set_result_limit(OcvarPipe* pipe, int max_per_frame);

/* Same, frames in host memory (copied over PCIe first; greyed frames are copied back when requested). */
int ocvar_hip_detect_host(OcvarHip* ctx, uint8_t* h_bgr, int width, int height, int row_stride, size_t frame_stride,
                          int n_frames, int grey_in_place, const OcvarMarker* prev, const int* prev_counts,
                          OcvarMarker* markers, int* counts, int max_per_frame);

/* cvarFindSquares on one 8-bit single-channel host image (the reference runs it on grey 3-channel images
 * whose channels are equal).  quads: up to max_quads x 8 ints in the reference's sequence order. */
int ocvar_hip_find_squares(OcvarHip* ctx, const uint8_t* h_gray, int width, int height, int row_stride, int* quads,
                           int max_quads, int* n_quads);

/* Parity hooks on the state left by the last detect/enqueue+collect call. */
int ocvar_hip_debug_gray(OcvarHip* ctx, int frame, uint8_t* h_gray /* width*height */);
int ocvar_hip_debug_binary(OcvarHip* ctx, int frame, uint8_t* h_bin /* (w&~1)*(h&~1), values 0/255 */);
/* the frame pass's 8-neighbour masks, untiled: bit s of pixel (x,y) = the neighbour in direction s (0..7 = E,NE,N,NW,W,SW,S,SE) is set */
int ocvar_hip_debug_masks(OcvarHip* ctx, int frame, uint8_t* h_masks /* (w&~1)*(h&~1) */);
int ocvar_hip_debug_frame_quads(OcvarHip* ctx, int frame, int* quads /* OCVAR_MAX_QUADS*8 */, int* n_quads);
int ocvar_hip_debug_candidates(OcvarHip* ctx, int frame, OcvarCandidate* cands, int max_cands, int* n_cands);

/* Measurement aid: runs a dword-per-lane copy of `bytes` bytes (the binarise kernel's access widths) so that a
 * rocprofv3 PMC pass can calibrate FETCH_SIZE / WRITE_SIZE for that pattern (tools/collect_traffic.py). */
int ocvar_hip_debug_calibrate(OcvarHip* ctx, size_t bytes);

/* Per-kernel device time of the last enqueue in milliseconds (HIP events on the launch stream), 12 entries:
 * [0] binarise(frames) [1..3] follower tiers 1,2,3 (frames) [4] order/crops [5] binarise(crops)
 * [6..8] follower tiers 1,2,3 (crops) [9] decode [10] dedupe+pose [11] whole batch.  Returns the number written. */
int ocvar_hip_stage_ms(OcvarHip* ctx, float* ms, int n);
/* The same 13 stage boundaries ([0] start of binarise(frames) ... [12] end of the copy-out) as milliseconds after ref_event, a
 * hipEvent_t the caller recorded (with timing) on this device before enqueueing: stamps of different contexts are on one clock,
 * so a caller with several contexts in flight can tell how launches of one kernel overlapped.  Returns the number written. */
int ocvar_hip_stage_stamps(OcvarHip* ctx, void* ref_event, float* ms, int n);
/* Work counters of the last batch: [0] frame start candidates [1] crop ROIs [2] crop tiles
 * [3] crop start candidates [4] sum of crop areas (pixels) [5] point-pool ints used [6],[7] starts handed to follower
 * tier 2 (frames, crops) [8],[9] borders handed to tier 3 (frames, crops) [10]..[41] profiling slots of builds made with
 * -DOCVAR_PROF (tools/prof_tier2.py), zero in the product build. */
int ocvar_hip_counters(OcvarHip* ctx, long long* out, int n);

#ifdef __cplusplus
}
#endif
#endif
