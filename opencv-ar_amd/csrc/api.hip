// api.hip -- the thin C ABI of include/ocvar_hip.h: context, device workspace, launch sequence, result copy-out.
//
// One batch = 12 kernel launches on one HIP stream, no host round trip in between (work counts stay in
// device memory and the second-pass kernels are launched with fixed grids that read them):
//   binarise(frames) -> follower tiers 1,2,3 (frames) -> order+crops -> binarise(crops) -> follower tiers 1, 2 (two phases:
//   follow.hip), 3 (crops) -> decode -> finalise
// There is deliberately no CPU path here: if the device or the code object is missing, create() fails.
#include "kernels.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

using namespace ocvar;

static_assert(sizeof(TemplateRec) == sizeof(OcvarTemplate) && sizeof(TemplateRec) == 48, "CvarTemplate layout");
static_assert(sizeof(CameraRec) == sizeof(OcvarCamera) && sizeof(CameraRec) == 248, "CvarCamera layout");
static_assert(sizeof(MarkerRec) == sizeof(OcvarMarker) && sizeof(MarkerRec) == 184, "CvarMarker layout");

// At most `width` binarise kernels of the contexts that share the gate run at once: launch n waits (on its stream) for the
// event recorded behind launch n - width.  Host side only keeps the ring of events.
struct OcvarGate {
    int device = 0;
    int width = 2;
    unsigned long long issued = 0;
    std::vector<hipEvent_t> ring;   // far more slots than launches can be in flight (contexts x 2)
};

struct OcvarHip {
    int device = 0;
    OcvarGate* gate = nullptr;
    int result_limit = OCVAR_MAX_MARKERS;   // marker records per frame copied to the host (ocvar_hip_set_result_limit)
    int tune[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // ocvar_hip_set_tuning: 0 = default
    Workspace ws{};
    hipStream_t stream = nullptr;
    hipStream_t hp_stream = nullptr;   // high-priority stream for the kernels OCVAR_TUNE_HP_MASK names (created on first use)
    hipStream_t last_stream = nullptr;
    hipEvent_t ev[13]{};   // 12 intervals: see ocvar_hip_stage_ms
    std::vector<void*> allocs;
    uint8_t* d_frames = nullptr;  // staging for the host-buffer entry points
    size_t d_frames_bytes = 0;
    uint8_t* h_stage[2] = {nullptr, nullptr};   // page-locked staging of the host entry points (double buffer)
    size_t h_stage_bytes = 0;                   // size of each
    uint8_t* h_grey[2] = {nullptr, nullptr};    // page-locked staging of the in-place grey on its way back (double buffer)
    size_t h_grey_bytes = 0;
    hipStream_t h2d_stream = nullptr, d2h_stream = nullptr;   // host transport of ocvar_hip_detect_host (created on first use)
    std::vector<hipEvent_t> h2d_done;
    hipEvent_t computed = nullptr;
    MarkerRec* h_markers = nullptr;  // pinned
    MarkerRec* h_prev = nullptr;     // pinned: the caller's previous markers on their way to the device
    int* h_prev_counts = nullptr;    // pinned
    int* h_counts = nullptr;         // pinned
    int* h_counters = nullptr;       // pinned
    bool pending = false;
    bool have_templates = false, have_camera = false;
    int capacity_flags = 0;   // flag word of the last batch that failed with OCVAR_E_CAPACITY
    std::string err;
};

#define HIP_TRY(ctx, call)                                                                              \
    do {                                                                                                \
        hipError_t e_ = (call);                                                                         \
        if (e_ != hipSuccess) {                                                                         \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                            \
            return OCVAR_E_HIP;                                                                         \
        }                                                                                               \
    } while (0)

template <typename T>
static int dev_alloc(OcvarHip* c, T** p, size_t n) {
    void* q = nullptr;
    HIP_TRY(c, hipMalloc(&q, n * sizeof(T) + 256));
    c->allocs.push_back(q);
    *p = static_cast<T*>(q);
    return OCVAR_OK;
}

extern "C" int ocvar_hip_create(OcvarHip** out, int device, int max_width, int max_height, int max_batch) {
    return ocvar_hip_create_ex(out, device, max_width, max_height, max_batch, OCVAR_MAX_QUADS);
}

extern "C" int ocvar_hip_create_ex(OcvarHip** out, int device, int max_width, int max_height, int max_batch, int max_quads) {
    // (corner points travel packed as x | y << 16 through the follower tiers: coordinates stay below 2^15)
    if (!out || max_width < 16 || max_height < 16 || max_width > 32767 || max_height > 32767 || max_batch < 1 || max_quads < 1 ||
        max_quads > OCVAR_MAX_QUADS_EX)
        return OCVAR_E_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return OCVAR_E_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return OCVAR_E_NO_DEVICE;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        std::fprintf(stderr, "ocvar_hip: device %d is %s; this library carries gfx950 code only\n", device, prop.gcnArchName);
        return OCVAR_E_NO_DEVICE;
    }
    OcvarHip* c = new (std::nothrow) OcvarHip();
    if (!c) return OCVAR_E_HIP;
    *out = c;  // returned even on failure below so the caller can read the error text, then destroy
    c->device = device;
    HIP_TRY(c, hipSetDevice(device));
    HIP_TRY(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    for (auto& e : c->ev) HIP_TRY(c, hipEventCreate(&e));
    Workspace& w = c->ws;
    w.max_w = max_width;
    w.max_h = max_height;
    w.max_batch = max_batch;
    w.maxq = max_quads;
    w.maxc = max_quads;   // set with the templates (maxq * n_templates, at most what 64 KB of LDS hold)
    const size_t B = (size_t)max_batch, WH = (size_t)max_width * max_height;
    size_t per_frame_cands = WH / 16 < 16384 ? 16384 : WH / 16;
    w.cap_frame_cands = (int)std::min<size_t>(B * per_frame_cands, (size_t)1 << 30);
    w.cap_crop_cands = w.cap_frame_cands;
    w.cap_crop_rois = (int)(B * max_quads);
    w.cap_crop_tiles = (int)std::min<size_t>(B * 4096, (size_t)1 << 30);
    w.cap_crop_quads = (int)(B * max_quads * 4);
    // only tier-2 borders with more corner points than a lane slab holds land here; the fixed part lets a small context take
    // a pathological frame (full-frame noise: thousands of long ragged borders)
    w.cap_pool_ints = (long long)B * (1 << 18) + (1 << 24);
    w.cap_crop_pixels = (long long)(2 * B * (size_t)(max_width + 16) * (max_height + 8));
    int rc;
    if ((rc = dev_alloc(c, &w.gray, B * (size_t)gray_plane_bytes(max_width, max_height)))) return rc;   // (panels: hd.h::gray_col)
    if ((rc = dev_alloc(c, &w.nbr_frame, B * (size_t)(max_width + 16) * (max_height + 8)))) return rc;
    if ((rc = dev_alloc(c, &w.nbr_crop, (size_t)w.cap_crop_pixels))) return rc;
    if ((rc = dev_alloc(c, &w.cands_frame, (size_t)w.cap_frame_cands))) return rc;
    if ((rc = dev_alloc(c, &w.cands_crop, (size_t)w.cap_crop_cands))) return rc;
    if ((rc = dev_alloc(c, &w.pool, (size_t)w.cap_pool_ints))) return rc;
    // follower grids (and their slabs) scale with the batch: a one-frame context (the reference's per-frame call) does not
    // need -- or pay for -- the 512 + 1024 workgroups that keep a 2048-frame batch busy
    w.max_mid_blocks = (int)std::min<size_t>(MID_BLOCKS_MAX, std::max<size_t>(32, B));
    w.max_long_blocks = (int)std::min<size_t>(LONG_BLOCKS_MAX, std::max<size_t>(128, B * 8));
    if ((rc = dev_alloc(c, &w.slab, (size_t)w.max_mid_blocks * 256 * SLAB_STRIDE))) return rc;
    if ((rc = dev_alloc(c, &w.slab3, (size_t)w.max_long_blocks * 4 * SLAB3_STRIDE))) return rc;
    w.cap_long = (int)std::min<size_t>(std::max<size_t>(B * 4096, (size_t)1 << 18), (size_t)1 << 28);   // survivors of tier 1 / tier 2: a noise frame has ~10^4
    if ((rc = dev_alloc(c, &w.mid_frame, (size_t)w.cap_long))) return rc;
    if ((rc = dev_alloc(c, &w.mid_crop, (size_t)w.cap_long))) return rc;
    if ((rc = dev_alloc(c, &w.mid_first_crop, (size_t)w.cap_long))) return rc;
    if ((rc = dev_alloc(c, &w.long_frame, (size_t)w.cap_long))) return rc;
    if ((rc = dev_alloc(c, &w.long_crop, (size_t)w.cap_long))) return rc;
    if ((rc = dev_alloc(c, &w.quads_frame, B * max_quads))) return rc;
    if ((rc = dev_alloc(c, &w.n_quads_frame, B))) return rc;
    if ((rc = dev_alloc(c, &w.squares, B * max_quads * 8))) return rc;
    if ((rc = dev_alloc(c, &w.n_squares, B))) return rc;
    if ((rc = dev_alloc(c, &w.crop_of, B * max_quads))) return rc;
    if ((rc = dev_alloc(c, &w.rois_crop, (size_t)w.cap_crop_rois))) return rc;
    if ((rc = dev_alloc(c, &w.tiles_crop, (size_t)w.cap_crop_tiles))) return rc;
    if ((rc = dev_alloc(c, &w.quads_crop, (size_t)w.cap_crop_quads))) return rc;
    if ((rc = dev_alloc(c, &w.best_crop, (size_t)w.cap_crop_rois))) return rc;
    if ((rc = dev_alloc(c, &w.crop_min_rest, (size_t)w.cap_crop_rois))) return rc;
    if ((rc = dev_alloc(c, &w.ring_frame, B))) return rc;
    if ((rc = dev_alloc(c, &w.ring_crop, (size_t)w.cap_crop_rois))) return rc;
    if ((rc = dev_alloc(c, &w.cand_recs, B * max_quads * MAXT))) return rc;
    if ((rc = dev_alloc(c, &w.prev, B * MAXM))) return rc;
    if ((rc = dev_alloc(c, &w.n_prev, B))) return rc;
    if ((rc = dev_alloc(c, &w.reserve, B * MAXM))) return rc;
    if ((rc = dev_alloc(c, &w.n_reserve, B))) return rc;
    if ((rc = dev_alloc(c, &w.markers, B * MAXM))) return rc;
    if ((rc = dev_alloc(c, &w.pose_jobs, B * MAXM))) return rc;
    if ((rc = dev_alloc(c, &w.n_markers, B))) return rc;
    if ((rc = dev_alloc(c, &w.templates, (size_t)MAXT))) return rc;
    if ((rc = dev_alloc(c, &w.camera, (size_t)1))) return rc;
    if ((rc = dev_alloc(c, &w.counters, (size_t)CNT_COUNT))) return rc;

    w.crop_pixels = reinterpret_cast<unsigned long long*>(w.counters + CNT_CROP_PIXELS);
    HIP_TRY(c, hipMemset(w.n_prev, 0, B * sizeof(int)));
    HIP_TRY(c, hipMemset(w.counters, 0, CNT_COUNT * sizeof(int)));
    HIP_TRY(c, hipHostMalloc((void**)&c->h_markers, B * MAXM * sizeof(MarkerRec)));
    HIP_TRY(c, hipHostMalloc((void**)&c->h_counts, B * sizeof(int)));
    HIP_TRY(c, hipHostMalloc((void**)&c->h_prev, B * MAXM * sizeof(MarkerRec)));
    HIP_TRY(c, hipHostMalloc((void**)&c->h_prev_counts, B * sizeof(int)));
    HIP_TRY(c, hipHostMalloc((void**)&c->h_counters, CNT_COUNT * sizeof(int)));
    return OCVAR_OK;
}

extern "C" void ocvar_hip_destroy(OcvarHip* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (void* p : c->allocs) (void)hipFree(p);
    if (c->d_frames) (void)hipFree(c->d_frames);
    for (auto p : c->h_stage)
        if (p) (void)hipHostFree(p);
    for (auto p : c->h_grey)
        if (p) (void)hipHostFree(p);
    if (c->h_markers) (void)hipHostFree(c->h_markers);
    if (c->h_counts) (void)hipHostFree(c->h_counts);
    if (c->h_prev) (void)hipHostFree(c->h_prev);
    if (c->h_prev_counts) (void)hipHostFree(c->h_prev_counts);
    if (c->h_counters) (void)hipHostFree(c->h_counters);
    for (auto& e : c->ev)
        if (e) (void)hipEventDestroy(e);
    for (auto& e : c->h2d_done) (void)hipEventDestroy(e);
    if (c->computed) (void)hipEventDestroy(c->computed);
    if (c->h2d_stream) (void)hipStreamDestroy(c->h2d_stream);
    if (c->d2h_stream) (void)hipStreamDestroy(c->d2h_stream);
    if (c->hp_stream) (void)hipStreamDestroy(c->hp_stream);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int ocvar_hip_gate_create(OcvarGate** out, int device, int width) {
    if (!out || width < 1 || width > 64) return OCVAR_E_ARG;
    *out = nullptr;
    if (hipSetDevice(device) != hipSuccess) return OCVAR_E_NO_DEVICE;
    OcvarGate* g = new (std::nothrow) OcvarGate();
    if (!g) return OCVAR_E_HIP;
    g->device = device;
    g->width = width;
    g->ring.resize(256);
    for (auto& e : g->ring)
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
            for (auto& d : g->ring)
                if (d) (void)hipEventDestroy(d);
            delete g;
            return OCVAR_E_HIP;
        }
    *out = g;
    return OCVAR_OK;
}

extern "C" void ocvar_hip_gate_destroy(OcvarGate* g) {
    if (!g) return;
    (void)hipSetDevice(g->device);
    for (auto& e : g->ring)
        if (e) (void)hipEventDestroy(e);
    delete g;
}

extern "C" int ocvar_hip_set_gate(OcvarHip* c, OcvarGate* g) {
    if (!c || (g && g->device != c->device)) return OCVAR_E_ARG;
    c->gate = g;
    return OCVAR_OK;
}

// before / after a gated launch on stream s
static hipError_t gate_enter(OcvarGate* g, hipStream_t s) {
    if (!g || g->issued < (unsigned long long)g->width) return hipSuccess;
    return hipStreamWaitEvent(s, g->ring[(g->issued - g->width) % g->ring.size()], 0);
}
static hipError_t gate_leave(OcvarGate* g, hipStream_t s) {
    if (!g) return hipSuccess;
    const hipError_t e = hipEventRecord(g->ring[g->issued % g->ring.size()], s);
    g->issued++;
    return e;
}

extern "C" void* ocvar_hip_stream(const OcvarHip* c) { return c ? (void*)c->stream : nullptr; }

extern "C" const char* ocvar_hip_last_error(const OcvarHip* c) { return c ? c->err.c_str() : "null context"; }
extern "C" int ocvar_hip_capacity_flags(const OcvarHip* c) { return c ? c->capacity_flags : 0; }

extern "C" int ocvar_hip_set_templates(OcvarHip* c, const OcvarTemplate* t, int n) {
    if (!c || !t || n < 1 || n > MAXT) return OCVAR_E_ARG;
    for (int i = 0; i < n; i++)
        if (t[i].width < 1 || t[i].height < 1 || t[i].width * t[i].height > 64) return OCVAR_E_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpy(c->ws.templates, t, n * sizeof(OcvarTemplate), hipMemcpyHostToDevice));
    c->ws.n_templates = n;
    c->ws.maxc = std::min(c->ws.maxq * n, 7000);   // 9 bytes of LDS per candidate, 64 KB per workgroup
    c->have_templates = true;
    return OCVAR_OK;
}

extern "C" int ocvar_hip_set_camera(OcvarHip* c, const OcvarCamera* cam) {
    if (!c || !cam) return OCVAR_E_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpy(c->ws.camera, cam, sizeof(OcvarCamera), hipMemcpyHostToDevice));
    c->have_camera = true;
    return OCVAR_OK;
}

// A result-invariant launch parameter: the context's own setting (ocvar_hip_set_tuning), else the default.  Profiling builds
// (-DOCVAR_PROF) also listen to the environment variable of the same purpose; the product library reads no tuning from the
// environment, so a benchmark number cannot depend on the caller's shell.
constexpr int HP_MASK_DEFAULT = 0;

static long long tuned(const OcvarHip* c, int knob, const char* env_name, long long dflt) {
    if (knob > 0 && knob < 12 && c->tune[knob] > 0) return knob == OCVAR_TUNE_HP_MASK ? c->tune[knob] - 1 : c->tune[knob];
#ifdef OCVAR_PROF
    if (const char* e = std::getenv(env_name)) return std::atoll(e);
#else
    (void)env_name;
#endif
    return dflt;
}

extern "C" int ocvar_hip_set_tuning(OcvarHip* c, int knob, int value) {
    if (!c || knob < 1 || knob > 8 || value < 0 || c->pending) return OCVAR_E_ARG;
    c->tune[knob] = knob == OCVAR_TUNE_HP_MASK ? value + 1 : value;   // (0 is a meaningful mask: stored off by one, 0 = default)
    return OCVAR_OK;
}

extern "C" const char* ocvar_hip_build_info(void) {
    return "libocvar_hip gfx950"
#ifdef OCVAR_NBR_TILED
           " OCVAR_NBR_TILED"
#endif
#ifdef OCVAR_PROF
           " OCVAR_PROF(env tuning + knock-out switches + cycle counters: NOT a product build)"
#else
           " product(no environment tuning, no knock-out switches; OCVAR_TRACE_LAUNCHES only)"
#endif
        ;
}

// OCVAR_TRACE_LAUNCHES=1: wait after every launch and name it on stderr (locating a faulting or hanging kernel)
static bool trace_launches() {
    static const bool on = std::getenv("OCVAR_TRACE_LAUNCHES") != nullptr;
    return on;
}
#define TRACE_LAUNCH(name, st)                                                             \
    do {                                                                                 \
        if (trace_launches()) {                                                          \
            std::fprintf(stderr, "ocvar: %s ...", name);                                 \
            std::fflush(stderr);                                                         \
            hipError_t e_ = hipStreamSynchronize(st);                                    \
            std::fprintf(stderr, " %s\n", e_ == hipSuccess ? "ok" : hipGetErrorString(e_)); \
        }                                                                                \
    } while (0)

static int enqueue_impl(OcvarHip* c, uint8_t* d_bgr, int width, int height, int row_stride, size_t frame_stride, int n_frames,
                        int grey_in_place, const OcvarMarker* prev, const int* prev_counts, hipStream_t s, int stages,
                        bool prev_on_device = false) {
    Workspace& w = c->ws;
    if (!d_bgr || width < 16 || height < 16 || width > w.max_w || height > w.max_h || n_frames < 1 || n_frames > w.max_batch ||
        (size_t)width * height > (size_t)w.max_w * w.max_h || row_stride < 3 * width)
        return OCVAR_E_ARG;
    if (c->pending) {
        c->err = "the previous batch of this context has not been collected";
        return OCVAR_E_ARG;
    }
    if (stages > 2 && (!c->have_templates || !c->have_camera)) {
        c->err = "templates and camera must be set before detection";
        return OCVAR_E_ARG;
    }
    HIP_TRY(c, hipSetDevice(c->device));
    w.W = width;
    w.H = height;
    w.sw = width & ~1;
    w.sh = height & ~1;
    w.ns = (w.sw + 15) & ~15;
    w.n_frames = n_frames;
    // Tier 2's step budget: a batch of a few frames has too few borders to fill the GPU with one-lane walks, and its
    // duration is then the longest walk (~800 one-microsecond steps around a crop) -- such batches hand everything longer
    // than 128 steps to the wave tier, which crosses straight runs 64 pixels at a time (1080p, one frame per call: 3.2 ->
    // 2.5 ms).  Large batches keep the long budget: there the one-lane walks are what fills the machine.
    w.mid_steps = tuned(c, OCVAR_TUNE_MID_STEPS, "OCVAR_MID_STEPS", n_frames <= 8 ? 128 : MID_STEPS);
    if (w.mid_steps < 32) w.mid_steps = 32;
    // Crop tier 2 in two phases saves half of its steps but chains two launches: throughput for batches (+1..2 %), 0.1 ms of
    // latency for a one-frame call -- which therefore keeps the single launch.
    w.crop_phases = tuned(c, OCVAR_TUNE_CROP_PHASES, "OCVAR_CROP_PHASES", n_frames <= 8 ? 1 : 2) == 1 ? 1 : 2;
    // Tier 2's grid: alone on the GPU a context wants every lane it can get (its duration is a chain of dependent loads; 1024
    // workgroups: 2.8 ms for the crop pass of 2048 frames, 256: 4.6 ms).  A context that shares the GPU with others (it has a
    // gate) takes a quarter: tier 2's 110-register waves then leave room for the other contexts' binarise waves (4 contexts:
    // 171 -> 176 k frames/s; 128 or 512 workgroups: 171 / 173 k).
    w.mid_blocks = tuned(c, OCVAR_TUNE_MID_BLOCKS, "OCVAR_MID_BLOCKS", c->gate ? std::max(32, w.max_mid_blocks / 4) : w.max_mid_blocks);
    if (w.mid_blocks < 1 || w.mid_blocks > w.max_mid_blocks) w.mid_blocks = w.max_mid_blocks;
    w.long_blocks = tuned(c, OCVAR_TUNE_LONG_BLOCKS, "OCVAR_LONG_BLOCKS", w.max_long_blocks);
    if (w.long_blocks < 1 || w.long_blocks > w.max_long_blocks) w.long_blocks = w.max_long_blocks;
    // the fixed grids of the work-queue kernels shrink with the batch: a one-frame call does not launch (and wait out) the
    // thousands of workgroups that keep a 2048-frame batch busy
    w.short_blocks = tuned(c, OCVAR_TUNE_SHORT_BLOCKS, "OCVAR_SHORT_BLOCKS", n_frames >= 128 ? 1024 : (n_frames * 8 < 16 ? 16 : n_frames * 8));
    if (w.short_blocks < 1 || w.short_blocks > 65535) w.short_blocks = 1024;
    w.crop_blocks = n_frames >= 128 ? 2048 : (n_frames * 16 < 32 ? 32 : n_frames * 16);
    w.frame_strips = (w.sw + MARCH_STRIP - 1) / MARCH_STRIP;
    {   // rows per binarise work unit: even, chunks of equal size.  Every chunk re-reads ~12 halo rows, so chunks are as
        // tall as the batch allows while the launch still has >= 64K waves (env OCVAR_MIN_UNITS); never < ~128 rows.  (Measured: choosing the
        // count to fill whole "rounds" of resident waves is no better -- the kernel is issue-bound, not round-bound --
        // and three 360-row chunks per 1080p frame were 20 % slower than eight 136-row ones at 256 frames.)
        int chunks = (w.sh + 64) / 128;
        if (chunks < 1) chunks = 1;
        // (a context that shares the GPU -- it has a gate -- takes the tallest chunks that still give 16 K units: at 2048 frames one
        // 1080-row chunk per strip.  Alone that launch is 8 % slower than four 272-row chunks, 5.9 against 5.4 ms -- fewer, longer
        // waves hide less --, with four contexts in flight it is the faster one: 192 against 185 k frames/s, fewer halo rows and
        // fewer workgroup turnovers for the other contexts' kernels to queue behind)
        const long long min_units = tuned(c, OCVAR_TUNE_MIN_UNITS, "OCVAR_MIN_UNITS", c->gate ? 16384 : 65536);
        while (chunks > 1 && (long long)w.frame_strips * (chunks / 2) * n_frames >= min_units) chunks /= 2;
        int rows = (w.sh + chunks - 1) / chunks;
        rows = (rows + 7) & ~7;   // whole mask tiles (8 rows) per work unit: binarise.hip writes the mask plane tile by tile
        w.frame_chunk_rows = rows;
        w.frame_chunks = (w.sh + rows - 1) / rows;
    }
    HIP_TRY(c, hipMemsetAsync(w.counters, 0, CNT_COUNT * sizeof(int), s));
    HIP_TRY(c, hipMemsetAsync(w.n_quads_frame, 0, n_frames * sizeof(int), s));
    if (prev && prev_counts && prev_on_device) {
        // the previous step's markers never left the device (ocvar_hip_enqueue_tracked: streams of a tracker)
        HIP_TRY(c, hipMemcpyAsync(w.prev, prev, (size_t)n_frames * MAXM * sizeof(MarkerRec), hipMemcpyDeviceToDevice, s));
        HIP_TRY(c, hipMemcpyAsync(w.n_prev, prev_counts, n_frames * sizeof(int), hipMemcpyDeviceToDevice, s));
    } else if (prev && prev_counts) {
        // through the context's page-locked buffers: the device never touches the caller's (pageable, possibly tiny) arrays.
        // (A batch is collected before the next one is enqueued on a context, so the buffers are free again by then.)
        std::memcpy(c->h_prev, prev, (size_t)n_frames * MAXM * sizeof(MarkerRec));
        std::memcpy(c->h_prev_counts, prev_counts, n_frames * sizeof(int));
        HIP_TRY(c, hipMemcpyAsync(w.prev, c->h_prev, (size_t)n_frames * MAXM * sizeof(MarkerRec), hipMemcpyHostToDevice, s));
        HIP_TRY(c, hipMemcpyAsync(w.n_prev, c->h_prev_counts, n_frames * sizeof(int), hipMemcpyHostToDevice, s));
    } else {
        HIP_TRY(c, hipMemsetAsync(w.n_prev, 0, n_frames * sizeof(int), s));
    }
    // Which kernels run on the context's high-priority stream (OCVAR_TUNE_HP_MASK, one bit per launch after the first
    // binarise kernel: 1 tier 1 (frames), 2 tier 2, 4 tier 3, 8 order/crops, 16 tier 1 (crops), 32 tier 2, 64 tier 3, 128 decode,
    // 256 dedupe+pose).  With several contexts in flight a binarise kernel of another context has tens of thousands of
    // 80-register workgroups queued, and every slot a finished one frees is refilled from that queue at once: a kernel with a
    // larger footprint -- decode and dedupe+pose: 256 / 64 threads at 128 registers -- finds room only by accident and takes 6 ms
    // in-region for 0.3 ms of work, at the end of its context's chain.  On a high-priority queue its workgroups are placed
    // first.  (All followers on the high-priority stream -- round 2's OCVAR_SPLIT_STREAMS -- halved their in-region durations and
    // lengthened binarise's by as much; the mask chooses kernel by kernel.)  The events that time the stages also order the
    // streams.
    const int hp_mask = (int)tuned(c, OCVAR_TUNE_HP_MASK, "OCVAR_HP_MASK", HP_MASK_DEFAULT) & 0x1ff;
    if (hp_mask && !c->hp_stream) {
        int lo = 0, hi = 0;
        HIP_TRY(c, hipDeviceGetStreamPriorityRange(&lo, &hi));   // hi: numerically lowest = greatest priority
        HIP_TRY(c, hipStreamCreateWithPriority(&c->hp_stream, hipStreamNonBlocking, hi));
    }
    hipStream_t cur = s;   // the stream the chain is on
    auto stage = [&](int k, int bit) -> hipError_t {   // timing event k at the end of the previous stage; the next one runs where its bit says
        hipStream_t to = (bit >= 0 && ((hp_mask >> bit) & 1)) ? c->hp_stream : s;
        hipError_t e = hipEventRecord(c->ev[k], cur);
        if (e == hipSuccess && to != cur) e = hipStreamWaitEvent(to, c->ev[k], 0);
        cur = to;
        return e;
    };
    const int gate_mode = (int)tuned(c, OCVAR_TUNE_GATE_MODE, "OCVAR_GATE_MODE", 0);   // which binarise kernels the gate covers: 0 both, 1 the frames kernel only, 2 the crops kernel only
    if (gate_mode != 2) HIP_TRY(c, gate_enter(c->gate, s));   // (before the first timing event: a wait at the gate is not binarise time)
    HIP_TRY(c, hipEventRecord(c->ev[0], s));
    launch_binarise_frames(w, d_bgr, row_stride, frame_stride, grey_in_place, s);
    if (gate_mode != 2) HIP_TRY(c, gate_leave(c->gate, s));
    TRACE_LAUNCH("binarise_frames", s);
    // Timing experiments (results are then incomplete or wrong): compiled into profiling builds only (-DOCVAR_PROF, `make prof`)
    //   OCVAR_ONLY_BINARISE=1        stop after the first kernel (tools/binarise_only.py)
    //   OCVAR_SKIP_CROP_KERNELS=bits knock out kernels of the crop pass: 1 binarise_crops, 2 tier 1, 4 tier 2, 8 tier 3
#ifdef OCVAR_PROF
    static const bool only_binarise = std::getenv("OCVAR_ONLY_BINARISE") != nullptr;
    static const int skip_crop = std::getenv("OCVAR_SKIP_CROP_KERNELS") ? std::atoi(std::getenv("OCVAR_SKIP_CROP_KERNELS")) : 0;
    if (only_binarise || skip_crop) {
        static bool warned = false;
        if (!warned) std::fprintf(stderr, "ocvar_hip: OCVAR_ONLY_BINARISE / OCVAR_SKIP_CROP_KERNELS set -- timing experiment, detection results are NOT valid\n");
        warned = true;
    }
#else
    constexpr bool only_binarise = false;
    constexpr int skip_crop = 0;
#endif
    if (only_binarise) stages = 0;
    if (stages > 0) {
        HIP_TRY(c, stage(1, 0));
        // (one frame per call: two more launches cost more than the walks they save -- the wave tier crosses a clean frame border 64
        // pixels at a time --: 2.37 -> 2.47 ms per 1080p call measured; batches: tier 3 on frames 0.32 -> 0.15 ms, tier 2 on crops -8 %)
        if (n_frames > 8) launch_ring_quads_frames(w, cur);
        else HIP_TRY(c, hipMemsetAsync(w.ring_frame, 0, n_frames * sizeof(int), cur));
        launch_follow_frames(w, cur);
        TRACE_LAUNCH("follow tier 1 (frames)", cur);
        HIP_TRY(c, stage(2, 1));
        launch_follow_mid_frames(w, cur);
        TRACE_LAUNCH("follow tier 2 (frames)", cur);
        HIP_TRY(c, stage(3, 2));
        launch_follow_long_frames(w, cur);
        TRACE_LAUNCH("follow tier 3 (frames)", cur);
        HIP_TRY(c, stage(4, 3));
        launch_order_and_crops(w, cur);
        TRACE_LAUNCH("order_and_crops", cur);
    } else {
        for (int k = 1; k < 5; k++) HIP_TRY(c, hipEventRecord(c->ev[k], s));
    }
    if (stages > 2) {
        // (the gate wait is enqueued on s before s is made to wait for the order kernel: a wait here is booked under the
        // order_crops interval)
        if (gate_mode != 1) HIP_TRY(c, gate_enter(c->gate, s));
        HIP_TRY(c, stage(5, -1));
        if (!(skip_crop & 1)) launch_binarise_crops(w, s);
        if (gate_mode != 1) HIP_TRY(c, gate_leave(c->gate, s));
        TRACE_LAUNCH("binarise_crops", s);
        HIP_TRY(c, stage(6, 4));
        if (!(skip_crop & 2) && n_frames > 8) launch_ring_quads_crops(w, cur);   // (else: order_and_crops_kernel cleared the crops' flags)
        if (!(skip_crop & 2)) launch_follow_crops(w, cur);
        TRACE_LAUNCH("follow tier 1 (crops)", cur);
        HIP_TRY(c, stage(7, 5));
        if (!(skip_crop & 4)) launch_follow_mid_crops(w, cur);
        TRACE_LAUNCH("follow tier 2 (crops)", cur);
        HIP_TRY(c, stage(8, 6));
        if (!(skip_crop & 8)) launch_follow_long_crops(w, cur);
        TRACE_LAUNCH("follow tier 3 (crops)", cur);
        HIP_TRY(c, stage(9, 7));
        launch_decode(w, cur);
        TRACE_LAUNCH("decode", cur);
        HIP_TRY(c, stage(10, 8));
        launch_finalise(w, cur);
        TRACE_LAUNCH("finalise", cur);
        HIP_TRY(c, stage(11, -1));
        HIP_TRY(c, hipMemcpyAsync(c->h_counts, w.n_markers, n_frames * sizeof(int), hipMemcpyDeviceToHost, s));
        if (c->result_limit >= MAXM)
            HIP_TRY(c, hipMemcpyAsync(c->h_markers, w.markers, (size_t)n_frames * MAXM * sizeof(MarkerRec), hipMemcpyDeviceToHost, s));
        else   // the first result_limit records of every frame, at their usual places in the host block
            HIP_TRY(c, hipMemcpy2DAsync(c->h_markers, (size_t)MAXM * sizeof(MarkerRec), w.markers, (size_t)MAXM * sizeof(MarkerRec),
                                        (size_t)c->result_limit * sizeof(MarkerRec), (size_t)n_frames, hipMemcpyDeviceToHost, s));
    } else {
        HIP_TRY(c, stage(5, -1));
        for (int k = 6; k < 12; k++) HIP_TRY(c, hipEventRecord(c->ev[k], s));
    }
    HIP_TRY(c, hipMemcpyAsync(c->h_counters, w.counters, CNT_COUNT * sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipEventRecord(c->ev[12], s));
    HIP_TRY(c, hipGetLastError());
    c->last_stream = s;
    c->pending = true;
    return OCVAR_OK;
}

static int wait_impl(OcvarHip* c) {
    if (!c->pending) {
        c->err = "nothing enqueued";
        return OCVAR_E_ARG;
    }
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->last_stream));
    c->pending = false;
    const int e = c->h_counters[CNT_ERR];
    c->capacity_flags = e;
    if (e) {
        char buf[160];
        std::snprintf(buf, sizeof buf, "device work list overflow / trace overrun, flags 0x%x (1 starts, 2 point pool, 4 quads, 8 overrun, 16 crops, 32 tiles, 64 ticket runaway, 128 markers)", e);
        c->err = buf;
        return OCVAR_E_CAPACITY;
    }
    return OCVAR_OK;
}

extern "C" int ocvar_hip_enqueue(OcvarHip* c, uint8_t* d_bgr, int width, int height, int row_stride, size_t frame_stride,
                                 int n_frames, int grey_in_place, const OcvarMarker* prev, const int* prev_counts, void* stream) {
    if (!c) return OCVAR_E_ARG;
    return enqueue_impl(c, d_bgr, width, height, row_stride, frame_stride, n_frames, grey_in_place, prev, prev_counts,
                        stream ? (hipStream_t)stream : c->stream, 3);
}

extern "C" int ocvar_hip_enqueue_tracked(OcvarHip* c, uint8_t* d_bgr, int width, int height, int row_stride, size_t frame_stride,
                                         int n_frames, int grey_in_place, const OcvarMarker* d_prev, const int* d_prev_counts, void* stream) {
    if (!c || !d_prev || !d_prev_counts) return OCVAR_E_ARG;
    return enqueue_impl(c, d_bgr, width, height, row_stride, frame_stride, n_frames, grey_in_place, d_prev, d_prev_counts,
                        stream ? (hipStream_t)stream : c->stream, 3, true);
}

extern "C" int ocvar_hip_results_to_device(OcvarHip* c, OcvarMarker* d_markers, int* d_counts, void* stream) {
    return ocvar_hip_results_to_device_ex(c, d_markers, d_counts, MAXM, stream);
}

extern "C" int ocvar_hip_results_to_device_ex(OcvarHip* c, OcvarMarker* d_markers, int* d_counts, int max_per_frame, void* stream) {
    if (!c || !d_markers || !d_counts || !c->pending || max_per_frame < 1 || max_per_frame > MAXM) return OCVAR_E_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->last_stream;
    if (s != c->last_stream) {  // order behind the batch
        HIP_TRY(c, hipStreamWaitEvent(s, c->ev[12], 0));
    }
    const int n = c->ws.n_frames;
    if (max_per_frame == MAXM)
        HIP_TRY(c, hipMemcpyAsync(d_markers, c->ws.markers, (size_t)n * MAXM * sizeof(MarkerRec), hipMemcpyDeviceToDevice, s));
    else   // the first max_per_frame records of every frame: a strided copy
        HIP_TRY(c, hipMemcpy2DAsync(d_markers, (size_t)max_per_frame * sizeof(MarkerRec), c->ws.markers, (size_t)MAXM * sizeof(MarkerRec),
                                    (size_t)max_per_frame * sizeof(MarkerRec), (size_t)n, hipMemcpyDeviceToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(d_counts, c->ws.n_markers, n * sizeof(int), hipMemcpyDeviceToDevice, s));
    return OCVAR_OK;
}

extern "C" int ocvar_hip_collect(OcvarHip* c, OcvarMarker* markers, int* counts, int max_per_frame) {
    if (!c || !counts || max_per_frame < 0 || (max_per_frame > 0 && !markers)) return OCVAR_E_ARG;
    int rc = wait_impl(c);
    if (rc) return rc;
    const int n = c->ws.n_frames;
    for (int f = 0; f < n; f++) {
        counts[f] = c->h_counts[f];
        int k = counts[f] < max_per_frame ? counts[f] : max_per_frame;
        if (k > c->result_limit) k = c->result_limit;
        if (k > 0) std::memcpy(markers + (size_t)f * max_per_frame, c->h_markers + (size_t)f * MAXM, k * sizeof(OcvarMarker));
    }
    return OCVAR_OK;
}

extern "C" int ocvar_hip_ready(OcvarHip* c) {
    if (!c || !c->pending) return OCVAR_E_ARG;
    const hipError_t e = hipEventQuery(c->ev[12]);
    if (e == hipSuccess) return 1;
    if (e == hipErrorNotReady) {
        (void)hipGetLastError();
        return 0;
    }
    c->err = std::string("hipEventQuery: ") + hipGetErrorString(e);
    return OCVAR_E_HIP;
}

extern "C" int ocvar_hip_set_result_limit(OcvarHip* c, int max_per_frame) {
    if (!c || c->pending || max_per_frame < 1 || max_per_frame > MAXM) return OCVAR_E_ARG;
    c->result_limit = max_per_frame;
    return OCVAR_OK;
}

extern "C" int ocvar_hip_detect_device(OcvarHip* c, uint8_t* d_bgr, int width, int height, int row_stride, size_t frame_stride,
                                       int n_frames, int grey_in_place, const OcvarMarker* prev, const int* prev_counts,
                                       OcvarMarker* markers, int* counts, int max_per_frame) {
    int rc = ocvar_hip_enqueue(c, d_bgr, width, height, row_stride, frame_stride, n_frames, grey_in_place, prev, prev_counts, nullptr);
    if (rc) return rc;
    return ocvar_hip_collect(c, markers, counts, max_per_frame);
}

static int reserve_staging(OcvarHip* c, size_t bytes) {
    if (bytes > c->d_frames_bytes) {
        if (c->d_frames) (void)hipFree(c->d_frames);
        c->d_frames = nullptr;
        c->d_frames_bytes = 0;
        HIP_TRY(c, hipMalloc((void**)&c->d_frames, bytes));
        c->d_frames_bytes = bytes;
    }
    return OCVAR_OK;
}

// two page-locked host buffers of at least `bytes` each (the library's own: the only host memory of a frame transfer the copy
// engines ever see, unless the caller's buffer is page-locked by the caller)
static int reserve_host_stage(OcvarHip* c, size_t bytes) {
    if (bytes > c->h_stage_bytes) {
        for (auto& p : c->h_stage) {
            if (p) (void)hipHostFree(p);
            p = nullptr;
        }
        c->h_stage_bytes = 0;
        for (auto& p : c->h_stage) HIP_TRY(c, hipHostMalloc((void**)&p, bytes));
        c->h_stage_bytes = bytes;
    }
    return OCVAR_OK;
}

static int reserve_host_grey(OcvarHip* c, size_t bytes) {
    if (bytes > c->h_grey_bytes) {
        for (auto& p : c->h_grey) {
            if (p) (void)hipHostFree(p);
            p = nullptr;
        }
        c->h_grey_bytes = 0;
        for (auto& p : c->h_grey) HIP_TRY(c, hipHostMalloc((void**)&p, bytes));
        c->h_grey_bytes = bytes;
    }
    return OCVAR_OK;
}

static int stage_frames(OcvarHip* c, const uint8_t* h, int height, int row_stride, size_t frame_stride, int n_frames) {
    const size_t bytes = (size_t)(n_frames - 1) * frame_stride + (size_t)height * row_stride;
    int rc = reserve_staging(c, bytes);
    if (rc) return rc;
    if ((rc = reserve_host_stage(c, bytes))) return rc;
    std::memcpy(c->h_stage[0], h, bytes);
    HIP_TRY(c, hipMemcpyAsync(c->d_frames, c->h_stage[0], bytes, hipMemcpyHostToDevice, c->stream));
    return OCVAR_OK;
}

// A host-to-host copy of a sub-batch (hundreds of megabytes at 1080p) by a few threads: one core copies ~10 GB/s, the PCIe
// link behind it takes 46.
static void host_copy(uint8_t* dst, const uint8_t* src, size_t n) {
    constexpr size_t PIECE = (size_t)4 << 20;
    unsigned nt = (unsigned)std::min<size_t>(8, n / PIECE);
    const unsigned hw = std::thread::hardware_concurrency();
    if (hw && nt > hw) nt = hw;
    if (nt < 2) {
        std::memcpy(dst, src, n);
        return;
    }
    std::vector<std::thread> th;
    const size_t per = ((n / nt) + 63) & ~(size_t)63;
    for (unsigned t = 1; t < nt; t++) {
        const size_t off = per * t, len = off >= n ? 0 : std::min(per, n - off);
        if (!len) continue;
        try {
            th.emplace_back([=] { std::memcpy(dst + off, src + off, len); });
        } catch (...) {   // no thread to be had: this piece is copied here (nothing is thrown across the C ABI)
            std::memcpy(dst + off, src + off, len);
        }
    }
    std::memcpy(dst, src, std::min(per, n));
    for (auto& t : th) t.join();
}

// is [p, p + bytes) host memory the CALLER has page-locked (hipHostMalloc / hipHostRegister on their side)?
static bool caller_pinned(const uint8_t* p, size_t bytes) {
    hipPointerAttribute_t a{}, b{};
    const bool ok = hipPointerGetAttributes(&a, p) == hipSuccess && a.type == hipMemoryTypeHost &&
                    hipPointerGetAttributes(&b, p + bytes - 1) == hipSuccess && b.type == hipMemoryTypeHost;
    (void)hipGetLastError();   // "not a HIP pointer" is the ordinary answer for pageable memory
    return ok;
}

// Frames in host memory (the reference's caller hands a host IplImage, samples/ARTest.cpp:44-57).
// The call is cut into sub-batches; while the kernels of sub-batch k run, sub-batch k+1 is copied into one of the context's two
// page-locked staging buffers (by a few host threads) and from there to the device by the copy engine, and the in-place grey of
// sub-batch k-1 (opencvar.cpp:624-627) travels back the same way on a third stream.  The library NEVER page-locks the caller's
// memory (no hipHostRegister / hipHostUnregister: round 2's version did that, and a small heap-resident batch -- which shares
// its first and last page with whatever malloc put next to it -- ended in a GPU memory fault on a host address; DESIGN.md
// section 9 lists what that range shared pages with).  A caller that wants the copies straight from its own buffer
// page-locks it itself (hipHostMalloc, or hipHostRegister for as long as it likes): such a buffer is recognised
// (hipPointerGetAttributes) and used in place.
constexpr int HOST_SUB_BATCH = 64;

extern "C" int ocvar_hip_detect_host(OcvarHip* c, uint8_t* h_bgr, int width, int height, int row_stride, size_t frame_stride,
                                     int n_frames, int grey_in_place, const OcvarMarker* prev, const int* prev_counts,
                                     OcvarMarker* markers, int* counts, int max_per_frame) {
    if (!c || !h_bgr || n_frames < 1 || height < 1 || row_stride < 1) return OCVAR_E_ARG;
    if (n_frames > 1 && frame_stride < (size_t)height * row_stride) return OCVAR_E_ARG;
    if (!counts || max_per_frame < 0 || (max_per_frame > 0 && !markers)) return OCVAR_E_ARG;
    if (c->pending) {
        c->err = "the previous batch of this context has not been collected";
        return OCVAR_E_ARG;
    }
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t frame_bytes = (size_t)height * row_stride;
    const size_t bytes = (size_t)(n_frames - 1) * frame_stride + frame_bytes;
    const int sub = c->ws.max_batch < HOST_SUB_BATCH ? c->ws.max_batch : HOST_SUB_BATCH;
    const int n_sub = (n_frames + sub - 1) / sub;
    auto span = [&](int k, size_t* off, int* cnt) {   // sub-batch k: byte offset in the caller's buffer, frames, bytes
        *cnt = (k + 1) * sub <= n_frames ? sub : n_frames - k * sub;
        *off = (size_t)k * sub * frame_stride;
        return (size_t)(*cnt - 1) * frame_stride + frame_bytes;
    };
    const size_t span_max = (size_t)((n_frames < sub ? n_frames : sub) - 1) * frame_stride + frame_bytes;
    const bool direct = caller_pinned(h_bgr, bytes);
    int rc = reserve_staging(c, 2 * span_max + 256);   // two device slots (256-byte aligned)
    if (rc) return rc;
    const size_t slot_bytes = (span_max + 255) & ~(size_t)255;
    if (!direct && (rc = reserve_host_stage(c, span_max))) return rc;
    if (!direct && grey_in_place && (rc = reserve_host_grey(c, span_max))) return rc;
    if (!c->h2d_stream) HIP_TRY(c, hipStreamCreateWithFlags(&c->h2d_stream, hipStreamNonBlocking));
    if (!c->d2h_stream) HIP_TRY(c, hipStreamCreateWithFlags(&c->d2h_stream, hipStreamNonBlocking));
    while (c->h2d_done.size() < 2) {
        hipEvent_t ev;
        HIP_TRY(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        c->h2d_done.push_back(ev);
    }
    // On any failure: nothing of this call may still be in flight when it returns (the staging buffers are reused by the next)
    std::thread* copier_ref = nullptr;   // (set once the helper thread exists: fail() must not leave it running)
    auto fail = [&](int code) {
        (void)hipStreamSynchronize(c->h2d_stream);
        (void)hipStreamSynchronize(c->d2h_stream);
        (void)hipStreamSynchronize(c->stream);
        if (copier_ref && copier_ref->joinable()) copier_ref->join();
        c->pending = false;
        return code;
    };
#define HIP_TRY_HOST(expr)                                            \
    do {                                                              \
        hipError_t e_ = (expr);                                       \
        if (e_ != hipSuccess) {                                       \
            c->err = std::string(#expr) + ": " + hipGetErrorString(e_); \
            return fail(OCVAR_E_HIP);                                 \
        }                                                             \
    } while (0)
    // slot k & 1 (host and device) carries sub-batch k
    auto upload = [&](int k) -> hipError_t {
        size_t off;
        int cnt;
        const size_t len = span(k, &off, &cnt);
        const uint8_t* src = h_bgr + off;
        if (!direct) {
            host_copy(c->h_stage[k & 1], src, len);
            src = c->h_stage[k & 1];
        }
        hipError_t e = hipMemcpyAsync(c->d_frames + (size_t)(k & 1) * slot_bytes, src, len, hipMemcpyHostToDevice, c->h2d_stream);
        if (e == hipSuccess) e = hipEventRecord(c->h2d_done[k & 1], c->h2d_stream);
        return e;
    };
    // the grey frames of sub-batch k, already on their way: wait for the copy engine, then a helper thread copies them from the
    // staging buffer into the caller's frames while this thread stages the next upload (different buffers)
    std::thread grey_copier;
    copier_ref = &grey_copier;
    auto grey_join = [&] { if (grey_copier.joinable()) grey_copier.join(); };
    auto grey_home = [&](int k) -> hipError_t {
        const hipError_t e = hipStreamSynchronize(c->d2h_stream);
        if (e != hipSuccess || direct) return e;
        size_t off;
        int cnt;
        const size_t len = span(k, &off, &cnt);
        grey_join();
        uint8_t* dst = h_bgr + off;
        const uint8_t* src = c->h_grey[k & 1];
        try {
            grey_copier = std::thread([=] { host_copy(dst, src, len); });
        } catch (...) {
            host_copy(dst, src, len);
        }
        return hipSuccess;
    };
    HIP_TRY_HOST(upload(0));
    for (int k = 0; k < n_sub; k++) {
        size_t off;
        int cnt;
        const size_t len = span(k, &off, &cnt);
        uint8_t* d_slot = c->d_frames + (size_t)(k & 1) * slot_bytes;
        HIP_TRY_HOST(hipStreamWaitEvent(c->stream, c->h2d_done[k & 1], 0));
        rc = enqueue_impl(c, d_slot, width, height, row_stride, frame_stride, cnt, grey_in_place,
                          prev ? prev + (size_t)k * sub * MAXM : nullptr, prev_counts ? prev_counts + k * sub : nullptr, c->stream, 3);
        if (rc) return fail(rc);
        // while sub-batch k computes: bring sub-batch k-1's grey home, then stage sub-batch k+1 into the slot it leaves
        if (k > 0 && grey_in_place) HIP_TRY_HOST(grey_home(k - 1));
        if (k + 1 < n_sub) HIP_TRY_HOST(upload(k + 1));
        rc = ocvar_hip_collect(c, markers ? markers + (size_t)k * sub * max_per_frame : nullptr, counts + k * sub, max_per_frame);
        if (rc) return fail(rc);
        if (grey_in_place)   // (the kernels of sub-batch k have finished: collect waited for them)
            // (slot k & 1 last held sub-batch k - 2, whose copy-back thread was joined when sub-batch k - 1's was started)
            HIP_TRY_HOST(hipMemcpyAsync(direct ? h_bgr + off : c->h_grey[k & 1], d_slot, len, hipMemcpyDeviceToHost, c->d2h_stream));
    }
    if (grey_in_place) HIP_TRY_HOST(grey_home(n_sub - 1));
    grey_join();
#undef HIP_TRY_HOST
    return OCVAR_OK;
}

extern "C" int ocvar_hip_find_squares(OcvarHip* c, const uint8_t* h_gray, int width, int height, int row_stride, int* quads,
                                      int max_quads, int* n_quads) {
    if (!c || !h_gray || !quads || !n_quads || width < 16 || height < 16 || row_stride < width) return OCVAR_E_ARG;
    std::vector<uint8_t> bgr((size_t)width * height * 3);
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) {
            const uint8_t g = h_gray[(size_t)y * row_stride + x];
            uint8_t* p = &bgr[((size_t)y * width + x) * 3];
            p[0] = p[1] = p[2] = g;  // grey of an equal-channel pixel is the pixel
        }
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = stage_frames(c, bgr.data(), height, width * 3, (size_t)width * height * 3, 1);
    if (rc) return rc;
    rc = enqueue_impl(c, c->d_frames, width, height, width * 3, (size_t)width * height * 3, 1, 0, nullptr, nullptr, c->stream, 2);
    if (rc) return rc;
    rc = wait_impl(c);
    if (rc) return rc;
    int n = 0;
    HIP_TRY(c, hipMemcpy(&n, c->ws.n_squares, sizeof(int), hipMemcpyDeviceToHost));
    std::vector<float> sq((size_t)c->ws.maxq * 8);
    HIP_TRY(c, hipMemcpy(sq.data(), c->ws.squares, sq.size() * sizeof(float), hipMemcpyDeviceToHost));
    *n_quads = n;
    for (int i = 0; i < n && i < max_quads; i++)
        for (int k = 0; k < 8; k++) quads[8 * i + k] = (int)sq[8 * (size_t)i + k];
    return OCVAR_OK;
}

extern "C" int ocvar_hip_debug_gray(OcvarHip* c, int frame, uint8_t* h) {
    if (!c || !h || frame < 0 || frame >= c->ws.n_frames) return OCVAR_E_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    const int W = c->ws.W, H = c->ws.H, pitch = gray_pitch(W);
    const size_t plane = (size_t)gray_plane_bytes(W, H);
    std::vector<uint8_t> g(plane);
    HIP_TRY(c, hipMemcpy(g.data(), c->ws.gray + frame * plane, plane, hipMemcpyDeviceToHost));
    for (int y = 0; y < H; y++)   // out of the panels, into plain rows
        for (int x = 0; x < W; x++) h[(size_t)y * W + x] = g[(size_t)y * pitch + gray_col(x)];
    return OCVAR_OK;
}

extern "C" int ocvar_hip_debug_binary(OcvarHip* c, int frame, uint8_t* h) {
    if (!c || !h || frame < 0 || frame >= c->ws.n_frames) return OCVAR_E_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    const int sw = c->ws.sw, sh = c->ws.sh, ns = c->ws.ns;
    const size_t bytes = (size_t)nbr_plane_bytes(ns, sh);
    std::vector<uint8_t> nbr(bytes);
    HIP_TRY(c, hipMemcpy(nbr.data(), c->ws.nbr_frame + (size_t)frame * bytes, bytes, hipMemcpyDeviceToHost));
    // pixel (x,y) is the west neighbour (bit 4) of (x+1,y); the 1-px frame is zero as cvFindContours makes it
    for (int y = 0; y < sh; y++)
        for (int x = 0; x < sw; x++) h[(size_t)y * sw + x] = (x + 1 < sw && ((nbr[(size_t)nbr_addr(x + 1, y, ns)] >> 4) & 1)) ? 255 : 0;
    return OCVAR_OK;
}

extern "C" int ocvar_hip_debug_masks(OcvarHip* c, int frame, uint8_t* h) {
    if (!c || !h || frame < 0 || frame >= c->ws.n_frames) return OCVAR_E_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    const int sw = c->ws.sw, sh = c->ws.sh, ns = c->ws.ns;
    const size_t bytes = (size_t)nbr_plane_bytes(ns, sh);
    std::vector<uint8_t> nbr(bytes);
    HIP_TRY(c, hipMemcpy(nbr.data(), c->ws.nbr_frame + (size_t)frame * bytes, bytes, hipMemcpyDeviceToHost));
    for (int y = 0; y < sh; y++)
        for (int x = 0; x < sw; x++) h[(size_t)y * sw + x] = nbr[(size_t)nbr_addr(x, y, ns)];
    return OCVAR_OK;
}

extern "C" int ocvar_hip_debug_frame_quads(OcvarHip* c, int frame, int* quads, int* n_quads) {
    if (!c || !quads || !n_quads || frame < 0 || frame >= c->ws.n_frames) return OCVAR_E_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    int n = 0;
    HIP_TRY(c, hipMemcpy(&n, c->ws.n_squares + frame, sizeof(int), hipMemcpyDeviceToHost));
    // `quads` holds OCVAR_MAX_QUADS quads (the documented size); a context made for fewer or more squares per frame
    // (ocvar_hip_create_ex) reports what both hold
    const int lim = c->ws.maxq < OCVAR_MAX_QUADS ? c->ws.maxq : OCVAR_MAX_QUADS;
    std::vector<float> sq((size_t)lim * 8);
    HIP_TRY(c, hipMemcpy(sq.data(), c->ws.squares + (size_t)frame * c->ws.maxq * 8, sq.size() * sizeof(float), hipMemcpyDeviceToHost));
    *n_quads = n;
    for (int i = 0; i < n && i < lim; i++)
        for (int k = 0; k < 8; k++) quads[8 * i + k] = (int)sq[8 * (size_t)i + k];
    return OCVAR_OK;
}

extern "C" int ocvar_hip_debug_candidates(OcvarHip* c, int frame, OcvarCandidate* cands, int max_cands, int* n_cands) {
    if (!c || !cands || !n_cands || frame < 0 || frame >= c->ws.n_frames) return OCVAR_E_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    int nsq = 0;
    HIP_TRY(c, hipMemcpy(&nsq, c->ws.n_squares + frame, sizeof(int), hipMemcpyDeviceToHost));
    if (nsq > c->ws.maxq) nsq = c->ws.maxq;
    std::vector<CandRec> recs((size_t)c->ws.maxq * MAXT);
    HIP_TRY(c, hipMemcpy(recs.data(), c->ws.cand_recs + (size_t)frame * c->ws.maxq * MAXT, recs.size() * sizeof(CandRec), hipMemcpyDeviceToHost));
    int n = 0;
    for (int i = 0; i < nsq; i++)
        for (int j = 0; j < c->ws.n_templates; j++) {
            const CandRec& r = recs[(size_t)i * MAXT + j];
            if (!r.valid) continue;
            if (n < max_cands) {
                OcvarCandidate& o = cands[n];
                o.markerId = i;
                o.templateId = j;
                o.orient = r.orient;
                o.valid = 1;
                o.bit = r.bit;
                std::memcpy(o.square, r.square, sizeof o.square);
                std::memcpy(o.patPoint, r.patPoint, sizeof o.patPoint);
            }
            n++;
        }
    *n_cands = n;
    return OCVAR_OK;
}

extern "C" int ocvar_hip_stage_ms(OcvarHip* c, float* ms, int n) {
    if (!c || !ms || n < 1 || c->pending) return OCVAR_E_ARG;
    int k = 0;
    for (; k < 11 && k < n; k++)
        if (hipEventElapsedTime(&ms[k], c->ev[k], c->ev[k + 1]) != hipSuccess) ms[k] = -1.f;
    if (k < n && k == 11) {
        if (hipEventElapsedTime(&ms[11], c->ev[0], c->ev[12]) != hipSuccess) ms[11] = -1.f;
        k = 12;
    }
    return k;
}

// Where the last batch's 13 stage events lie on the device's clock, in milliseconds after the caller's reference event
// (recorded on any stream of this device before the batch was enqueued).  With several contexts in flight the launches of one
// kernel overlap each other; their start/end stamps let a caller compute how long the GPU was running that kernel at all.
extern "C" int ocvar_hip_stage_stamps(OcvarHip* c, void* ref_event, float* ms, int n) {
    if (!c || !ref_event || !ms || n < 1 || c->pending) return OCVAR_E_ARG;
    int k = 0;
    for (; k < 13 && k < n; k++)
        if (hipEventElapsedTime(&ms[k], (hipEvent_t)ref_event, c->ev[k]) != hipSuccess) {
            (void)hipGetLastError();
            ms[k] = -1.f;
        }
    return k;
}

// Calibration load for rocprofv3's FETCH_SIZE / WRITE_SIZE counters (MI355X_MICROARCH.md, HBM section: the
// counters are only calibrated for 16-byte-per-lane streams): copies `bytes` with the access widths the binarise kernel
// uses (one dword per lane, coalesced), so a profile of this launch gives bytes-per-counter-unit for that pattern.
__global__ void calib_copy_dword_kernel(const unsigned* src, unsigned* dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i] + 1u;
}

extern "C" int ocvar_hip_debug_calibrate(OcvarHip* c, size_t bytes) {
    if (!c || bytes < 1024) return OCVAR_E_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    unsigned *a = nullptr, *b = nullptr;
    HIP_TRY(c, hipMalloc((void**)&a, bytes));
    HIP_TRY(c, hipMalloc((void**)&b, bytes));
    HIP_TRY(c, hipMemset(a, 1, bytes));
    hipLaunchKernelGGL(calib_copy_dword_kernel, dim3(4096), dim3(256), 0, c->stream, a, b, bytes / 4);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    (void)hipFree(a);
    (void)hipFree(b);
    return OCVAR_OK;
}

extern "C" int ocvar_hip_counters(OcvarHip* c, long long* out, int n) {
    if (!c || !out || n < 1 || c->pending) return OCVAR_E_ARG;
    const int* h = c->h_counters;
    long long v[10] = {h[CNT_FRAME_CANDS], h[CNT_CROP_ROIS], h[CNT_CROP_TILES], h[CNT_CROP_CANDS],
                      (long long)*reinterpret_cast<const unsigned long long*>(h + CNT_CROP_PIXELS),
                      (long long)*reinterpret_cast<const unsigned long long*>(h + CNT_POOL_INTS),
                      h[CNT_MID_F], h[CNT_MID_C], h[CNT_LONG_F], h[CNT_LONG_C]};
    int k = 0;
    for (; k < 10 && k < n; k++) out[k] = v[k];
    // values 10..41: profiling slots (zero unless the library was built with -DOCVAR_PROF)
    for (; k < 42 && k < n; k++) out[k] = (long long)reinterpret_cast<const unsigned long long*>(h + CNT_PROF)[k - 10];
    return k;
}
