// pipe.hip -- several contexts on one GPU as one detector (include/ocvar_hip.h: ocvar_hip_pipe_*).
//
// One context runs a batch as 12 kernels back to back; its latency-bound border followers leave the memory pipeline idle and
// its streaming binarise kernels leave the followers' slots idle.  Several contexts in flight fill each other's gaps (DESIGN.md
// section 6: four contexts with at most two binarise kernels at a time reach 1.4x the frames/s of one).  bench.py does that
// orchestration in Python; this file is the same schedule behind the C ABI for C/C++ callers that hold many frames on the
// device: the frames are cut into chunks, the chunks go round the contexts, every context has one chunk in flight on its own
// stream, the first chunks are shortened so that the contexts do not run in lock-step, and the host only ever waits for the
// context whose chunk is oldest.  Built on the public entry points only (ocvar_hip_create / enqueue / collect / gate).
#include "ocvar_hip.h"
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <new>
#include <string>
#include <vector>

struct OcvarPipe {
    int device = 0, n_ctx = 0, chunk = 0;
    std::vector<OcvarHip*> ctx;
    OcvarGate* gate = nullptr;
    // tracking state of ocvar_hip_pipe_track_device: every stream's markers of the last step, on the device
    // streaming form (ocvar_hip_pipe_submit / _collect): chunk k runs on context k % n_ctx; chunks [tail, head) are in flight
    long long head = 0, tail = 0;
    std::vector<long long> tags;
    std::vector<int> counts_in_flight;
    OcvarMarker* d_state = nullptr;
    int* d_state_counts = nullptr;
    long long state_streams = 0;
    std::string err;
};

extern "C" void ocvar_hip_pipe_destroy(OcvarPipe* p) {
    if (!p) return;
    for (OcvarHip* c : p->ctx) ocvar_hip_destroy(c);
    ocvar_hip_gate_destroy(p->gate);
    if (p->d_state || p->d_state_counts) (void)hipSetDevice(p->device);
    if (p->d_state) (void)hipFree(p->d_state);
    if (p->d_state_counts) (void)hipFree(p->d_state_counts);
    delete p;
}

extern "C" int ocvar_hip_pipe_create(OcvarPipe** out, int device, int max_width, int max_height, int chunk_frames, int n_contexts,
                                     int gate_width) {
    if (!out || chunk_frames < 1 || n_contexts < 1 || n_contexts > 16 || gate_width < 0) return OCVAR_E_ARG;
    *out = nullptr;
    // every context's stream needs a hardware queue of its own ("Hardware queues"): effective only when this is the
    // process's first HIP call -- a process that touched HIP earlier exports GPU_MAX_HW_QUEUES itself
    (void)setenv("GPU_MAX_HW_QUEUES", n_contexts + 4 > 8 ? "20" : "8", 0);
    OcvarPipe* p = new (std::nothrow) OcvarPipe();
    if (!p) return OCVAR_E_HIP;
    *out = p;   // returned even on failure below so that the caller can read the error text, then destroy
    p->device = device;
    p->n_ctx = n_contexts;
    p->chunk = chunk_frames;
    if (gate_width > 0 && n_contexts > 1) {
        const int rc = ocvar_hip_gate_create(&p->gate, device, gate_width);
        if (rc) { p->err = "ocvar_hip_gate_create failed"; return rc; }
    }
    p->tags.assign((size_t)n_contexts, 0);
    p->counts_in_flight.assign((size_t)n_contexts, 0);
    for (int i = 0; i < n_contexts; i++) {
        OcvarHip* c = nullptr;
        const int rc = ocvar_hip_create(&c, device, max_width, max_height, chunk_frames);
        if (c) p->ctx.push_back(c);
        if (rc) { p->err = c ? ocvar_hip_last_error(c) : "ocvar_hip_create failed"; return rc; }
        if (p->gate) (void)ocvar_hip_set_gate(c, p->gate);
    }
    return OCVAR_OK;
}

extern "C" const char* ocvar_hip_pipe_last_error(const OcvarPipe* p) { return p ? p->err.c_str() : "null pipe"; }

extern "C" int ocvar_hip_pipe_set_templates(OcvarPipe* p, const OcvarTemplate* t, int n) {
    if (!p) return OCVAR_E_ARG;
    for (OcvarHip* c : p->ctx) {
        const int rc = ocvar_hip_set_templates(c, t, n);
        if (rc) { p->err = ocvar_hip_last_error(c); return rc; }
    }
    return OCVAR_OK;
}

extern "C" int ocvar_hip_pipe_set_camera(OcvarPipe* p, const OcvarCamera* cam) {
    if (!p) return OCVAR_E_ARG;
    for (OcvarHip* c : p->ctx) {
        const int rc = ocvar_hip_set_camera(c, cam);
        if (rc) { p->err = ocvar_hip_last_error(c); return rc; }
    }
    return OCVAR_OK;
}

// The chunk schedule shared by the stateless and the stateful entry point.  tracked: chunk [start, start + cnt) takes its
// previous markers from the pipe's device-resident state and leaves its results there (stream-ordered behind its kernels;
// the state rows of different chunks are disjoint).
static int pipe_run(OcvarPipe* p, uint8_t* d_bgr, int width, int height, int row_stride, size_t frame_stride, long long n_frames,
                    int grey_in_place, bool tracked, OcvarMarker* markers, int* counts, int max_per_frame) {
    const int K = p->n_ctx;
    // the contexts bring back as many records per frame as the caller keeps (24 MB per 2048-frame chunk at 64, 3 MB at 8)
    const int limit = max_per_frame < 1 ? 1 : (max_per_frame > OCVAR_MAX_MARKERS ? OCVAR_MAX_MARKERS : max_per_frame);
    for (OcvarHip* c : p->ctx) {
        const int rc = ocvar_hip_set_result_limit(c, limit);
        if (rc) { p->err = "a context of the pipe still has a batch in flight"; return rc; }
    }
    struct Flight { long long start = 0; int count = 0; };
    std::vector<Flight> fl((size_t)K);
    long long next = 0;
    int first_err = OCVAR_OK;
    auto launch = [&](int i, int want) -> int {
        const long long left = n_frames - next;
        const int cnt = (int)(left < want ? left : want);
        fl[i].count = 0;
        if (cnt <= 0) return OCVAR_OK;
        uint8_t* src = d_bgr + (size_t)next * frame_stride;
        int rc;
        if (tracked) {
            OcvarMarker* st = p->d_state + (size_t)next * OCVAR_MAX_MARKERS;
            rc = ocvar_hip_enqueue_tracked(p->ctx[i], src, width, height, row_stride, frame_stride, cnt, grey_in_place, st, p->d_state_counts + next, nullptr);
            if (!rc) rc = ocvar_hip_results_to_device(p->ctx[i], st, p->d_state_counts + next, nullptr);
        } else {
            rc = ocvar_hip_enqueue(p->ctx[i], src, width, height, row_stride, frame_stride, cnt, grey_in_place, nullptr, nullptr, nullptr);
        }
        if (rc) { p->err = ocvar_hip_last_error(p->ctx[i]); return rc; }
        fl[i].start = next;
        fl[i].count = cnt;
        next += cnt;
        return OCVAR_OK;
    };
    // the first chunks are (i + 1) / K of a chunk: contexts that start together stay in lock-step, all in the streaming kernels,
    // then all in the followers
    for (int i = 0; i < K; i++) {
        int want = (int)(((long long)p->chunk * (i + 1)) / K);
        if (want < 1) want = 1;
        const int rc = launch(i, want);
        if (rc && !first_err) first_err = rc;
    }
    for (bool any = true; any;) {
        any = false;
        for (int i = 0; i < K; i++) {
            if (fl[i].count == 0) continue;
            any = true;
            const int rc = ocvar_hip_collect(p->ctx[i], markers ? markers + (size_t)fl[i].start * max_per_frame : nullptr,
                                             counts + fl[i].start, max_per_frame);
            if (rc && !first_err) {   // keep draining the other contexts; report the first failure
                first_err = rc;
                p->err = ocvar_hip_last_error(p->ctx[i]);
            }
            fl[i].count = 0;
            if (!first_err) {
                const int rc2 = launch(i, p->chunk);
                if (rc2 && !first_err) first_err = rc2;
            }
        }
    }
    return first_err;
}

extern "C" int ocvar_hip_pipe_detect_device(OcvarPipe* p, uint8_t* d_bgr, int width, int height, int row_stride, size_t frame_stride,
                                            long long n_frames, int grey_in_place, OcvarMarker* markers, int* counts, int max_per_frame) {
    if (!p || !d_bgr || n_frames < 1 || !counts || max_per_frame < 0 || (max_per_frame > 0 && !markers)) return OCVAR_E_ARG;
    return pipe_run(p, d_bgr, width, height, row_stride, frame_stride, n_frames, grey_in_place, false, markers, counts, max_per_frame);
}

extern "C" int ocvar_hip_pipe_track_device(OcvarPipe* p, uint8_t* d_bgr, int width, int height, int row_stride, size_t frame_stride,
                                           long long n_streams, int grey_in_place, int reset, OcvarMarker* markers, int* counts, int max_per_frame) {
    if (!p || !d_bgr || n_streams < 1 || !counts || max_per_frame < 0 || (max_per_frame > 0 && !markers)) return OCVAR_E_ARG;
    if (hipSetDevice(p->device) != hipSuccess) { p->err = "hipSetDevice failed"; return OCVAR_E_HIP; }
    if (n_streams != p->state_streams) {
        if (p->d_state) (void)hipFree(p->d_state);
        if (p->d_state_counts) (void)hipFree(p->d_state_counts);
        p->d_state = nullptr;
        p->d_state_counts = nullptr;
        p->state_streams = 0;
        if (hipMalloc((void**)&p->d_state, (size_t)n_streams * OCVAR_MAX_MARKERS * sizeof(OcvarMarker)) != hipSuccess ||
            hipMalloc((void**)&p->d_state_counts, (size_t)n_streams * sizeof(int)) != hipSuccess) {
            p->err = "out of device memory for the tracking state";
            return OCVAR_E_HIP;
        }
        p->state_streams = n_streams;
        reset = 1;
    }
    // (the contexts' streams do not synchronise with the null stream: the cleared counts must have landed before a chunk reads them)
    if (reset && (hipMemset(p->d_state_counts, 0, (size_t)n_streams * sizeof(int)) != hipSuccess || hipDeviceSynchronize() != hipSuccess)) {
        p->err = "clearing the tracking state failed";
        return OCVAR_E_HIP;
    }
    const int rc = pipe_run(p, d_bgr, width, height, row_stride, frame_stride, n_streams, grey_in_place, true, markers, counts, max_per_frame);
    if (rc) {   // a failed step leaves no half-updated state behind
        (void)hipMemset(p->d_state_counts, 0, (size_t)n_streams * sizeof(int));
        (void)hipDeviceSynchronize();
    }
    return rc;
}

// ---- streaming form --------------------------------------------------------------------------------------------------------
// ocvar_hip_pipe_detect_device fills and drains the pipeline once per call (159 k frames/s per 16 384-frame call against the 197 k
// of a pipeline that never drains).  A caller with an endless supply of frames -- video servers: the reference's per-frame loop,
// samples/ARTest.cpp:43-82, for many cameras -- keeps it full instead: submit hands the next chunk to the next context (stream
// order per context, the shared gate between them) and returns at once; collect waits for the OLDEST chunk in flight and returns
// its results with the tag it was submitted under.  At most n_contexts chunks are in flight.

extern "C" int ocvar_hip_pipe_in_flight(const OcvarPipe* p) { return p ? (int)(p->head - p->tail) : 0; }

extern "C" int ocvar_hip_pipe_submit(OcvarPipe* p, uint8_t* d_bgr, int width, int height, int row_stride, size_t frame_stride, int n_frames,
                                     int grey_in_place, long long tag) {
    if (!p || !d_bgr || n_frames < 1 || n_frames > p->chunk) return OCVAR_E_ARG;
    if (p->head - p->tail >= p->n_ctx) {
        p->err = "every context has a chunk in flight: collect one first";
        return OCVAR_E_BUSY;
    }
    const int i = (int)(p->head % p->n_ctx);
    const int rc = ocvar_hip_enqueue(p->ctx[i], d_bgr, width, height, row_stride, frame_stride, n_frames, grey_in_place, nullptr, nullptr, nullptr);
    if (rc) { p->err = ocvar_hip_last_error(p->ctx[i]); return rc; }
    p->tags[i] = tag;
    p->counts_in_flight[i] = n_frames;
    p->head++;
    return OCVAR_OK;
}

extern "C" int ocvar_hip_pipe_collect(OcvarPipe* p, long long* tag, OcvarMarker* markers, int* counts, int max_per_frame) {
    if (!p || !counts || max_per_frame < 0 || (max_per_frame > 0 && !markers)) return OCVAR_E_ARG;
    if (p->head == p->tail) return 0;   // nothing in flight
    const int i = (int)(p->tail % p->n_ctx);
    const int rc = ocvar_hip_collect(p->ctx[i], markers, counts, max_per_frame);
    p->tail++;   // (the chunk is gone either way: a failed batch is not collected twice)
    if (rc) { p->err = ocvar_hip_last_error(p->ctx[i]); return rc; }
    if (tag) *tag = p->tags[i];
    return p->counts_in_flight[i];
}

extern "C" int ocvar_hip_pipe_set_result_limit(OcvarPipe* p, int max_per_frame) {
    if (!p || p->head != p->tail) return OCVAR_E_ARG;
    for (OcvarHip* c : p->ctx) {
        const int rc = ocvar_hip_set_result_limit(c, max_per_frame);
        if (rc) return rc;
    }
    return OCVAR_OK;
}
