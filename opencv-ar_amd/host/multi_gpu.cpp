// multi_gpu.cpp -- include/ocvar_multi.h: one process, N GPUs, frames sharded by index, one RCCL gather per batch.
//
// Reference side: the per-frame caller loop of /root/reference/samples/ARTest.cpp:43-82 (one cvarArMultRegistration per
// frame); frames are independent in stateless mode, so they are the shards.  Built as its own library
// (lib/libocvar_multi.so, links librccl) so that the single-GPU library carries no RCCL dependency.
#include "ocvar_multi.h"
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

struct OcvarMulti {
    int n = 0, max_local = 0;
    std::vector<int> dev;
    std::vector<OcvarHip*> ctx;
    std::vector<ncclComm_t> comm;
    std::vector<hipStream_t> stream;
    std::vector<uint8_t*> d_block;    // per device: [max_local] x MAXM markers, then [max_local] counts
    std::vector<uint8_t*> d_frames;   // per device: staging of ocvar_multi_detect_host
    std::vector<OcvarMarker*> d_state;      // per device: tracking state of its streams, [max_local][MAXM] (ocvar_multi_track_*)
    std::vector<int*> d_state_counts;       // per device: [max_local]
    std::vector<uint8_t*> h_frames;   // per device: page-locked host staging (the copy engines never see the caller's own memory)
    size_t d_frames_bytes = 0;
    uint8_t* d_all = nullptr;         // root: n blocks
    uint8_t* h_all = nullptr;         // pinned
    size_t block_bytes = 0;
    std::string err;
};

namespace {
constexpr int MAXM = OCVAR_MAX_MARKERS;
size_t marker_bytes(int frames, int per_frame = MAXM) { return (size_t)frames * per_frame * sizeof(OcvarMarker); }

#define M_HIP(m, call)                                                              \
    do {                                                                            \
        hipError_t e_ = (call);                                                     \
        if (e_ != hipSuccess) {                                                     \
            (m)->err = std::string(#call) + ": " + hipGetErrorString(e_);          \
            return OCVAR_E_HIP;                                                     \
        }                                                                           \
    } while (0)
#define M_NCCL(m, call)                                                             \
    do {                                                                            \
        ncclResult_t r_ = (call);                                                   \
        if (r_ != ncclSuccess) {                                                    \
            (m)->err = std::string(#call) + ": " + ncclGetErrorString(r_);         \
            return OCVAR_E_RCCL;                                                    \
        }                                                                           \
    } while (0)
}  // namespace

extern "C" int ocvar_multi_create(OcvarMulti** out, const int* devices, int n_devices, int max_width, int max_height, int max_frames_per_device) {
    if (!out || n_devices < 1 || max_frames_per_device < 1) return OCVAR_E_ARG;
    *out = nullptr;
    // RCCL between GPUs needs dmabuf IPC on hosts whose driver has no legacy IPC (this pool): HSA_ENABLE_IPC_MODE_LEGACY=0,
    // read by the ROCm runtime when it initialises.  Set here if the caller's environment lacks it -- effective only when
    // this is the process's first HIP call; a process that touched HIP earlier must export it itself.
    (void)setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);
    (void)setenv("GPU_MAX_HW_QUEUES", "12", 0);   // (one stream per device here, but RCCL brings its own: see INTEGRATION.md, "Hardware queues")
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < n_devices) return OCVAR_E_NO_DEVICE;
    OcvarMulti* m = new OcvarMulti();
    *out = m;   // returned even on failure so the caller can read the error text, then destroy
    m->n = n_devices;
    m->max_local = max_frames_per_device;
    for (int d = 0; d < n_devices; d++) m->dev.push_back(devices ? devices[d] : d);
    m->block_bytes = marker_bytes(max_frames_per_device) + (size_t)max_frames_per_device * sizeof(int);
    m->ctx.assign(n_devices, nullptr);
    m->stream.assign(n_devices, nullptr);
    m->d_block.assign(n_devices, nullptr);
    m->d_frames.assign(n_devices, nullptr);
    m->h_frames.assign(n_devices, nullptr);
    m->d_state.assign(n_devices, nullptr);
    m->d_state_counts.assign(n_devices, nullptr);
    for (int d = 0; d < n_devices; d++) {
        const int rc = ocvar_hip_create(&m->ctx[d], m->dev[d], max_width, max_height, max_frames_per_device);
        if (rc != OCVAR_OK) {
            m->err = std::string("ocvar_hip_create on device ") + std::to_string(m->dev[d]) + ": " +
                     (m->ctx[d] ? ocvar_hip_last_error(m->ctx[d]) : "no gfx950 device");
            return rc;
        }
        M_HIP(m, hipSetDevice(m->dev[d]));
        M_HIP(m, hipStreamCreateWithFlags(&m->stream[d], hipStreamNonBlocking));
        M_HIP(m, hipMalloc((void**)&m->d_block[d], m->block_bytes));
        M_HIP(m, hipMalloc((void**)&m->d_state[d], marker_bytes(max_frames_per_device)));
        M_HIP(m, hipMalloc((void**)&m->d_state_counts[d], (size_t)max_frames_per_device * sizeof(int)));
        M_HIP(m, hipMemset(m->d_state_counts[d], 0, (size_t)max_frames_per_device * sizeof(int)));
        M_HIP(m, hipDeviceSynchronize());
    }
    M_HIP(m, hipSetDevice(m->dev[0]));
    M_HIP(m, hipMalloc((void**)&m->d_all, m->block_bytes * n_devices));
    M_HIP(m, hipHostMalloc((void**)&m->h_all, m->block_bytes * n_devices));
    m->comm.assign(n_devices, nullptr);
    M_NCCL(m, ncclCommInitAll(m->comm.data(), n_devices, m->dev.data()));
    return OCVAR_OK;
}

extern "C" void ocvar_multi_destroy(OcvarMulti* m) {
    if (!m) return;
    for (size_t d = 0; d < m->comm.size(); d++)
        if (m->comm[d]) (void)ncclCommDestroy(m->comm[d]);
    for (int d = 0; d < (int)m->ctx.size(); d++) {
        (void)hipSetDevice(m->dev[d]);
        if (m->stream[d]) {
            (void)hipStreamSynchronize(m->stream[d]);
            (void)hipStreamDestroy(m->stream[d]);
        }
        if (m->d_block[d]) (void)hipFree(m->d_block[d]);
        if (m->d_frames[d]) (void)hipFree(m->d_frames[d]);
        if (m->h_frames[d]) (void)hipHostFree(m->h_frames[d]);
        if (m->d_state[d]) (void)hipFree(m->d_state[d]);
        if (m->d_state_counts[d]) (void)hipFree(m->d_state_counts[d]);
        if (m->ctx[d]) ocvar_hip_destroy(m->ctx[d]);
    }
    if (!m->dev.empty()) (void)hipSetDevice(m->dev[0]);
    if (m->d_all) (void)hipFree(m->d_all);
    if (m->h_all) (void)hipHostFree(m->h_all);
    delete m;
}

extern "C" const char* ocvar_multi_last_error(const OcvarMulti* m) { return m ? m->err.c_str() : "null context"; }
extern "C" int ocvar_multi_devices(const OcvarMulti* m) { return m ? m->n : 0; }

extern "C" int ocvar_multi_set_templates(OcvarMulti* m, const OcvarTemplate* t, int n) {
    if (!m) return OCVAR_E_ARG;
    for (int d = 0; d < m->n; d++) {
        const int rc = ocvar_hip_set_templates(m->ctx[d], t, n);
        if (rc) { m->err = ocvar_hip_last_error(m->ctx[d]); return rc; }
    }
    return OCVAR_OK;
}

extern "C" int ocvar_multi_set_camera(OcvarMulti* m, const OcvarCamera* cam) {
    if (!m) return OCVAR_E_ARG;
    for (int d = 0; d < m->n; d++) {
        const int rc = ocvar_hip_set_camera(m->ctx[d], cam);
        if (rc) { m->err = ocvar_hip_last_error(m->ctx[d]); return rc; }
    }
    return OCVAR_OK;
}

// tracked: every device takes its frames' previous markers from its device-resident state (stream s = local slot s / N of
// device s mod N) and leaves the new ones there, behind the kernels on its own stream.
static int multi_run(OcvarMulti* m, uint8_t* const* d_bgr, int width, int height, int row_stride, size_t frame_stride,
                     const int* n_local, bool tracked, OcvarMarker* markers, int* counts, int max_per_frame) {
    if (!m || !d_bgr || !n_local || !counts || max_per_frame < 0 || (max_per_frame > 0 && !markers)) return OCVAR_E_ARG;
    const int N = m->n;
    for (int d = 0; d < N; d++)
        if (n_local[d] < 0 || n_local[d] > m->max_local || (n_local[d] > 0 && !d_bgr[d])) return OCVAR_E_ARG;
    // The blocks carry the first K records of every frame -- as many as the caller takes per frame -- and the frames' full
    // counts (ocvar_hip_results_to_device_ex): a caller that keeps 8 markers per frame gathers 1.5 KB per frame instead of 12.
    const int K = max_per_frame < 1 ? 1 : (max_per_frame > MAXM ? MAXM : max_per_frame);
    const size_t block_bytes = marker_bytes(m->max_local, K) + (size_t)m->max_local * sizeof(int);   // <= m->block_bytes (K = MAXM)
    // A failure after the first enqueue must not leave batches behind: a context refuses a new batch while one is
    // uncollected, so every device enqueued so far is drained (collected without results) before the error goes back.
    std::vector<char> enqueued((size_t)N, 0);
    std::vector<int> scratch_counts((size_t)m->max_local);
    auto drain = [&](int code, const std::string& what) {
        m->err = what;
        for (int d = 0; d < N; d++)
            if (enqueued[d]) {
                (void)hipSetDevice(m->dev[d]);
                (void)ocvar_hip_collect(m->ctx[d], nullptr, scratch_counts.data(), 0);
            }
        return code;
    };
    // every device: detect its share (all kernels of the single-GPU path on the device's own stream), then put the result
    // block into the send buffer behind them
    for (int d = 0; d < N; d++) {
        if (n_local[d] == 0) continue;
        if (hipSetDevice(m->dev[d]) != hipSuccess) return drain(OCVAR_E_HIP, "hipSetDevice failed");
        int rc = tracked ? ocvar_hip_enqueue_tracked(m->ctx[d], d_bgr[d], width, height, row_stride, frame_stride, n_local[d], 0,
                                                     m->d_state[d], m->d_state_counts[d], m->stream[d])
                         : ocvar_hip_enqueue(m->ctx[d], d_bgr[d], width, height, row_stride, frame_stride, n_local[d], 0, nullptr, nullptr, m->stream[d]);
        if (rc) return drain(rc, ocvar_hip_last_error(m->ctx[d]));
        enqueued[d] = 1;
        if (tracked) {
            rc = ocvar_hip_results_to_device(m->ctx[d], m->d_state[d], m->d_state_counts[d], m->stream[d]);
            if (rc) return drain(rc, ocvar_hip_last_error(m->ctx[d]));
        }
        rc = ocvar_hip_results_to_device_ex(m->ctx[d], reinterpret_cast<OcvarMarker*>(m->d_block[d]),
                                            reinterpret_cast<int*>(m->d_block[d] + marker_bytes(m->max_local, K)), K, m->stream[d]);
        if (rc) return drain(rc, ocvar_hip_last_error(m->ctx[d]));
    }
    // one gather of the fixed-size blocks to the root (device 0 of the list); a single-thread caller issues all ranks'
    // calls inside one group
    {
        ncclResult_t r = ncclGroupStart();
        for (int d = 0; d < N && r == ncclSuccess; d++)
            r = ncclGather(m->d_block[d], d == 0 ? m->d_all : nullptr, block_bytes, ncclUint8, 0, m->comm[d], m->stream[d]);
        const ncclResult_t e = ncclGroupEnd();
        if (r == ncclSuccess) r = e;
        if (r != ncclSuccess) return drain(OCVAR_E_RCCL, std::string("ncclGather: ") + ncclGetErrorString(r));
    }
    if (hipSetDevice(m->dev[0]) != hipSuccess ||
        hipMemcpyAsync(m->h_all, m->d_all, block_bytes * N, hipMemcpyDeviceToHost, m->stream[0]) != hipSuccess)
        return drain(OCVAR_E_HIP, "copy-out of the gathered blocks failed");
    // collect per device (reports capacity errors of that device's batch), root last: its stream carries the copy-out
    int first_err = OCVAR_OK;
    for (int d = N - 1; d >= 0; d--) {
        if (n_local[d] == 0) continue;
        M_HIP(m, hipSetDevice(m->dev[d]));
        const int rc = ocvar_hip_collect(m->ctx[d], nullptr, scratch_counts.data(), 0);
        if (rc && !first_err) {
            first_err = rc;
            m->err = std::string("device ") + std::to_string(m->dev[d]) + ": " + ocvar_hip_last_error(m->ctx[d]);
        }
    }
    M_HIP(m, hipSetDevice(m->dev[0]));
    M_HIP(m, hipStreamSynchronize(m->stream[0]));
    if (first_err) return first_err;
    // root: blocks in rank order -> caller order (global frame d + N*i)
    for (int d = 0; d < N; d++) {
        const uint8_t* blk = m->h_all + (size_t)d * block_bytes;
        const OcvarMarker* mk = reinterpret_cast<const OcvarMarker*>(blk);
        const int* cn = reinterpret_cast<const int*>(blk + marker_bytes(m->max_local, K));
        for (int i = 0; i < n_local[d]; i++) {
            const size_t g = (size_t)d + (size_t)N * i;
            counts[g] = cn[i];
            int k = cn[i] < max_per_frame ? cn[i] : max_per_frame;
            if (k > K) k = K;
            if (k > 0) std::memcpy(markers + g * max_per_frame, mk + (size_t)i * K, (size_t)k * sizeof(OcvarMarker));
        }
    }
    return OCVAR_OK;
}

extern "C" int ocvar_multi_detect_device(OcvarMulti* m, uint8_t* const* d_bgr, int width, int height, int row_stride, size_t frame_stride,
                                         const int* n_local, OcvarMarker* markers, int* counts, int max_per_frame) {
    return multi_run(m, d_bgr, width, height, row_stride, frame_stride, n_local, false, markers, counts, max_per_frame);
}

static int multi_reset_state(OcvarMulti* m) {
    for (int d = 0; d < m->n; d++) {
        M_HIP(m, hipSetDevice(m->dev[d]));
        M_HIP(m, hipMemset(m->d_state_counts[d], 0, (size_t)m->max_local * sizeof(int)));
        M_HIP(m, hipDeviceSynchronize());   // (the devices' streams do not synchronise with the null stream)
    }
    return OCVAR_OK;
}

extern "C" int ocvar_multi_track_device(OcvarMulti* m, uint8_t* const* d_bgr, int width, int height, int row_stride, size_t frame_stride,
                                        const int* n_local, int reset, OcvarMarker* markers, int* counts, int max_per_frame) {
    if (!m) return OCVAR_E_ARG;
    if (reset) {
        const int rc = multi_reset_state(m);
        if (rc) return rc;
    }
    const int rc = multi_run(m, d_bgr, width, height, row_stride, frame_stride, n_local, true, markers, counts, max_per_frame);
    if (rc && rc != OCVAR_E_ARG) (void)multi_reset_state(m);   // a failed step leaves no half-updated state behind
    return rc;
}

static int multi_stage_host(OcvarMulti* m, const uint8_t* h_bgr, int width, int height, int row_stride, size_t frame_stride, int n_frames,
                            std::vector<int>& n_local);

extern "C" int ocvar_multi_detect_host(OcvarMulti* m, const uint8_t* h_bgr, int width, int height, int row_stride, size_t frame_stride,
                                       int n_frames, OcvarMarker* markers, int* counts, int max_per_frame) {
    std::vector<int> n_local;
    const int rc = multi_stage_host(m, h_bgr, width, height, row_stride, frame_stride, n_frames, n_local);
    if (rc) return rc;
    return multi_run(m, m->d_frames.data(), width, height, row_stride, (size_t)height * row_stride, n_local.data(), false, markers, counts, max_per_frame);
}

extern "C" int ocvar_multi_track_host(OcvarMulti* m, const uint8_t* h_bgr, int width, int height, int row_stride, size_t frame_stride,
                                      int n_streams, int reset, OcvarMarker* markers, int* counts, int max_per_frame) {
    std::vector<int> n_local;
    const int rc = multi_stage_host(m, h_bgr, width, height, row_stride, frame_stride, n_streams, n_local);
    if (rc) return rc;
    return ocvar_multi_track_device(m, m->d_frames.data(), width, height, row_stride, (size_t)height * row_stride, n_local.data(), reset,
                                    markers, counts, max_per_frame);
}

// frames of a host batch onto their devices (frame f -> device f mod N, local slot f / N), ordered before the detection on
// each device's stream
static int multi_stage_host(OcvarMulti* m, const uint8_t* h_bgr, int width, int height, int row_stride, size_t frame_stride, int n_frames,
                            std::vector<int>& n_local_out) {
    if (!m || !h_bgr || n_frames < 1 || n_frames > m->n * m->max_local || height < 1 || row_stride < 3 * width) return OCVAR_E_ARG;
    if (n_frames > 1 && frame_stride < (size_t)height * row_stride) return OCVAR_E_ARG;
    const int N = m->n;
    const size_t fb = (size_t)height * row_stride;
    std::vector<int>& n_local = n_local_out;
    n_local.assign(N, 0);
    for (int f = 0; f < n_frames; f++) n_local[f % N]++;
    if (fb * m->max_local > m->d_frames_bytes) {
        for (int d = 0; d < N; d++) {
            M_HIP(m, hipSetDevice(m->dev[d]));
            if (m->d_frames[d]) (void)hipFree(m->d_frames[d]);
            m->d_frames[d] = nullptr;
            M_HIP(m, hipMalloc((void**)&m->d_frames[d], fb * m->max_local));
        }
        for (int d = 0; d < N; d++) {
            M_HIP(m, hipSetDevice(m->dev[d]));
            if (m->h_frames[d]) (void)hipHostFree(m->h_frames[d]);
            m->h_frames[d] = nullptr;
            M_HIP(m, hipHostMalloc((void**)&m->h_frames[d], fb * m->max_local));
        }
        m->d_frames_bytes = fb * m->max_local;
    }
    // frame f -> device f mod N, local slot f / N.  A device's frames are collected in its page-locked staging buffer (the
    // caller's buffer is ordinary pageable memory: the copy engines only ever see memory this library page-locked itself,
    // as in ocvar_hip_detect_host) and go over in one copy, ordered before the detection on the device's stream; the
    // transfer to device d overlaps the host-side collection for device d + 1.
    for (int d = 0; d < N; d++) {
        if (n_local[d] == 0) continue;
        for (int i = 0; i < n_local[d]; i++)
            std::memcpy(m->h_frames[d] + (size_t)i * fb, h_bgr + (size_t)(d + (size_t)N * i) * frame_stride, fb);
        M_HIP(m, hipSetDevice(m->dev[d]));
        M_HIP(m, hipMemcpyAsync(m->d_frames[d], m->h_frames[d], fb * n_local[d], hipMemcpyHostToDevice, m->stream[d]));
    }
    return OCVAR_OK;
}
