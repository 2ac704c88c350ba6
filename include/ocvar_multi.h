/*
 * ocvar_multi.h -- one process, several MI355X: frames sharded over the GPUs of a node, CvarMarker arrays gathered to one
 * root GPU over RCCL (xGMI).
 *
 * This is the multi-GPU form of the reference's only call site, the per-frame loop of samples/ARTest.cpp:43-82, for a
 * C/C++ caller that owns several frames at once (independent frames are the shards, "one process with
 * ncclCommInitAll"; BASELINE.json configs[3]: frame f -> GPU f mod N, RCCL gather of the CvarMarker arrays).  Each device
 * runs the single-GPU path of include/ocvar_hip.h on its frames; there is no data-path collective.  The only exchange is
 * ONE ncclGather (/opt/rocm/include/rccl/rccl.h:745) of the fixed-size result block per batch -- [frames][k] marker records,
 * k = the max_per_frame of the call (at most OCVAR_MAX_MARKERS), followed by [frames] full counts, as bytes -- to the root
 * device, which copies it to the host and puts the frames back in caller order.  The messages are small (184 B per record:
 * 1.5 KB per frame at 8 records, 12 KB at 64): latency-bound, so one collective per batch.
 *
 * Plain pointers and sizes; returns 0 or a negative OCVAR_E_* code (ocvar_hip.h); none throws.
 */
#ifndef OCVAR_MULTI_H
#define OCVAR_MULTI_H

#include "ocvar_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct OcvarMulti OcvarMulti;

enum { OCVAR_E_RCCL = -5 };   /* an RCCL call failed; see ocvar_multi_last_error */

/* devices: n_devices HIP device indices (NULL: 0 .. n_devices-1); the first one is the gather root.
 * max_frames_per_device: largest per-device share of a batch. */
int ocvar_multi_create(OcvarMulti** m, const int* devices, int n_devices, int max_width, int max_height, int max_frames_per_device);
void ocvar_multi_destroy(OcvarMulti* m);
const char* ocvar_multi_last_error(const OcvarMulti* m);
int ocvar_multi_devices(const OcvarMulti* m);

int ocvar_multi_set_templates(OcvarMulti* m, const OcvarTemplate* templates, int n);
int ocvar_multi_set_camera(OcvarMulti* m, const OcvarCamera* camera);

/* Stateless detection of n_frames (<= n_devices * max_frames_per_device) frames in host memory; frame f runs on device
 * f mod n_devices.  markers [n_frames][max_per_frame] and counts [n_frames] are host outputs in caller order
 * (cvarArMultRegistration's result per frame). */
int ocvar_multi_detect_host(OcvarMulti* m, const uint8_t* h_bgr, int width, int height, int row_stride, size_t frame_stride,
                            int n_frames, OcvarMarker* markers, int* counts, int max_per_frame);

/* Same with every device's share already resident in its memory: d_bgr[d] holds frames d, d + N, d + 2N, ... of the batch
 * back to back (frame_stride apart), n_local[d] of them.  Global frame index of local frame i on device d: d + N*i. */
int ocvar_multi_detect_device(OcvarMulti* m, uint8_t* const* d_bgr, int width, int height, int row_stride, size_t frame_stride,
                              const int* n_local, OcvarMarker* markers, int* counts, int max_per_frame);

/* Stateful form: one call = one time step of n_streams video streams (frame s = stream s's current frame); stream s lives on
 * device s mod n_devices, local slot s / n_devices, for as long as the tracker is used, and its markers of the previous step
 * -- the tracking input of cvarArMultRegistration, /root/reference/src/opencvar.cpp:635-668 -- stay in that device's memory
 * between calls ("stateful tracking is sequential per stream: shard by stream").  reset != 0: the streams
 * start without markers (a stateless first step).  The gather is the same single ncclGather per step. */
int ocvar_multi_track_host(OcvarMulti* m, const uint8_t* h_bgr, int width, int height, int row_stride, size_t frame_stride,
                           int n_streams, int reset, OcvarMarker* markers, int* counts, int max_per_frame);
int ocvar_multi_track_device(OcvarMulti* m, uint8_t* const* d_bgr, int width, int height, int row_stride, size_t frame_stride,
                             const int* n_local, int reset, OcvarMarker* markers, int* counts, int max_per_frame);

#ifdef __cplusplus
}
#endif
#endif
