// decode.hip -- per-quad code readout and the per-frame tail (dedupe, pose, marker records) on gfx950.
//
// decode_kernel replaces the inner template loop of cvarArMultRegistration
// (/root/reference/src/opencvar.cpp:700-774) for one frame-pass quad per lane.  The reference re-runs the whole
// square finder on the crop once per template with an identical result; here the crop pass ran
// once and every template reads the same crop quad.  Templates are visited in order inside the lane because
// the orient 2/4 corner rotation of one template leaks into the next.
// finalise_kernel replaces opencvar.cpp:662-668 and 780-801 (+ cvarSquareToMatrix 524-540) for one frame per
// workgroup: lane 0 replays the order-dependent elimination, then the survivors' poses are solved one per lane.
#include "kernels.h"

namespace ocvar {

// (register budget of 4 waves per SIMD = 128 VGPRs: the binarise kernels of other contexts fill the SIMDs with 7 waves of
// 70 VGPRs, and a wave that needs 246 registers waits until most of them have drained)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 8))) void decode_kernel(Workspace ws) {
    const int f = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ws.n_squares[f] || i >= MAXQ) return;
    CandRec* out = ws.cand_recs + ((size_t)f * MAXQ + i) * MAXT;
    const int r = ws.crop_of[(size_t)f * MAXQ + i];
    unsigned long long best = ~0ull;
    if (r >= 0) best = ws.best_crop[r];
    if (best == ~0ull) {  // no quad in the crop: no candidate for this square (opencvar.cpp:704)
        for (int j = 0; j < ws.n_templates; j++) out[j].valid = 0;
        return;
    }
    const Roi roi = ws.rois_crop[r];
    const QuadRec q = ws.quads_crop[(unsigned)(best & 0xffffffffu)];
    float pat[8], pts[8];
    for (int k = 0; k < 8; k++) {
        pat[k] = (float)q.pt[k];
        pts[k] = ws.squares[((size_t)f * MAXQ + i) * 8 + k];
    }
    const uint8_t* crop = ws.gray + (size_t)roi.frame * ws.W * ws.H + (size_t)roi.y0 * ws.W + roi.x0;
    for (int j = 0; j < ws.n_templates; j++) {
        const TemplateRec t = ws.templates[j];
        const long long bit = read_code(crop, roi.w, roi.h, ws.W, pat, t.width, t.height);
        const int orient = match_orient(bit, t);
        if (orient == 4) rot_square(pts, 2);
        else if (orient == 2) rot_square(pts, 4);
        CandRec c;
        c.valid = 1;
        c.orient = orient;
        c.bit = bit;
        for (int k = 0; k < 8; k++) {
            c.square[k] = pts[k];
            c.patPoint[k] = pat[k];
        }
        out[j] = c;
    }
}

constexpr int MAXC = 2048;  // candidates per frame the tail keeps in LDS

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 8), amdgpu_num_vgpr(128))) void finalise_kernel(Workspace ws) {
    __shared__ int s_mid[MAXC];
    __shared__ int s_tid[MAXC];
    __shared__ unsigned char s_score[MAXC];
    __shared__ int s_src[MAXM];  // >= 0: candidate index, < 0: -(1 + index into prev)
    __shared__ int s_nout;
    const int f = blockIdx.x;
    const int T = ws.n_templates;
    if (threadIdx.x == 0) {
        const int nsq = ws.n_squares[f];
        int n = 0;
        bool overflow = false;
        for (int i = 0; i < nsq && i < MAXQ; i++)
            for (int j = 0; j < T; j++) {
                const CandRec* c = ws.cand_recs + ((size_t)f * MAXQ + i) * MAXT + j;
                if (!c->valid) continue;
                if (n >= MAXC) {
                    overflow = true;
                    continue;
                }
                s_mid[n] = i;
                s_tid[n] = j;
                s_score[n] = c->orient ? 1 : 0;
                n++;
            }
        if (overflow) atomicOr(ws.counters + CNT_ERR, ERR_QUAD_OVERFLOW);
        // opencvar.cpp:780-792
        for (int a = 0; a < n; a++)
            for (int b = 0; b < a; b++)
                if (s_mid[a] == s_mid[b] || s_tid[a] == s_tid[b]) {
                    if (s_score[a] > s_score[b]) s_mid[b] = -1;
                    else s_mid[a] = -1;
                }
        int nout = 0, total = 0;
        const int nr = ws.n_reserve[f];
        for (int k = 0; k < nr; k++) {  // tracked markers first (662-668)
            if (k < MAXM && nout < MAXM) s_src[nout++] = -(1 + ws.reserve[(size_t)f * MAXM + k]);
            total++;
        }
        for (int a = 0; a < n; a++)
            if (s_mid[a] >= 0) {
                if (nout < MAXM) s_src[nout++] = a;
                total++;
            }
        s_nout = nout;
        // more markers than a frame's output block holds (tracking keeps duplicates, opencvar.cpp:662-668): the next frame's
        // `prev` would be cut short and its tracking would diverge from the reference -- fail loudly instead
        if (total > MAXM) atomicOr(ws.counters + CNT_ERR, ERR_MARKER_OVERFLOW);
        ws.n_markers[f] = total;
    }
    __syncthreads();
    const CameraRec cam = *ws.camera;
    for (int k = threadIdx.x; k < s_nout; k += blockDim.x) {
        MarkerRec m;
        const int src = s_src[k];
        if (src < 0) {
            m = ws.prev[(size_t)f * MAXM + (-src - 1)];  // square already updated by the tracking step
        } else {
            const int i = s_mid[src], j = s_tid[src];
            const CandRec* c = ws.cand_recs + ((size_t)f * MAXQ + i) * MAXT + j;
            const TemplateRec t = ws.templates[j];
            m.templateId = j;
            m.markerId = i;
            m.score = c->orient ? 1.0 : 0.0;
            for (int q = 0; q < 8; q++) m.square[q] = c->square[q];
            m.aspectRatio = (double)t.width / t.height;
        }
        square_to_glmatrix(m.square, cam, m.aspectRatio, m.glMatrix);
        ws.markers[(size_t)f * MAXM + k] = m;
    }
}

void launch_decode(const Workspace& ws, hipStream_t stream) {
    if (ws.n_frames > 0) hipLaunchKernelGGL(decode_kernel, dim3((MAXQ + 63) / 64, ws.n_frames), dim3(64), 0, stream, ws);
}
void launch_finalise(const Workspace& ws, hipStream_t stream) {
    if (ws.n_frames > 0) hipLaunchKernelGGL(finalise_kernel, dim3(ws.n_frames), dim3(64), 0, stream, ws);
}

}  // namespace ocvar
