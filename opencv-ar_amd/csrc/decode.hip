// decode.hip -- per-quad code readout and the per-frame tail (dedupe, pose, marker records) on gfx950.
//
// decode_kernel replaces the inner template loop of cvarArMultRegistration
// (/root/reference/src/opencvar.cpp:700-774) for one frame-pass quad per wave.  The reference re-runs the whole
// square finder on the crop once per template with an identical result; here the crop pass ran
// once and every template reads the same crop quad.  Templates are visited in order inside the lane because
// the orient 2/4 corner rotation of one template leaks into the next.
// finalise_kernel replaces opencvar.cpp:662-668 and 780-801 for one frame per workgroup (the order-dependent elimination,
// replayed 64 comparisons at a time, and the marker records); pose_kernel solves the survivors' poses (cvarSquareToMatrix
// 524-540), one lane per marker over the whole batch.
#include "kernels.h"

namespace ocvar {

// One wave per (frame, quarter of the frame's quads): workgroups of 64 threads, because with several contexts in flight a
// workgroup has to fit into the gap one retiring binarise wave leaves (round 2's 256-thread workgroups at 128 registers took 5 ms
// in-region for 0.3 ms of work).  Two phases per chunk of 8 quads.  Phase 1, one lane per (quad, template): the inverse
// homography (closed form in double, rounded to float32 and inverted; the destination quad is (tw+2) x (th+2), so it depends on
// the template) into LDS.  Phase 2, quad by quad, one lane per code cell: lane L samples the cell whose bit is bit L of the
// code -- acArray2DToBit packs row-major, columns right to left, first cell in the most significant bit (acmath.cpp:546-554)
// -- so the code is the ballot of the thresholded samples.
constexpr int DECODE_SLICES = 4;   // waves per frame: quad i belongs to slice i % 4
constexpr int DECODE_CHUNK = 8;    // quads of a slice per phase-1 round (x templates <= 64 lanes for up to 8 templates; more: lanes loop)

__global__ __launch_bounds__(64) void decode_kernel(Workspace ws) {
    extern __shared__ double sM_dyn[];   // [DECODE_CHUNK][n_templates][9]
    __shared__ int s_roi[DECODE_CHUNK];        // crop ROI of the chunk's quad, -1: no quad in its crop
    __shared__ unsigned s_slot[DECODE_CHUNK];  // quads_crop slot of the crop's quad
    const int f = blockIdx.x / DECODE_SLICES, slice = blockIdx.x % DECODE_SLICES;
    const int lane = threadIdx.x;
    const int T = ws.n_templates;
    int nsq = ws.n_squares[f];
    nsq = nsq < ws.maxq ? nsq : ws.maxq;
    const int mine = (nsq - slice + DECODE_SLICES - 1) / DECODE_SLICES;   // quads slice, slice + 4, ... < nsq
    for (int base = 0; base < mine; base += DECODE_CHUNK) {
        const int cnt = mine - base < DECODE_CHUNK ? mine - base : DECODE_CHUNK;
        if (lane < cnt) {
            const int i = slice + DECODE_SLICES * (base + lane);
            const int r = ws.crop_of[(size_t)f * ws.maxq + i];
            unsigned long long best = ~0ull;
            if (r >= 0) best = ws.best_crop[r];
            s_roi[lane] = best == ~0ull ? -1 : r;
            s_slot[lane] = (unsigned)(best & 0xffffffffu);
        }
        __syncthreads();
        for (int pair = lane; pair < cnt * T; pair += 64) {
            const int qi = pair / T, j = pair - qi * T;
            if (s_roi[qi] < 0) continue;
            const QuadRec q = ws.quads_crop[s_slot[qi]];
            float pat[8], m32[9];
            for (int k = 0; k < 8; k++) pat[k] = (float)q.pt[k];
            const TemplateRec t = ws.templates[j];
            if (!perspective_from_quad(pat, t.width + 2, t.height + 2, m32))
                for (int k = 0; k < 9; k++) m32[k] = 0.f;
            double M[9];
            invert_map(m32, M);
            for (int k = 0; k < 9; k++) sM_dyn[(qi * T + j) * 9 + k] = M[k];
        }
        __syncthreads();
        for (int qi = 0; qi < cnt; qi++) {
            const int i = slice + DECODE_SLICES * (base + qi);
            CandRec* out = ws.cand_recs + ((size_t)f * ws.maxq + i) * MAXT;
            const int r = s_roi[qi];
            if (r < 0) {  // no quad in the crop: no candidate for this square (opencvar.cpp:704)
                if (lane < T) out[lane].valid = 0;
                continue;
            }
            const Roi roi = ws.rois_crop[r];
            const QuadRec q = ws.quads_crop[s_slot[qi]];
            float pat[8], pts[8];
            for (int k = 0; k < 8; k++) {
                pat[k] = (float)q.pt[k];
                pts[k] = ws.squares[((size_t)f * ws.maxq + i) * 8 + k];
            }
            const uint8_t* plane = ws.gray + (size_t)roi.frame * gray_plane_bytes(ws.W, ws.H);
            const int pitch = gray_pitch(ws.W);
            auto crop_px = [=](int ix, int iy) -> int { return plane[(size_t)(roi.y0 + iy) * pitch + gray_col(roi.x0 + ix)]; };
            for (int j = 0; j < T; j++) {   // in order: the orient 2/4 corner rotation leaks into the next template (D4)
                const TemplateRec t = ws.templates[j];
                double M[9];
                for (int k = 0; k < 9; k++) M[k] = sM_dyn[(qi * T + j) * 9 + k];
                const int n = t.width * t.height;     // <= 64 (ocvar_hip_set_templates)
                bool v = false;
                if (lane < n) {
                    const int p = n - 1 - lane;       // position in acArray2DToBit's scan: row p / tw, column tw-1 - p % tw
                    const int ci = p / t.width, cj = t.width - 1 - p % t.width;
                    int cx, cy;
                    if (code_cell(ci * t.width + cj, t.width, t.height, &cx, &cy))
                        v = warp_sample_px(crop_px, roi.w, roi.h, M, cx + 1, cy + 1) > 100;
                }
                const long long bit = (long long)__ballot(v);
                const int orient = match_orient(bit, t);
                if (orient == 4) rot_square(pts, 2);
                else if (orient == 2) rot_square(pts, 4);
                if (lane == 0) {
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
        }
        __syncthreads();
    }
}

// (the tail keeps a frame's candidates in LDS: Workspace::maxc of them, 9 bytes each, sized at launch)

__global__ __launch_bounds__(64) void finalise_kernel(Workspace ws) {
    extern __shared__ int tail_lds[];
    const int MAXC = ws.maxc;
    int* s_mid = tail_lds;
    int* s_tid = tail_lds + MAXC;
    unsigned char* s_score = reinterpret_cast<unsigned char*>(tail_lds + 2 * MAXC);
    __shared__ int s_src[MAXM];  // >= 0: candidate index, < 0: -(1 + index into prev)
    __shared__ int s_nout, s_job;
    const int f = blockIdx.x;
    const int T = ws.n_templates;
    const int lane = threadIdx.x;   // one wave per frame
    // candidates in the reference's order (square-major, template-minor), compacted 64 slots at a time
    int n = 0;
    {
        int nsq = ws.n_squares[f];
        nsq = nsq < ws.maxq ? nsq : ws.maxq;
        const int slots = nsq * T;
        bool overflow = false;
        for (int base = 0; base < slots; base += 64) {
            const int s = base + lane;
            const int i = s / T, j = s - i * T;
            int orient = 0;
            bool valid = false;
            if (s < slots) {
                const CandRec* c = ws.cand_recs + ((size_t)f * ws.maxq + i) * MAXT + j;
                valid = c->valid != 0;
                orient = c->orient;
            }
            const unsigned long long m = __ballot(valid);
            const int at = n + __popcll(m & ((1ull << lane) - 1ull));
            if (valid) {
                if (at < MAXC) {
                    s_mid[at] = i;
                    s_tid[at] = j;
                    s_score[at] = orient ? 1 : 0;
                } else {
                    overflow = true;
                }
            }
            n += __popcll(m);
        }
        if (__ballot(overflow) && lane == 0) atomicOr(ws.counters + CNT_ERR, ERR_QUAD_OVERFLOW);
        n = n < MAXC ? n : MAXC;
    }
    __syncthreads();
    // opencvar.cpp:780-792, the order-dependent `||` elimination: for a, for b < a: if same square or same template, the
    // lower score (ties: a) gets markerId -1.  The inner loop is replayed 64 values of b at a time: until a itself is
    // eliminated -- at the first b it loses to -- every matching b with a lower score is eliminated; from there on
    // markerId[a] is -1, which still "matches" eliminated b's (a no-op) and same-template b's.  Same result as the
    // sequential loop, entry for entry.
    for (int a = 1; a < n; a++) {
        const int ma = s_mid[a], ta = s_tid[a];
        const int sa = s_score[a];
        bool dead = false;   // wave-uniform: markerId[a] has become -1 inside this inner loop
        for (int base = 0; base < a; base += 64) {
            const int b = base + lane;
            const bool in = b < a;
            const int mb = in ? s_mid[b] : -2, tb = in ? s_tid[b] : -2, sb = in ? (int)s_score[b] : 0;
            const bool pre = in && (ma == mb || ta == tb);        // match while markerId[a] is still ma
            const bool post = in && (mb == -1 || ta == tb);       // match once markerId[a] is -1
            const bool wins = sa > sb;
            unsigned long long lose = dead ? 0ull : __ballot(pre && !wins);
            const int bstar = lose ? __ffsll((long long)lose) - 1 : 64;
            const bool kill = wins && (dead ? post : (lane < bstar ? pre : (lane > bstar ? post : false)));
            if (kill) s_mid[b] = -1;
            dead = dead || lose != 0;
        }
        if (dead && lane == 0) s_mid[a] = -1;
        __syncthreads();
    }
    if (lane == 0) {
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
        s_job = nout > 0 ? atomicAdd(ws.counters + CNT_POSE_JOBS, nout) : 0;   // one list append per frame
    }
    __syncthreads();
    // the surviving markers' records, all but the pose; one pose job per marker for pose_kernel
    for (int k = threadIdx.x; k < s_nout; k += blockDim.x) {
        MarkerRec* m = ws.markers + (size_t)f * MAXM + k;
        const int src = s_src[k];
        if (src < 0) {
            *m = ws.prev[(size_t)f * MAXM + (-src - 1)];  // square already updated by the tracking step
        } else {
            const int i = s_mid[src], j = s_tid[src];
            const CandRec* c = ws.cand_recs + ((size_t)f * ws.maxq + i) * MAXT + j;
            const TemplateRec t = ws.templates[j];
            m->templateId = j;
            m->markerId = i;
            m->score = c->orient ? 1.0 : 0.0;
            for (int q = 0; q < 8; q++) m->square[q] = c->square[q];
            m->aspectRatio = (double)t.width / t.height;
        }
        ws.pose_jobs[s_job + k] = f * MAXM + k;
    }
}

// cvarSquareToMatrix (opencvar.cpp:524-540) for every surviving marker of the batch, one lane each.  Round 2 solved the poses
// inside the per-frame tail kernel: one 64-lane workgroup per frame with at most 3 lanes in the solver, capped at 128 registers so
// that it could start between binarise waves -- the solver's arrays lived in scratch memory and the kernel took 1.06 ms per
// 2048 frames for ~5 k poses.  Here the ~5 k jobs fill ~85 waves, every array of the solver is a register (loops unrolled to
// constant indices, the seeding homography in closed form: pose_core.h) and the kernel is as long as one solve.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void pose_kernel(Workspace ws) {
    int n = ws.counters[CNT_POSE_JOBS];
    const int cap = ws.n_frames * MAXM;
    if (n > cap) n = cap;
    const CameraRec cam = *ws.camera;
    for (int j = blockIdx.x * 64 + threadIdx.x; j < n; j += gridDim.x * 64) {
        MarkerRec* m = ws.markers + ws.pose_jobs[j];
        float sq[8];
        for (int q = 0; q < 8; q++) sq[q] = m->square[q];
        double gl[16];
        square_to_glmatrix(sq, cam, m->aspectRatio, gl);
        for (int q = 0; q < 16; q++) m->glMatrix[q] = gl[q];
    }
}

void launch_decode(const Workspace& ws, hipStream_t stream) {
    if (ws.n_frames > 0)
        hipLaunchKernelGGL(decode_kernel, dim3(ws.n_frames * DECODE_SLICES), dim3(64), (size_t)DECODE_CHUNK * ws.n_templates * 9 * sizeof(double), stream, ws);
}
void launch_finalise(const Workspace& ws, hipStream_t stream) {
    if (ws.n_frames <= 0) return;
    hipLaunchKernelGGL(finalise_kernel, dim3(ws.n_frames), dim3(64), (size_t)ws.maxc * 9 + 16, stream, ws);
    // a stateless frame keeps at most one marker per template: a grid of one wave per 8 frames takes a batch's poses in one
    // pass; stateful batches with many tracked markers per frame loop (grid-stride)
    const int blocks = ws.n_frames / 8 + 1;
    hipLaunchKernelGGL(pose_kernel, dim3(blocks), dim3(64), 0, stream, ws);
}

}  // namespace ocvar
