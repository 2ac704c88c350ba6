// Microbenchmark: what the frame binarise kernel's MEMORY PATTERN alone reaches on this GPU -- no filter arithmetic.
// One wave walks a strip of 64 * PX columns (PX = 4: the kernel's pattern, 768 bytes of BGR per row and wave; PX = 8: 1536 bytes)
// down CH rows: per row one load of 3 * PX bytes per lane, a PX-byte "grey" store from the strip's output lanes, and every 8 rows the
// strip's share of a 16x8-tiled mask plane as contiguous 16-byte pieces.  Prints TB/s of algorithmic bytes (5 per pixel), to be read
// against binarise_frames_kernel's 3.8 - 4.6 TB/s and a plain copy's 5.6 TB/s.   hipcc --offload-arch=gfx950 -O3 -o strip_stream strip_stream.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned v3u __attribute__((ext_vector_type(3)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

template <int PX, int AHEAD, int MODE>   // MODE bits: 1 grey stores, 2 mask stores, 4 loads without the non-temporal... (plain), 8: 16-byte aligned 4-dword loads
__global__ __launch_bounds__(64) void strip_kernel(const unsigned char* bgr, unsigned char* grey, unsigned char* mask, int W, int H, int chunk_rows, int strips, int chunks, int n_frames) {
    const int lane = threadIdx.x;
    const int unit = blockIdx.x;
    const int per_frame = strips * chunks;
    if (unit >= per_frame * n_frames) return;
    const int f = unit / per_frame, rem = unit % per_frame, chunk = rem / strips, strip = rem % strips;
    // MODE & 4: strips of 64 * PX columns side by side without halo lanes (every store of a wave covers whole, aligned 64-byte sectors)
    constexpr int HALO = (MODE & 4) ? 0 : (MODE & 16) ? 16 : (MODE & 32) ? 32 : 8;   // columns each side
    constexpr int OUTC = 64 * PX - 2 * HALO;
    const int c0 = strip * OUTC - HALO + lane * PX;   // first column of the lane
    const bool in_img = c0 >= 0 && c0 + PX <= W;
    const bool out_lane = in_img && lane * PX >= HALO && lane * PX < 64 * PX - HALO;
    const unsigned char* src = bgr + (size_t)f * W * H * 3;
    unsigned char* g = grey + (size_t)f * W * H;
    unsigned char* m = mask + (size_t)f * W * H;
    const int y0 = chunk * chunk_rows, y1 = y0 + chunk_rows < H ? y0 + chunk_rows : H;
    const int lo = y0 - 8 < 0 ? 0 : y0 - 8, hi = y1 + 8 > H ? H : y1 + 8;   // halo rows like the kernel's
    constexpr int NV = PX / 4;   // 12-byte loads per lane and row
    v3u q[AHEAD + 1][NV];
    const unsigned off = in_img ? (unsigned)c0 * 3u : 0u;
    auto fetch = [&](int y, v3u* r) {
        const unsigned char* row = src + (size_t)y * W * 3 + off;
#pragma unroll
        for (int k = 0; k < NV; k++) r[k] = *reinterpret_cast<const v3u*>(row + 12 * k);
    };
#pragma unroll
    for (int a = 0; a <= AHEAD; a++) fetch(lo + a < hi ? lo + a : hi - 1, q[a]);
    unsigned acc[8 * NV];   // 8 rows of "masks"
#pragma unroll
    for (int i = 0; i < 8 * NV; i++) acc[i] = 0;
    for (int y = lo; y < hi; y += 8) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int yy = y + r;
            unsigned gw[NV];
#pragma unroll
            for (int k = 0; k < NV; k++) gw[k] = q[0][k].x + q[0][k].y * 3u + q[0][k].z * 5u;
#pragma unroll
            for (int a = 0; a < AHEAD; a++)
#pragma unroll
                for (int k = 0; k < NV; k++) q[a][k] = q[a + 1][k];
            const int yn = yy + AHEAD + 1 < hi ? yy + AHEAD + 1 : hi - 1;
            fetch(yn, q[AHEAD]);
            if ((MODE & 1) && out_lane && yy >= y0 && yy < y1 && yy < hi) {
#pragma unroll
                for (int k = 0; k < NV; k++) { if (MODE & 8) *reinterpret_cast<unsigned*>(g + (size_t)yy * W + c0 + 4 * k) = gw[k]; else __builtin_nontemporal_store(gw[k], reinterpret_cast<unsigned*>(g + (size_t)yy * W + c0 + 4 * k)); }
            }
#pragma unroll
            for (int k = 0; k < NV; k++) acc[r * NV + k] = gw[k] ^ (gw[k] >> 3);
        }
        // mask rows y .. y+7 of this lane's PX columns: 8 * PX bytes; stored as 16-byte pieces, contiguous across lanes per instruction
        if (!(MODE & 3)) { unsigned z = 0; for (int i = 0; i < 8 * NV; i++) z |= acc[i]; if (z == 0x12345u) g[0] = 1; }
        if ((MODE & 2) && out_lane && y >= y0 && y + 8 <= y1) {
            unsigned char* base = m + ((size_t)(y >> 3) * (W >> 4) << 7) + ((size_t)(c0 >> 4) << 7) + (size_t)(c0 & 15) * 8;
#pragma unroll
            for (int k = 0; k < (8 * NV) / 4; k++)
                __builtin_nontemporal_store((v4u){acc[4 * k], acc[4 * k + 1], acc[4 * k + 2], acc[4 * k + 3]}, reinterpret_cast<v4u*>(base + 16 * k));
        }
    }
}

template <int PX, int AHEAD, int MODE>
static double run(const unsigned char* bgr, unsigned char* grey, unsigned char* mask, int W, int H, int chunk_rows, int n) {
    constexpr int HALO = (MODE & 4) ? 0 : (MODE & 16) ? 16 : (MODE & 32) ? 32 : 8;
    constexpr int OUTC = 64 * PX - 2 * HALO;
    const int strips = (W + OUTC - 1) / OUTC, chunks = (H + chunk_rows - 1) / chunk_rows;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    double best = 1e9;
    for (int rep = 0; rep < 6; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((strip_kernel<PX, AHEAD, MODE>), dim3(strips * chunks * n), dim3(64), 0, 0, bgr, grey, mask, W, H, chunk_rows, strips, chunks, n);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    return best;
}

int main(int argc, char** argv) {
    const int W = 1920, H = 1080, n = argc > 1 ? atoi(argv[1]) : 1024;
    unsigned char *bgr, *grey, *mask;
    hipMalloc(&bgr, (size_t)n * W * H * 3); hipMalloc(&grey, (size_t)n * W * H); hipMalloc(&mask, (size_t)n * W * H + (1 << 20));
    hipMemset(bgr, 7, (size_t)n * W * H * 3);
    const double bytes = 5.0 * W * H * n;
    const int chunk_rows = 544;
    double t;
    t = run<4, 2, 0>(bgr, grey, mask, W, H, chunk_rows, n); printf("reads only            : %7.3f ms  %.2f TB/s of reads (3 B/px)\n", t, 3.0 * W * H * n / t / 1e9);
    t = run<4, 2, 1>(bgr, grey, mask, W, H, chunk_rows, n); printf("reads + grey stores   : %7.3f ms  %.2f TB/s (4 B/px)\n", t, 4.0 * W * H * n / t / 1e9);
    t = run<4, 2, 2>(bgr, grey, mask, W, H, chunk_rows, n); printf("reads + mask stores   : %7.3f ms  %.2f TB/s (4 B/px)\n", t, 4.0 * W * H * n / t / 1e9);
    t = run<4, 2, 3>(bgr, grey, mask, W, H, chunk_rows, n); printf("reads + both          : %7.3f ms  %.2f TB/s (5 B/px)\n", t, bytes / t / 1e9);
    t = run<4, 2, 5>(bgr, grey, mask, W, H, chunk_rows, n); printf("aligned strips: reads + grey : %7.3f ms  %.2f TB/s (4 B/px)\n", t, 4.0 * W * H * n / t / 1e9);
    t = run<4, 2, 6>(bgr, grey, mask, W, H, chunk_rows, n); printf("aligned strips: reads + mask : %7.3f ms  %.2f TB/s (4 B/px)\n", t, 4.0 * W * H * n / t / 1e9);
    t = run<4, 2, 7>(bgr, grey, mask, W, H, chunk_rows, n); printf("aligned strips: reads + both : %7.3f ms  %.2f TB/s (5 B/px)\n", t, bytes / t / 1e9);
    t = run<4, 2, 9>(bgr, grey, mask, W, H, chunk_rows, n); printf("240-col strips, PLAIN grey stores (no nt): reads + grey : %7.3f ms\n", t);
    t = run<4, 2, 11>(bgr, grey, mask, W, H, chunk_rows, n); printf("240-col strips, PLAIN grey stores: reads + both : %7.3f ms  %.2f TB/s\n", t, bytes / t / 1e9);
    t = run<4, 2, 19>(bgr, grey, mask, W, H, chunk_rows, n); printf("224-col strips (32-byte aligned), nt: reads + both : %7.3f ms  %.2f TB/s\n", t, bytes / t / 1e9);
    t = run<4, 2, 35>(bgr, grey, mask, W, H, chunk_rows, n); printf("192-col strips (64-byte aligned), nt: reads + both : %7.3f ms  %.2f TB/s\n", t, bytes / t / 1e9);
    t = run<8, 2, 0>(bgr, grey, mask, W, H, chunk_rows, n); printf("8 px: reads only      : %7.3f ms  %.2f TB/s of reads\n", t, 3.0 * W * H * n / t / 1e9);
    t = run<8, 2, 3>(bgr, grey, mask, W, H, chunk_rows, n); printf("8 px: reads + both    : %7.3f ms  %.2f TB/s (5 B/px)\n", t, bytes / t / 1e9);
    return 0;
}
