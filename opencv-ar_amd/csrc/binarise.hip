// binarise.hip -- fused grey + pyramid denoise + adaptive threshold + neighbour-mask kernel (gfx950).
//
// Replaces the image stages of cvarFindSquares (/root/reference/src/opencvar.cpp:156-184): cvCloneImage,
// cvPyrDown(5x5), cvPyrUp, cvCvtColor(BGR2GRAY), cvAdaptiveThreshold(GAUSSIAN_C, 7x7, delta 8) -- in frame mode
// also the BGR2GRAY of cvarArMultRegistration (opencvar.cpp:624-627) -- and the local part of cvFindContours
// (opencvar.cpp:183-184): zeroing the 1-px frame and spotting where a border can begin.
//
// "Wave march": one 64-lane wavefront (a workgroup of its own) owns a strip of 256 columns (4 pixels per lane: 240 output
// columns + an 8-column halo on each side) and walks down a chunk of rows.  Everything vertical lives in registers and nothing
// is a window that is shifted: the vertical pyrDown is two accumulators, the vertical pyrUp two running sums (times four, so a
// finished pixel is the high byte of its sum), the last 8 pyrUp rows a ring of bytes in two registers per pixel that swap names
// every four rows, the table outputs of the last two threshold rows three registers; everything horizontal is a 4-pixel packed
// word handed to the neighbour lane with a DPP wave shift.  One (unaligned) buffer load per lane and row, reflected columns
// included, two rows under way beyond the current one; rows are addressed through buffer resources (scalar row offset,
// loop-invariant lane offset, stores masked by range).  The image itself never goes through LDS, which only holds a 128-entry
// table (threshold-bit window -> neighbour-mask contributions and border-start nibbles), 8 mask rows per wave on their way to
// whole 16x8 tiles, and the staged border starts.  The grey plane is stored in 256-byte panels, one per strip (hd.h::gray_col):
// every store of the kernel is whole 64-byte sectors.  The row body exists in several instances (steady rows in groups of four
// with the ring's position known, strips away from the image edges) so that the hot loop carries no range tests; the filters'
// border rules at the image edges are per-lane byte selectors.
// HBM traffic per pixel: 3 B read + 1 B grey + 1 B mask (frame mode).
// The arithmetic is the integer arithmetic of the definition, so the output is bit-identical:
//   pyrDown  [1 4 6 4 1]^2, (v+128)>>8, BORDER_REFLECT_101        pyrUp  [1 6 1]/[4 4], (v+32)>>6, borders -1->1, n->n-1
//   Gaussian [8 28 56 72 56 28 8]^2, (v+32768)>>16, BORDER_REPLICATE    threshold  src - mean > -8
#include "kernels.h"
#include <cstdlib>
#include <type_traits>

#ifndef OCVAR_WAVES_F
#define OCVAR_WAVES_F 5   // waves per SIMD the frame kernel is compiled for (register budget 512 / n, in eights)
#endif
#ifndef OCVAR_ROWS_AHEAD_F
#define OCVAR_ROWS_AHEAD_F 2   // source rows a wave of the frame kernel has under way beyond the one it works on (1 .. 3)
#endif
#ifndef OCVAR_BIN_WG_WAVES
#define OCVAR_BIN_WG_WAVES 1   // waves per workgroup of the two binarise kernels (1: a wave can take any SIMD with room for it)
#endif
#ifndef OCVAR_ROWS_AHEAD_C
#define OCVAR_ROWS_AHEAD_C 1   // ... of the crop kernel
#endif
#ifndef OCVAR_WAVES_C
#define OCVAR_WAVES_C 5   // ... and the crop kernel
#endif

namespace ocvar {

constexpr int SV = MARCH_STRIP;  // output columns per strip
// Halo: an output pixel's mask depends on source pixels up to 9 columns to its left and 10 to its right (mask 1 (+1 for
// the start test), 7-tap Gaussian 3, pyrUp 3, pyrDown 2), but the stages exchange whole lanes: following the chain lane
// by lane, the first output lane needs the grey of 2 lanes to its left and the last one of 2 lanes to its right -- a third
// lane on the right would only serve the start filter's "row above at x+2" bit of the strip's last column, which is
// dropped instead (the filter is a necessary condition: without it that column merely yields a few more plausible starts).
// 240 output columns: 8 strips across 1920, 16 across 3840.
constexpr int HL = MARCH_HALO_L, HR = MARCH_HALO_R;

// A value that is the same in all lanes of the wave, moved to an SGPR.  The work-unit index comes from threadIdx.x >> 6,
// which the compiler must treat as per-lane: without this every row counter of the march lives in a VGPR, every loop
// test is a VALU compare + EXEC mask, and every row address is 64-bit VALU arithmetic.
__device__ __forceinline__ int wave_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
// a wave-uniform 64-bit offset, pinned to an SGPR pair (row offsets: the per-lane part of an address is then a 32-bit add)
__device__ __forceinline__ long long wave_uniform64(long long v) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)v >> 32));
    return (long long)(((unsigned long long)hi << 32) | lo);
}

// buffer resource over [p, p + bytes): raw (stride 0), 32-bit data format -- the addressing mode of the row loads and stores
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
constexpr int BUF_NT = 2;   // cache policy "non-temporal" of the buffer builtins

__device__ __forceinline__ int reflect101(int p, int len) {
    if (len == 1) return 0;
    while ((unsigned)p >= (unsigned)len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}

__device__ __forceinline__ unsigned grey_of(unsigned b, unsigned g, unsigned r) { return (b * 1868u + g * 9617u + r * 4899u + 8192u) >> 14; }

// lane i receives lane i-1 / lane i+1 (DPP whole-wave shifts: one VALU move, no LDS traffic)
__device__ __forceinline__ unsigned up1(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, true); }
__device__ __forceinline__ unsigned down1(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, true); }

typedef unsigned short us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ us2 as_us2(unsigned v) { return __builtin_bit_cast(us2, v); }
__device__ __forceinline__ unsigned as_u32(us2 v) { return __builtin_bit_cast(unsigned, v); }
__device__ __forceinline__ unsigned dot4(unsigned a, unsigned b, unsigned c) { return __builtin_amdgcn_udot4(a, b, c, false); }
__device__ __forceinline__ unsigned dot2(us2 a, unsigned k, unsigned c) { return __builtin_amdgcn_udot2(a, as_us2(k), c, false); }
constexpr unsigned K2(unsigned lo, unsigned hi) { return lo | (hi << 16); }
__device__ __forceinline__ unsigned alignb(unsigned hi, unsigned lo, unsigned sh) { return __builtin_amdgcn_alignbyte(hi, lo, sh); }
__device__ __forceinline__ unsigned byte_of(unsigned v, int j) { return (v >> (8 * j)) & 255u; }
__device__ __forceinline__ unsigned long long rotl64(unsigned long long v, unsigned s) { return s ? (v << s) | (v >> (64u - s)) : v; }
// nib = 2 * nib + (floor(S / 65536) < byte B of w): the threshold test of one pixel, the operands picked by the instruction itself
// (SDWA: the high word of S sign-extended, one byte of w) -- one compare and one add-with-carry, no operand to prepare.
template <int B>
__device__ __forceinline__ void push_bit(unsigned& nib, unsigned S, unsigned w) {
    static_assert(B >= 0 && B < 4, "byte of a dword");
    if constexpr (B == 0) asm("v_cmp_lt_i32_sdwa vcc, sext(%1), %2 src0_sel:WORD_1 src1_sel:BYTE_0\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(nib) : "v"(S), "v"(w) : "vcc");
    if constexpr (B == 1) asm("v_cmp_lt_i32_sdwa vcc, sext(%1), %2 src0_sel:WORD_1 src1_sel:BYTE_1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(nib) : "v"(S), "v"(w) : "vcc");
    if constexpr (B == 2) asm("v_cmp_lt_i32_sdwa vcc, sext(%1), %2 src0_sel:WORD_1 src1_sel:BYTE_2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(nib) : "v"(S), "v"(w) : "vcc");
    if constexpr (B == 3) asm("v_cmp_lt_i32_sdwa vcc, sext(%1), %2 src0_sel:WORD_1 src1_sel:BYTE_3\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(nib) : "v"(S), "v"(w) : "vcc");
}

// nib = (floor(S / 65536) < byte B of w): the first pixel's test starts the nibble (no register to clear)
template <int B>
__device__ __forceinline__ void first_bit(unsigned& nib, unsigned S, unsigned w) {
    static_assert(B >= 0 && B < 4, "byte of a dword");
    if constexpr (B == 0) asm("v_cmp_lt_i32_sdwa vcc, sext(%1), %2 src0_sel:WORD_1 src1_sel:BYTE_0\n\tv_cndmask_b32_e64 %0, 0, 1, vcc" : "=v"(nib) : "v"(S), "v"(w) : "vcc");
    if constexpr (B == 1) asm("v_cmp_lt_i32_sdwa vcc, sext(%1), %2 src0_sel:WORD_1 src1_sel:BYTE_1\n\tv_cndmask_b32_e64 %0, 0, 1, vcc" : "=v"(nib) : "v"(S), "v"(w) : "vcc");
    if constexpr (B == 2) asm("v_cmp_lt_i32_sdwa vcc, sext(%1), %2 src0_sel:WORD_1 src1_sel:BYTE_2\n\tv_cndmask_b32_e64 %0, 0, 1, vcc" : "=v"(nib) : "v"(S), "v"(w) : "vcc");
    if constexpr (B == 3) asm("v_cmp_lt_i32_sdwa vcc, sext(%1), %2 src0_sel:WORD_1 src1_sel:BYTE_3\n\tv_cndmask_b32_e64 %0, 0, 1, vcc" : "=v"(nib) : "v"(S), "v"(w) : "vcc");
}

// ---- neighbour masks and border starts by table ---------------------------------------------------------------------
// A lane's 4 threshold bits of one row plus the bits next to them form a 7-bit window: bit 0 = column c0-1 (the left
// lane's pixel 3), bits 1..4 = the lane's pixels 0..3, bits 5, 6 = columns c0+4, c0+5 (the right lane's pixels 0, 1).
// What a row contributes to the 8-neighbour masks of the 4 pixels -- as the row above (NE N NW), the pixels' own row
// (E W) and the row below (SW S SE) -- and to the border-start tests depends on that window only, so it is read from a
// 128-entry table in LDS (one ds_read_b128 per lane and row) instead of being spread out with multiplications:
//   x: mask bytes of the 4 pixels contributed as the row above      y: ... as their own row      z: ... as the row below
//   w: start nibbles (bit p = pixel p).  Low half, the row as the pixels' own row:  [0] centre & ~W   [1] E   [2] W & ~centre
//      [3] ~E.  High half, the row as the row above:  [4] ~(NW | N | NE)   [5] the pixel above E's east neighbour
//      [6] N   [7] ~NE.  With t = low(own row) & high(row above), t & ~(t >> 4) has the outer starts in nibble 0
//      (centre & ~W & ~NW & ~N & ~NE & ~(E & NEE)) and the hole starts in nibble 2 (W & ~centre & N & (E | NE)):
//      the necessary local conditions for being the raster-first pixel of a region (see march_unit).
__device__ __forceinline__ uint4 mask_table_entry(unsigned w) {
    uint4 t = make_uint4(0u, 0u, 0u, 0u);
    unsigned mA = 0, mB = 0, mC = 0, uA = 0, uB = 0, uC = 0, uD = 0;
    for (unsigned p = 0; p < 4; p++) {
        const unsigned xm = (w >> p) & 1u, x0 = (w >> (p + 1)) & 1u, xp = (w >> (p + 2)) & 1u, xpp = (w >> (p + 3)) & 1u;
        t.x |= ((xp << 1) | (x0 << 2) | (xm << 3)) << (8 * p);   // NE N NW
        t.y |= (xp | (xm << 4)) << (8 * p);                      // E W
        t.z |= ((xm << 5) | (x0 << 6) | (xp << 7)) << (8 * p);   // SW S SE
        mA |= (x0 & ~xm & 1u) << p;
        mB |= xp << p;
        mC |= (xm & ~x0 & 1u) << p;
        uA |= (~(xm | x0 | xp) & 1u) << p;
        uB |= xpp << p;
        uC |= x0 << p;
        uD |= xp << p;
    }
    t.w = mA | (mB << 4) | (mC << 8) | ((~mB & 15u) << 12) | (uA << 16) | (uB << 20) | (uC << 24) | ((~uD & 15u) << 28);
    return t;
}

// table index (march_unit: bits 0..3 own pixels, bit 4 the left lane's pixel 3, bits 5, 6 the right lane's pixels 0, 1) -> window
__device__ __forceinline__ unsigned table_window(unsigned i) { return ((i >> 4) & 1u) | ((i & 15u) << 1) | ((i >> 5) << 5); }

struct MarchOut {
    uint8_t* gray;          // frame mode: grey plane of this frame (stride gray_stride), else null
    long long gray_stride;
    uint8_t* nbr;           // neighbour-mask plane of this ROI, stride ns
    int ns;
    int roi;
    StartCand* cands;
    int* n_cands;
    int cap_cands;
    int* err;
};

// One work unit: strip `strip` of an (sw x sh) ROI, output rows [Y0, Y1).
// EDGE = false: the strip lies strictly inside the image (no lane holds a column < 0 or >= sw), so the border rules of the
// filters and their lane masks are compiled out -- six of a 1080p frame's eight strips; a crop's strips always touch an edge.
template <bool BGR, bool EDGE>
__device__ void march_unit(const uint8_t* src, long long src_stride, int x_org /* crops: the ROI's first column in the frame's grey plane */, int sw, int sh, int strip, int Y0, int Y1, const MarchOut& o,
                           unsigned* stage, unsigned* rowbuf /* LDS, 8 rows x 64 lanes */, const uint4* tab /* LDS, mask_table_entry */) {
    const int lane = threadIdx.x & 63;
    const int XS = strip * SV - 4 * HL;
    const int c0 = XS + 4 * lane;
    const int pw = sw >> 1, ph_ = sh >> 1;
    const int k0 = c0 >> 1;  // pyramid column of the lane's first pyramid sample (c0 is a multiple of 4)
    const bool out_lane = lane >= HL && lane < 64 - HR && (!EDGE || c0 < sw);
    const bool needed = !EDGE || (c0 + 3 >= -16 && c0 < sw + 16);  // beyond every halo: never consumed
    const bool fast = !EDGE || (c0 >= 0 && c0 + 3 < sw);
    const bool aligned = BGR ? ((src_stride & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 3) == 0) : true;
    // (wave-uniform flags as scalars: left as they are the compiler keeps them as lane masks and re-derives their negations with
    // vector instructions in every row)
    // The filters' border rules at the image's left and right edge are byte selections in the one or two lanes that hold the edge
    // columns; every lane carries its selectors (the identity elsewhere), so the row body needs no wave-uniform "this strip touches
    // an edge" tests -- as flags those cost scalar registers the kernel does not have, and a handful of vector instructions per row
    // to re-derive their negations.  (v_perm_b32: bytes 0..3 = the second operand, 4..7 = the first.)
    constexpr unsigned SEL_ID = 0x03020100u;
    // pyrUp's right border, "pyramid column pw stands for column pw-1", on the lane's pair (P[k0], P[k0+1]) with the left lane's pair as first operand
    const unsigned selP = !EDGE ? SEL_ID : k0 + 1 == pw ? 0x01000100u : k0 == pw ? 0x03020706u : SEL_ID;
    // BORDER_REPLICATE of the 7x7 Gaussian, applied to the vertical sums (a replicated column's sums are the edge column's): the lane
    // that holds column 0 takes (V0, V0) for what it would receive from its left neighbour; the lane that holds column sw-1 (its
    // pixel 1 or 3: sw is even) first overwrites its own pair (V2, V3) when those lie beyond the edge, then takes its last real
    // column for what it would receive from the right.
    const unsigned selL = EDGE && c0 == 0 ? 0x05040504u : SEL_ID;
    const unsigned selB = EDGE && c0 + 2 == sw ? 0x07060706u : SEL_ID;
    const unsigned selR = EDGE && c0 < sw && c0 + 4 >= sw ? 0x07060706u : SEL_ID;
    // byte offset of source column c (0 <= c < sw) in its row: 3 c in a BGR frame; in the panelled grey plane (hd.h::gray_col) the
    // crop's column c is the frame's column x_org + c (up to 9 consecutive columns follow contiguously from any column)
    auto col_off = [&](int c) -> unsigned { return BGR ? (unsigned)(3 * c) : gray_col(x_org + c); };
    // reflected source columns of the lanes that straddle an image edge (BORDER_REFLECT_101)
    unsigned xr0 = 0, xr1 = 0, xr2 = 0, xr3 = 0;
    if (needed && !fast) {
        xr0 = col_off(reflect101(c0, sw));
        xr1 = col_off(reflect101(c0 + 1, sw));
        xr2 = col_off(reflect101(c0 + 2, sw));
        xr3 = col_off(reflect101(c0 + 3, sw));
    }
    // Load plan: every lane loads its 4 pixels with ONE (unaligned) load per row -- 4 bytes grey, 12 bytes BGR -- also the
    // lanes that hold reflected columns, whose 4 (grey) bytes are then permuted.  With BORDER_REFLECT_101 a lane left of
    // column 0 holds columns -c0 .. -c0-3, a lane right of column sw-1 holds 2sw-2-c0 .. 2sw-5-c0: four contiguous source
    // pixels in reverse order; the lane that straddles the right edge (sw - c0 == 2: sw is even, c0 a multiple of 4) holds
    // c0, c0+1, c0, c0-1 and loads c0-2 .. c0+1.  (The byte loads of the generic path sit in a divergent branch, which makes
    // every row wait for its own loads instead of running a row ahead: a crop has edge lanes on both sides of every work
    // unit, and its binarise kernel was latency-bound by that.)  No byte outside columns 0 .. sw-1 of the row is read.
    const bool plan = sw >= 32;   // single reflection, every offset inside the row
    unsigned goff = col_off(0), gsel = 0x03020100u;
    if (plan && needed) {   // (first column of the lane's load)
        if (c0 + 3 < 0) { goff = col_off(-c0 - 3); gsel = 0x00010203u; }
        else if (c0 >= sw) { goff = col_off(2 * sw - 5 - c0); gsel = 0x00010203u; }
        else if (c0 + 3 >= sw) { goff = col_off(c0 - 2); gsel = 0x01020302u; }
        else goff = col_off(c0);
    }
    // per-lane byte offsets (unsigned: the row bases are wave-uniform, so loads/stores can use the SGPR-base + 32-bit VGPR offset form)
    const unsigned src_off = fast ? col_off(c0) : col_off(0);
    // Rows are addressed through buffer resources: base in four SGPRs, the row's byte offset in one SGPR (one s_mul per row), the
    // lane's offset in a loop-invariant VGPR -- no 64-bit address arithmetic, no per-lane pointers to keep or spill.  A lane that
    // must not store gets an offset beyond the resource's size: the hardware drops its store, so the grey and mask stores need
    // no EXEC mask and the row body stays one basic block.
    constexpr unsigned OOB = 0x80000000u;   // >= every resource size below (+ 16 does not wrap)
    const __amdgpu_buffer_rsrc_t src_rs = make_rsrc(src, 0x7fffffffu);
    const __amdgpu_buffer_rsrc_t gray_rs = make_rsrc(o.gray, o.gray ? (unsigned)(o.gray_stride * sh) : 0u);
    const __amdgpu_buffer_rsrc_t nbr_rs = make_rsrc(o.nbr, (unsigned)nbr_plane_bytes(o.ns, sh));
    // the grey plane's panel of this strip (hd.h::gray_col): all 64 lanes store, 256 contiguous bytes at a 256-byte boundary -- the
    // halo lanes' bytes are the neighbouring strips' columns again (same values), or columns outside the image (never read)
    const unsigned out_off = (unsigned)(strip * GRAY_PANEL_BYTES + 4 * lane);
    const unsigned rmask4 = lane == 63 - HR ? 0x10u : 0x30u;   // the last output lane's x+2 bit (row above) is not computed
    unsigned bit7;   // (a constant in a register, opaque to the compiler: then the AND below folds into the DPP move -- DPP operands cannot be literals)
    asm("v_mov_b32 %0, 0x80" : "=v"(bit7));
    unsigned colmask = 0;  // which of the lane's 4 columns lie inside cvFindContours' zeroed frame
    for (int j = 0; j < 4; j++) colmask |= (!EDGE || (c0 + j >= 1 && c0 + j <= sw - 2)) ? (1u << j) : 0u;
    // Which pixels may yield border starts: the output lanes' (inside the image).  Strips away from the edges: all four pixels of
    // lanes HL .. 63-HR, a constant lane mask applied to the ballots (scalar); strips at an edge: per lane, bits j and 8+j (outer /
    // hole start nibbles) of a register.
    constexpr unsigned long long OUT_LANES = (~0ull << HL) & (~0ull >> HR);
    unsigned pxsel = 0;
    if (EDGE)
        for (int j = 0; j < 4; j++) pxsel |= (out_lane && c0 + j < sw) ? (0x101u << j) : 0u;

    // Mask rows are collected in LDS, 8 rows at a time, and written as whole 16x8-pixel tiles (128 contiguous bytes, 32
    // per lane) -- a row at a time would be 16-byte pieces of 15 different cache lines per wave, and on this hardware a
    // store issued in every iteration makes the loop-top wait for the prefetched source row wait for that store too
    // (vmcnt counts loads and stores in order).  Y0 is a multiple of 8 (api.hip / MARCH_CROP_ROWS), so a group of 8 rows
    // belongs to one work unit; padding rows/columns of the plane (rows up to sh rounded to 8, columns up to ns) may
    // receive stale values, nothing reads them.
    // A group of 8 rows is 15 tiles of 128 bytes = 120 pieces of 16 bytes (piece k: row k & 7 of tile k >> 3, at byte 16 k of the
    // strip's part of the tile row group).  Two store instructions, each writing CONTIGUOUS memory: lane l stores piece l, then
    // (l < 56) piece 64 + l -- every 64-byte sector is written whole by one instruction.  (Lane = (pair of rows, tile) with both
    // rows' pieces from one lane made every instruction write 16-byte pieces at a stride of 32: twice the write requests, each
    // half a sector.)  Piece 64 + l is the same row, eight tiles (32 lanes' dwords) further on: one LDS offset register.
    const int tile1 = strip * (SV / 16) + (lane >> 3), tile2 = tile1 + 8;
    const unsigned flush_src = (unsigned)(lane & 7) * 64u + (unsigned)(HL + 4 * (lane >> 3));   // dword index in rowbuf: row l & 7, first lane of tile l >> 3
    const unsigned flush_dst1 = tile1 * 16 < o.ns ? ((unsigned)tile1 << 7) + (unsigned)(lane & 7) * 16u : OOB;   // byte offsets inside a tile row group
    const unsigned flush_dst2 = lane < 56 && tile2 * 16 < o.ns ? ((unsigned)tile2 << 7) + (unsigned)(lane & 7) * 16u : OOB;
    auto flush_rows = [&](int yr_last) {   // yr_last: last row written; its group is complete or the unit ends
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        {
            const uint2* s0 = reinterpret_cast<const uint2*>(rowbuf + flush_src);   // (lanes 56..63 read past the tiles in the second read, inside this wave's rows: their store is dropped)
            const uint2 a0 = s0[0], a1 = s0[1];       // piece l: 16 bytes
            const uint2 b0 = s0[16], b1 = s0[17];     // piece 64 + l: 32 dwords further on
            const unsigned so = (unsigned)wave_uniform(((yr_last >> 3) * (o.ns >> 4)) << 7);
            typedef unsigned v4u __attribute__((ext_vector_type(4)));   // (non-temporal stores: written once, read much later)
            // The row group's offset is added to the lanes' offsets, the scalar offset field stays 0.  A store of more than 64 bits
            // must not be followed at once by a vector instruction that writes its data registers; the compiler's hazard
            // recogniser inserts the wait state only for buffer stores WITHOUT a register in the scalar offset field (with one the
            // ISA manual promises no hazard) -- and gfx950 has it all the same: with the offset in an SGPR a register copy the
            // allocator placed right behind the second store overwrote its first data dword now and then (a selector constant
            // in the mask plane, first dword of 16-byte pieces, only while the memory system was busy: frames >= 16 of a batch).
            // (OOB + so < 2^32: dropped all the same.)
            __builtin_amdgcn_raw_buffer_store_b128((v4u){a0.x, a0.y, a1.x, a1.y}, nbr_rs, (int)(flush_dst1 + so), 0, BUF_NT);
            __builtin_amdgcn_raw_buffer_store_b128((v4u){b0.x, b0.y, b1.x, b1.y}, nbr_rs, (int)(flush_dst2 + so), 0, BUF_NT);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    };

    // Border starts are staged per wave in LDS and appended to the global list with ONE atomic per flush: a
    // single list counter only sustains ~90 atomics/us chip-wide, which one atomic per image row would exceed.
    int staged = 0;  // wave-uniform
    auto flush = [&]() {
        if (staged == 0) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        int base = 0;
        if (lane == 0) base = atomicAdd(o.n_cands, staged);
        base = wave_uniform(base);
        for (int i = lane; i < staged; i += 64) {
            const unsigned e = stage[i];
            if (base + i < o.cap_cands) {
                StartCand sc;
                sc.roi = o.roi;
                sc.pos = (int)(e & 0x7fffffffu);
                sc.is_hole = (int)(e >> 31);
                o.cands[base + i] = sc;
            } else {
                atomicOr(o.err, ERR_CAND_OVERFLOW);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        staged = 0;
    };

    // raw source words of virtual row v (3 dwords BGR / 1 dword grey for fast lanes, 4 pixels for edge lanes)
    struct Raw { unsigned d0, d1, d2; };
    auto fetch = [&](int v, bool inside /* 0 <= v < sh is known; only asked for when the load plan holds (steady rows) */) -> Raw {
        Raw r = {0u, 0u, 0u};
        const int rv = inside ? v : reflect101(v, sh);
        if (inside || plan) {
            const int so = wave_uniform(rv * (int)src_stride);   // (< 2^31: a frame, or rows of the grey plane from the crop's origin on)
            if (BGR) {   // one (unaligned) buffer_load_dwordx3.  Not non-temporal: with the hint the kernel is 7 % slower alone on the GPU
                         // (5.8 against 5.4 ms per 2048 frames: the halo columns two strips share stop hitting in the L2); the grey
                         // and mask STORES carry it (+2 % with four contexts, round 3)
                typedef unsigned v3u __attribute__((ext_vector_type(3)));
                const v3u d = __builtin_amdgcn_raw_buffer_load_b96(src_rs, (int)goff, so, 0);
                r.d0 = d.x; r.d1 = d.y; r.d2 = d.z;
            } else {     // crops: ordinary loads -- the two concentric quads of a marker give two crops over nearly the same pixels
                r.d0 = __builtin_amdgcn_raw_buffer_load_b32(src_rs, (int)goff, so, 0);
            }
            return r;
        }
        const uint8_t* row = src + wave_uniform64((long long)rv * src_stride);
        if (!needed) return r;
        if (fast) {
            if (BGR) {
                if (aligned) {
                    const unsigned* p = reinterpret_cast<const unsigned*>(row + src_off);
                    r.d0 = p[0]; r.d1 = p[1]; r.d2 = p[2];
                } else {
                    const uint8_t* p = row + src_off;
                    r.d0 = p[0] | (p[1] << 8) | (p[2] << 16) | ((unsigned)p[3] << 24);
                    r.d1 = p[4] | (p[5] << 8) | (p[6] << 16) | ((unsigned)p[7] << 24);
                    r.d2 = p[8] | (p[9] << 8) | (p[10] << 16) | ((unsigned)p[11] << 24);
                }
            } else {
                const uintptr_t a = reinterpret_cast<uintptr_t>(row + src_off);
                const unsigned* p = reinterpret_cast<const unsigned*>(a & ~(uintptr_t)3);
                r.d0 = alignb(p[1], p[0], (unsigned)(a & 3));
            }
        } else if (BGR) {  // pack the 4 reflected pixels into the same 12-byte layout
            const uint8_t *p0 = row + xr0, *p1 = row + xr1, *p2 = row + xr2, *p3 = row + xr3;
            r.d0 = p0[0] | (p0[1] << 8) | (p0[2] << 16) | ((unsigned)p1[0] << 24);
            r.d1 = p1[1] | (p1[2] << 8) | (p2[0] << 16) | ((unsigned)p2[1] << 24);
            r.d2 = p2[2] | (p3[0] << 8) | (p3[1] << 16) | ((unsigned)p3[2] << 24);
        } else {
            r.d0 = row[xr0] | (row[xr1] << 8) | (row[xr2] << 16) | ((unsigned)row[xr3] << 24);
        }
        return r;
    };
    auto to_grey = [&](const Raw& r) -> unsigned {
        if (!BGR) return r.d0;
        // (1868 B + 9617 G + 4899 R + 8192) >> 14 with everything times four: (7472 B + 38468 G + 19596 R + 32768) >> 16 -- the
        // grey value is then BYTE 2 of the sum (< 2^24), which a byte permute picks: no shift per pixel.  Coefficient halves for
        // v_dot4_u32_u8: 7472 = 29*256+48, 38468 = 150*256+68, 19596 = 76*256+140.
        const unsigned KH = 29u | (150u << 8) | (76u << 16), KL = 48u | (68u << 8) | (140u << 16);
        const unsigned s0 = (dot4(r.d0, KH, 0u) << 8) + dot4(r.d0, KL, 32768u);                          // bytes b0 g0 r0 (x)
        const unsigned w1 = alignb(r.d1, r.d0, 3), w2 = alignb(r.d2, r.d1, 2);
        const unsigned s1 = (dot4(w1, KH, 0u) << 8) + dot4(w1, KL, 32768u);                              // b1 g1 r1 (x)
        const unsigned s2 = (dot4(w2, KH, 0u) << 8) + dot4(w2, KL, 32768u);                              // b2 g2 r2 (x)
        const unsigned s3 = (dot4(r.d2, KH << 8, 0u) << 8) + dot4(r.d2, KL << 8, 32768u);                // (x) b3 g3 r3
        const unsigned g01 = __builtin_amdgcn_perm(s1, s0, 0x0c0c0602u);   // byte 2 of s0, byte 2 of s1, 0, 0
        const unsigned g23 = __builtin_amdgcn_perm(s3, s2, 0x0c0c0602u);
        return g01 | (g23 << 16);
    };

    // Pyramid rows the unit needs: from qa (two rows before the first pyrUp row, Y0 - 4, that its first threshold row reads) to qb.
    // At the image's top the rows above 0 are copies of row 0 (BORDER_REPLICATE), so nothing before pyramid row -1 is read; at its
    // bottom the last pyrUp rows, sh - 2 and sh - 1, come with pyramid row sh / 2 (= row sh / 2 - 1 again) and the rows below are
    // copies: four source rows less at either end than the general formula asks for (a crop of 170 rows: 8 of 187).
    const int qa = Y0 == 0 ? -1 : Y0 / 2 - 3, qb = (Y1 + 3) / 2 + 1 < sh / 2 ? (Y1 + 3) / 2 + 1 : sh / 2;
    const int v_first = 2 * qa - 2, v_last = 2 * qb + 2;
    // Vertical pyrDown [1 4 6 4 1] by accumulation: pyramid row k (centre: source row 2k) is complete with source row 2k+2.  An
    // even row 2k gives 1 to row k-1 (which it completes), 6 to row k, 1 to row k+1; an odd row gives 4 to its two neighbours.
    // accA collects the row that completes next, accB the one after it (with the rounding constant): no 5-row window to shift.
    us2 accA = as_us2(0u), accB = as_us2(0u);   // packed pairs (column c0 | column c0+2 << 16), <= 16 * 4080 + 128
    unsigned prevP = 0;                      // previous pyramid row for the bottom border
    // Vertical pyrUp, all values times four so that a finished pixel is the HIGH byte of its 16-bit sum (no shift; the ring insert
    // picks that byte): with r(q) the horizontally up-sampled pyramid row q times 4 (even columns r0|r2<<16, odd r1|r3<<16),
    // pyrUp row 2(q-1)   = (r(q-2) + 6 r(q-1) + r(q) + 128) >> 8,   pyrUp row 2(q-1)+1 = (4 (r(q-1) + r(q)) + 128) >> 8
    // (the definition's (x + 32) >> 6 with x times 4; <= 4 * 16320 + 128 < 2^16).  Kept between rows: T = r(q-2) + 6 r(q-1) + 128
    // and b = r(q-1) + 128 -- a new row costs one add for the even row, an add and a multiply-add for the odd row, two for the state.
    us2 Te = as_us2(0u), To = as_us2(0u), be = as_us2(0u), bo = as_us2(0u);
    // The last 8 (virtual) pyrUp rows, one pair of registers per pixel with a row per byte, kept as a RING: virtual row vu goes
    // to byte vu & 3 of wa[p], which holds the current group of four rows (and, in the bytes not yet overwritten, the group
    // before the previous one); wb[p] holds the previous group; when a group begins the two words change names (v_swap_b32).
    // A new row is one byte permute per pixel (it overwrites a row that has left the window; nothing is shifted), the vertical
    // 7-tap Gaussian of a pixel is two v_dot4_u32_u8 whose coefficient words are the kernel rotated to the ring's position,
    // and the threshold's source byte (row vu-3) is picked by the compare itself.  In the steady rows the position is a
    // compile-time constant (the loop is unrolled over 4 rows); the rows at the top and bottom of a unit compute selectors
    // and coefficients on the scalar unit.  (An 8-row ring without the renaming needs the loop unrolled over 8 rows: with the
    // two instances of this function in one kernel that was 52 KB of code and 30 % slower -- instruction cache.)
    unsigned wa[4] = {0u, 0u, 0u, 0u}, wb[4] = {0u, 0u, 0u, 0u};
    unsigned m_prev = 0, u_prev = 0, x_prev = 0;   // mask of row y-1 still without its row below; what row y-1 gives to the row below it; its start nibbles

    // One source row.  S ("steady"): v lies where every range test below has a known outcome -- the rows it completes are
    // inside the work unit and away from the image's first and last rows -- so the tests (a scalar compare and branch
    // each, dozens per row) are compiled out; the rows at the top and bottom of a unit take the generic instance.
    // Source rows in flight per wave: AHEAD rows beyond the one being worked on (frames: 2, crops: 1).  With one row under way the
    // ~6 K resident waves have 4.7 MB in flight, which at the latency of a loaded memory system caps the frame kernel near 4 TB/s;
    // and when other contexts' kernels hold half of the wave slots the frame kernel's waves must each keep more in flight to
    // keep the memory system busy.
    constexpr int AHEAD = BGR ? OCVAR_ROWS_AHEAD_F : OCVAR_ROWS_AHEAD_C;
    Raw nxt = fetch(v_first, false), nxt2 = fetch(v_first + 1, false), nxt3 = {0u, 0u, 0u}, nxt4 = {0u, 0u, 0u};
    if (AHEAD >= 2) nxt3 = fetch(v_first + 2, false);
    if (AHEAD >= 3) nxt4 = fetch(v_first + 3, false);
    auto row = [&](const int v, auto steady_tag, auto parity_tag, auto ring_tag) {
        constexpr bool S = decltype(steady_tag)::value;
        constexpr int PAR = decltype(parity_tag)::value;   // 1: v is odd, 2: v is even, 0: not known at compile time
        constexpr int VM = decltype(ring_tag)::value;      // v & 3 of an even steady row (the ring position follows from it), -1: not known
        // Two rows in flight per wave (with one, the ~6 K resident waves have 4.7 MB under way, which at the latency of a loaded
        // memory system caps the kernel near 4 TB/s): this row's words were loaded two rows ago, and the row after the next is
        // requested as soon as they have been turned into grey -- into the registers they leave, so the two sets swap names
        // every row and nothing is copied (loaded before, a third set was live and the sets rotated through moves).
        unsigned g = to_grey(nxt);
        if (BGR) asm volatile("" : "+v"(g) : : "memory");   // (keeps the load below this point, and the conversion above it)
        else asm volatile("" : : : "memory");
        // (the request goes into the registers the converted row leaves: the sets swap names, nothing is copied in the unrolled loop)
        if (AHEAD == 1) {
            nxt = nxt2;
            if (S || v + 1 < v_last) nxt2 = fetch(v + 2, S);
        } else if (AHEAD == 2) {
            nxt = nxt2; nxt2 = nxt3;
            if (S || v + 2 < v_last) nxt3 = fetch(v + 3, S);
        } else {
            nxt = nxt2; nxt2 = nxt3; nxt3 = nxt4;
            if (S || v + 3 < v_last) nxt4 = fetch(v + 4, S);
        }
        if (EDGE && (S || plan)) g = __builtin_amdgcn_perm(g, g, gsel);   // (gsel: the identity in lanes that hold no reflected columns)
        // (a 32-bit store reads its data register as it issues: unlike the 128-bit tile stores in flush_rows it may keep the row's
        // offset in the scalar offset field; test_planes_of_every_frame_of_a_busy_batch compares the grey plane too)
        if (BGR && (S || (v >= Y0 && v < Y1)))   // (BGR: the frame pass, which always has a grey plane) rows [Y0,Y1) are real rows, each loaded exactly once
            __builtin_amdgcn_raw_buffer_store_b32(g, gray_rs, (int)out_off, wave_uniform(v * (int)o.gray_stride), BUF_NT);
        us2 hsum;
        {   // horizontal [1 4 6 4 1] at the lane's two even columns c0 and c0+2
            const unsigned gl = up1(g), gr = down1(g);
            const unsigned a = dot4(alignb(g, gl, 2), 0x04060401u, byte_of(g, 2));
            const unsigned b = dot4(g, 0x04060401u, gr & 255u);
            hsum = as_us2(a | (b << 16));
        }
        if (PAR == 1 || (PAR == 0 && (v & 1))) {
            accA += hsum * (unsigned short)4;
            accB += hsum * (unsigned short)4;
            return;
        }
        us2 P = (accA + hsum) >> (unsigned short)8;
        accA = accB + hsum * (unsigned short)6;
        accB = hsum + (unsigned short)128;
        if (!S && v < v_first + 4) return;
        const int q = (v - 2) >> 1;  // pyramid row completed by source row 2q+2
        if (EDGE) P = as_us2(__builtin_amdgcn_perm(up1(as_u32(P)), as_u32(P), selP));   // pyrUp right border: column pw stands for column pw-1
        if (!S && q == ph_) P = as_us2(prevP);       // pyrUp bottom border: row ph stands for row ph-1
        else if (!S && q < ph_) prevP = as_u32(P);   // (steady rows end before source row sh-2: generic rows with q < ph follow them and set it)
        unsigned rCe, rCo;
        {   // horizontal up-sampling at the lane's 4 columns: even (r0,r2) and odd (r1,r3)
            const unsigned pk = as_u32(P);
            const us2 Lv = as_us2(alignb(pk, up1(pk), 2));     // (P[k0-1], P[k0])
            const us2 Rv = as_us2(alignb(down1(pk), pk, 2));   // (P[k0+1], P[k0+2])
            rCe = as_u32((Lv + P * (unsigned short)6 + Rv) << (unsigned short)2);
            rCo = as_u32((P + Rv) << (unsigned short)4);
        }
        if (S || q >= qa + 2) {
            for (int par = 0; par < 2; par++) {
                const int u = 2 * (q - 1) + par;
                if (!S && (u < 0 || u >= sh)) continue;
                us2 Ue, Uo;   // pyrUp row u at the lane's even columns (c0, c0+2) and odd columns (c0+1, c0+3)
                if (par == 0) {
                    Ue = Te + as_us2(rCe);
                    Uo = To + as_us2(rCo);
                } else {   // 4 (b - 128 + r) + 128 = 4 (b + r) - 384 (mod 2^16)
                    Ue = (be + as_us2(rCe)) * (unsigned short)4 + (unsigned short)(65536 - 384);
                    Uo = (bo + as_us2(rCo)) * (unsigned short)4 + (unsigned short)(65536 - 384);
                }
                // rows above 0 / below sh-1 replicate row 0 / sh-1; one more virtual row flushes the last mask row
                const int vlo = (!S && u == 0) ? -3 : u, vhi = (!S && u == sh - 1) ? sh + 3 : u;
                for (int vu = vlo; vu <= vhi; vu++) {
                    const int slot = VM >= 0 ? ((VM + par) & 3) : (vu & 3);   // steady rows: vu = u = v - 4 + par
                    if (slot == 0) {   // a new group of four rows: the words change names
                        if constexpr (VM >= 0) {
#pragma unroll
                            for (int p = 0; p < 4; p++) asm("v_swap_b32 %0, %1" : "+v"(wa[p]), "+v"(wb[p]));
                        } else {
#pragma unroll
                            for (int p = 0; p < 4; p++) { const unsigned t = wa[p]; wa[p] = wb[p]; wb[p] = t; }
                        }
                    }
                    {   // pixel p of the new row (u0, u2 = bytes 1, 3 of Ue -- the high bytes of its halves --; u1, u3 of Uo) into byte `slot` of wa[p]
                        const unsigned kb = 8u * (unsigned)slot, keep = 0x03020100u & ~(0xffu << kb);
                        const unsigned s4 = keep | (5u << kb), s6 = keep | (7u << kb);
                        wa[0] = __builtin_amdgcn_perm(as_u32(Ue), wa[0], s4);
                        wa[1] = __builtin_amdgcn_perm(as_u32(Uo), wa[1], s4);
                        wa[2] = __builtin_amdgcn_perm(as_u32(Ue), wa[2], s6);
                        wa[3] = __builtin_amdgcn_perm(as_u32(Uo), wa[3], s6);
                    }
                    const int y = vu - 3;   // the ring now holds pyrUp rows y-3 .. y+3 (and row vu-7: weight 0)
                    if (!S && (y < Y0 - 1 || y > Y1 || y < 0)) continue;
                    unsigned nib = 0;
                    if (S || (y >= 1 && y <= sh - 2)) {
                        // 7x7 Gaussian, vertical pass first: [8 28 56 72 56 28 8] over rows y-3 .. y+3.  Seen as eight slots (wa's bytes,
                        // then wb's) with the newest row in slot 7 the weights would be 0 8 28 56 | 72 56 28 8; the newest row is in
                        // slot `slot` of wa: that word pair rotated by whole bytes.
                        // (<= 256*255: the sums fit 16 bits and are packed in pairs for the horizontal pass)
                        const unsigned long long KW = rotl64(0x081C3848381C0800ull, 8u * (unsigned)((slot + 1) & 7));
                        const unsigned kwa = (unsigned)KW, kwb = (unsigned)(KW >> 32);
                        unsigned V[4];
#pragma unroll
                        for (int p = 0; p < 4; p++) V[p] = dot4(wa[p], kwa, dot4(wb[p], kwb, 0u));
                        const us2 Va = as_us2(V[0] | (V[1] << 16));
                        us2 Vb = as_us2(V[2] | (V[3] << 16));
                        if (EDGE) Vb = as_us2(__builtin_amdgcn_perm(as_u32(Va), as_u32(Vb), selB));   // (columns sw, sw+1 of the lane that straddles the right edge)
                        // horizontal pass: 16-bit column sums of this lane (Va = V0 V1, Vb = V2 V3), the lane to the left (l) and to
                        // the right (r), two taps per v_dot2_u32_u16 -- with neighbouring columns paired every pixel's 7 taps are 4
                        // instructions.  The accumulator starts at the rounding constant minus 8<<16:
                        // src - mean > -8  <=>  sum + 32768 - (8<<16) < src<<16
                        us2 Val = as_us2(up1(as_u32(Va))), Vbl = as_us2(up1(as_u32(Vb)));     // V-4 V-3, V-2 V-1
                        us2 Var = as_us2(down1(as_u32(Va))), Vbr = as_us2(down1(as_u32(Vb))); // V4 V5, V6 V7
                        if (EDGE) {   // BORDER_REPLICATE: columns < 0 are column 0, columns >= sw are column sw-1 (selL, selR above)
                            Val = as_us2(__builtin_amdgcn_perm(as_u32(Va), as_u32(Val), selL));
                            Vbl = as_us2(__builtin_amdgcn_perm(as_u32(Va), as_u32(Vbl), selL));
                            Var = as_us2(__builtin_amdgcn_perm(as_u32(Vb), as_u32(Var), selR));
                            Vbr = as_us2(__builtin_amdgcn_perm(as_u32(Vb), as_u32(Vbr), selR));
                        }
                        const unsigned C0 = 32768u - (8u << 16);
                        const unsigned S0 = dot2(Val, K2(0, 8), dot2(Vbl, K2(28, 56), dot2(Va, K2(72, 56), dot2(Vb, K2(28, 8), C0))));
                        const unsigned S1 = dot2(Vbl, K2(8, 28), dot2(Va, K2(56, 72), dot2(Vb, K2(56, 28), dot2(Var, K2(8, 0), C0))));
                        const unsigned S2 = dot2(Vbl, K2(0, 8), dot2(Va, K2(28, 56), dot2(Vb, K2(72, 56), dot2(Var, K2(28, 8), C0))));
                        const unsigned S3 = dot2(Va, K2(8, 28), dot2(Vb, K2(56, 72), dot2(Var, K2(56, 28), dot2(Vbr, K2(8, 0), C0))));
                        // pyrUp row y itself -- slot (slot - 3) & 7 of the eight -- is the threshold's source; nib = 2 * nib + (S < src << 16),
                        // pixel 3 first: one compare and one add-with-carry per pixel
                        const int ts = (slot + 5) & 7;
                        if constexpr (VM >= 0) {
                            constexpr int TS = ((VM & 3) + 5) & 7;   // par = 0; par = 1 is the next slot
                            if (par == 0) {
                                first_bit<TS & 3>(nib, S3, TS < 4 ? wa[3] : wb[3]);
                                push_bit<TS & 3>(nib, S2, TS < 4 ? wa[2] : wb[2]);
                                push_bit<TS & 3>(nib, S1, TS < 4 ? wa[1] : wb[1]);
                                push_bit<TS & 3>(nib, S0, TS < 4 ? wa[0] : wb[0]);
                            } else {
                                constexpr int T1 = (TS + 1) & 7;
                                first_bit<T1 & 3>(nib, S3, T1 < 4 ? wa[3] : wb[3]);
                                push_bit<T1 & 3>(nib, S2, T1 < 4 ? wa[2] : wb[2]);
                                push_bit<T1 & 3>(nib, S1, T1 < 4 ? wa[1] : wb[1]);
                                push_bit<T1 & 3>(nib, S0, T1 < 4 ? wa[0] : wb[0]);
                            }
                        } else {
                            const unsigned tsel = 0x0c000c0cu | ((unsigned)ts << 16);   // byte ts of (wb : wa) into byte 2, zeros elsewhere
#define OCVAR_PUSH_BIT(S, P) asm("v_cmp_lt_i32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(nib) : "v"(S), "v"(__builtin_amdgcn_perm(wb[P], wa[P], tsel)) : "vcc")
                            OCVAR_PUSH_BIT(S3, 3);
                            OCVAR_PUSH_BIT(S2, 2);
                            OCVAR_PUSH_BIT(S1, 1);
                            OCVAR_PUSH_BIT(S0, 0);
#undef OCVAR_PUSH_BIT
                        }
                        if (EDGE || VM < 0) nib &= colmask;   // (strips inside the image: every column counts, and the four tests above left four bits)
                    }
                    // Row y's window -> table: its share of the masks of rows y-1, y, y+1 and its start nibbles.
                    // Byte offset of the table entry (16 bytes each; the table is laid out for it): bits 4..7 = own pixels, bit 8 = the left
                    // lane's pixel 3, bits 9, 10 = the right lane's pixels 0, 1 -- every lane shifts its nibble once, the neighbours'
                    // parts are an AND on the DPP-shifted value and a shift-or each.
                    const unsigned n4 = nib << 4;
                    const unsigned wdw16 = ((down1(n4) & rmask4) << 5) | (((up1(n4) & bit7) << 1) | n4);
                    const uint4 T = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(tab) + wdw16);
                    const unsigned nbr4 = m_prev | T.z;        // row y-1: E | NE N NW | W | SW S SE complete
                    unsigned st;                               // row y's own nibbles (low half of T.w) & row y-1's "row above" nibbles (high half of x_prev)
                    asm("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1" : "=v"(st) : "v"(T.w), "v"(x_prev));
                    m_prev = u_prev | T.y;
                    u_prev = T.x;
                    x_prev = T.w;
                    st = st & ~(st >> 4);                      // bits 0..3: outer starts, bits 8..11: hole starts of row y (other bits: don't care)
                    if (EDGE) st &= pxsel;
                    const int yr = y - 1;
                    if (S || (yr >= Y0 && yr < Y1)) {
                        rowbuf[(yr & 7) * 64 + lane] = nbr4;
                        if ((yr & 7) == 7 || (!S && yr == Y1 - 1)) flush_rows(yr);
                    }
                    // Plausible border starts of row y (sparse): necessary local conditions for being the raster-first pixel of
                    // a region.  Outer: foreground pixel whose W, NW, N, NE are background -- and not (E foreground and the pixel
                    // above E's east neighbour foreground: that one belongs to the same 8-connected component and comes earlier).
                    // Hole: background pixel whose W and N are foreground -- and E or NE foreground (if both are background they
                    // belong to the same 4-connected background region and NE comes earlier).
                    if ((!S && (y < Y0 || y >= Y1)) || (__ballot((st & 0xf0fu) != 0) & (EDGE ? ~0ull : OUT_LANES)) == 0) continue;
#pragma nounroll   // (rare rows, and the steady loop holds eight copies of this: kept small)
                    for (int j = 0; j < 4; j++) {   // (two vector instructions per pixel position that has no start: this runs on every third row)
                        const unsigned hit = st & (0x101u << j);   // pixel j: bit j an outer start, bit 8 + j a hole start (never both: foreground / background)
                        const unsigned long long mask = __ballot(hit != 0) & (EDGE ? ~0ull : OUT_LANES);
                        if (!mask) continue;
                        const int n = __popcll(mask);
                        if (staged + n > MARCH_STAGE) flush();
                        if ((mask >> lane) & 1ull)
                            stage[staged + __popcll(mask & ((1ull << lane) - 1ull))] = (unsigned)(y * o.ns + c0 + j) | (hit >> 8 ? 0x80000000u : 0u);
                        staged += n;
                    }
                }
            }
        }
        Te = as_us2(rCe) * (unsigned short)6 + be;
        To = as_us2(rCo) * (unsigned short)6 + bo;
        be = as_us2(rCe) + (unsigned short)128;
        bo = as_us2(rCo) + (unsigned short)128;
    };
    // steady rows: v and v+1 are real rows of this unit ([Y0, Y1), v+1 <= sh-1), and the rows an even v completes -- pyrUp
    // rows v-4, v-3, threshold rows v-7, v-6, mask rows v-8, v-7 -- are inside the unit, at least one row away from the
    // image's top and bottom (no replicated pyrUp rows, no zeroed contour-frame rows) and not the unit's last mask row
    // -- and the unit has the load plan (else no row is steady: the steady instance of the row body does not carry the byte-wise
    // loads of images narrower than 32 columns)
    const bool steady_ok = plan;
    const int vs0 = (Y0 + 8 > 8 ? Y0 + 8 : 8) | 1, vs1 = !steady_ok ? -1 : Y1 - 1 < sh - 2 - AHEAD ? Y1 - 1 : sh - 2 - AHEAD;   // (vs0 odd: steady rows run in pairs; v + 2 is a real row: prefetched unreflected)
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    using P2 = std::integral_constant<int, 2>;
    using RX = std::integral_constant<int, -1>;   // ring position not known at compile time
    int v = v_first;
#pragma nounroll
    for (int phase = 0; phase < 2; phase++) {   // generic rows, steady rows, generic rows: one copy of each instance
        const int end = phase == 0 ? (vs0 <= v_last + 1 ? vs0 : v_last + 1) : v_last + 1;
        for (; v < end; v++) row(v, std::false_type(), P0(), RX());
        if (phase == 0)
            while (v + 1 <= vs1) {
                if ((v & 3) == 1 && v + 3 <= vs1) {   // four rows with the ring's position known: the hot loop
                    row(v, std::true_type(), P1(), RX());
                    row(v + 1, std::true_type(), P2(), std::integral_constant<int, 2>());
                    row(v + 2, std::true_type(), P1(), RX());
                    row(v + 3, std::true_type(), P2(), std::integral_constant<int, 0>());
                    v += 4;
                } else {                               // a pair before the first or after the last group of four
                    row(v, std::true_type(), P1(), RX());
                    row(v + 1, std::true_type(), P2(), RX());
                    v += 2;
                }
            }
    }
    flush();
}

// Workgroups of ONE wave (BW = 1): the waves of a workgroup must be placed on their CU together, so a four-wave workgroup needs
// room on all four SIMDs at once -- and with other contexts' kernels on the GPU (a follower wave of 110 registers on one SIMD)
// a CU holds a whole workgroup less.  A one-wave workgroup takes any SIMD with 96 free registers and 6 KB of LDS.
constexpr int BW = OCVAR_BIN_WG_WAVES;

__global__ __launch_bounds__(64 * BW) __attribute__((amdgpu_waves_per_eu(OCVAR_WAVES_F, 8))) void binarise_frames_kernel(Workspace ws, const uint8_t* bgr, int row_stride, size_t frame_stride) {
    __shared__ unsigned stage[BW][MARCH_STAGE];
    __shared__ __attribute__((aligned(16))) unsigned rowbuf[BW][8 * 64];
    __shared__ uint4 tab[128];
    for (unsigned i = threadIdx.x; i < 128u; i += 64u * BW) tab[i] = mask_table_entry(table_window(i));
    __syncthreads();
    const int per_frame = ws.frame_strips * ws.frame_chunks;
    // XCD-aware order.  Consecutive workgroups go round the 8 XCDs, each with its own L2; in launch order the workgroups of one
    // frame -- neighbouring strips and chunks, which share halo columns and rows -- would land on 8 different L2s and every halo
    // would be fetched from the fabric twice.  Workgroup b therefore works on frame-group (b / 8) / G, frame b % 8 of that group,
    // workgroup (b / 8) % G of that frame (G workgroups per frame): a frame's workgroups run on one XCD, next to each other in
    // time.  (Only when a frame is a whole number of workgroups; the frames beyond the last group of 8 keep launch order.)
    int wg = (int)blockIdx.x;
    if (per_frame % BW == 0) {
        const int G = per_frame / BW, grouped = (ws.n_frames & ~7) * G;
        if (wg < grouped) {
            const int k = wg >> 3;
            wg = ((k / G) * 8 + (wg & 7)) * G + k % G;
        }
    }
    const int unit = wg * BW + wave_uniform((int)(threadIdx.x >> 6));
    if (unit >= per_frame * ws.n_frames) return;
    const int f = unit / per_frame, rem = unit % per_frame;
    const int chunk = rem / ws.frame_strips, strip = rem % ws.frame_strips;
    const int Y0 = chunk * ws.frame_chunk_rows;
    const int Y1 = Y0 + ws.frame_chunk_rows < ws.sh ? Y0 + ws.frame_chunk_rows : ws.sh;
    MarchOut o;
    o.gray = ws.gray + (size_t)f * gray_plane_bytes(ws.W, ws.H);
    o.gray_stride = gray_pitch(ws.W);
    o.nbr = ws.nbr_frame + (size_t)f * nbr_plane_bytes(ws.ns, ws.sh);
    o.ns = ws.ns;
    o.roi = f;
    o.cands = ws.cands_frame;
    o.n_cands = ws.counters + CNT_FRAME_CANDS;
    o.cap_cands = ws.cap_frame_cands;
    o.err = ws.counters + CNT_ERR;
    const int w4 = wave_uniform((int)(threadIdx.x >> 6));
    const bool edge = strip == 0 || ((ws.sw - 1 - (strip * MARCH_STRIP - 4 * MARCH_HALO_L)) >> 2) <= 63;   // march_unit's left_edge || right_edge
    if (edge) march_unit<true, true>(bgr + (size_t)f * frame_stride, row_stride, 0, ws.sw, ws.sh, strip, Y0, Y1, o, stage[w4], rowbuf[w4], tab);
    else march_unit<true, false>(bgr + (size_t)f * frame_stride, row_stride, 0, ws.sw, ws.sh, strip, Y0, Y1, o, stage[w4], rowbuf[w4], tab);
}

// Odd width / height: the last column / row lies outside the even working size (opencvar.cpp:158) but is still
// greyed by cvarArMultRegistration's BGR2GRAY.
__global__ __launch_bounds__(256) void grey_edges_kernel(Workspace ws, const uint8_t* bgr, int row_stride, size_t frame_stride) {
    const int f = blockIdx.y;
    const uint8_t* src = bgr + (size_t)f * frame_stride;
    uint8_t* g = ws.gray + (size_t)f * gray_plane_bytes(ws.W, ws.H);
    const int pitch = gray_pitch(ws.W);
    const int n_col = ws.W > ws.sw ? ws.H : 0, n_row = ws.H > ws.sh ? ws.W : 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_col + n_row; i += gridDim.x * blockDim.x) {
        const int x = i < n_col ? ws.sw : i - n_col, y = i < n_col ? i : ws.sh;
        const uint8_t* p = src + (long long)y * row_stride + 3 * x;
        const uint8_t v = (uint8_t)grey_of(p[0], p[1], p[2]);
        g[(long long)y * pitch + gray_col(x)] = v;
        // (a column within 8 of a panel's start is also kept at the end of the panel before: readers take the panel of a run's first column)
        if (x >= GRAY_PANEL_COLS && x % GRAY_PANEL_COLS < GRAY_PANEL_LEAD) g[(long long)y * pitch + gray_col(x - GRAY_PANEL_LEAD) + GRAY_PANEL_LEAD] = v;
    }
}

// The reference greys the caller's frame in place (opencvar.cpp:624-627).  Done as its own pass over the
// grey plane so that no wave ever reads a half-written BGR pixel of a neighbouring strip's halo.
__global__ __launch_bounds__(256) void grey_writeback_kernel(Workspace ws, uint8_t* bgr, int row_stride, size_t frame_stride) {
    const int f = blockIdx.y;
    const uint8_t* g = ws.gray + (size_t)f * gray_plane_bytes(ws.W, ws.H);
    const int pitch = gray_pitch(ws.W);
    uint8_t* dst = bgr + (size_t)f * frame_stride;
    const long long n = (long long)ws.W * ws.H;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % ws.W), y = (int)(i / ws.W);
        const uint8_t v = g[(long long)y * pitch + gray_col(x)];
        uint8_t* q = dst + (long long)y * row_stride + 3 * x;
        q[0] = q[1] = q[2] = v;
    }
}

__global__ __launch_bounds__(64 * BW) __attribute__((amdgpu_waves_per_eu(OCVAR_WAVES_C, 8))) void binarise_crops_kernel(Workspace ws) {
    __shared__ unsigned stage[BW][MARCH_STAGE];
    __shared__ __attribute__((aligned(16))) unsigned rowbuf[BW][8 * 64];
    __shared__ uint4 tab[128];
    for (unsigned i = threadIdx.x; i < 128u; i += 64u * BW) tab[i] = mask_table_entry(table_window(i));
    __syncthreads();
    int n_units = ws.counters[CNT_CROP_TILES];
    if (n_units > ws.cap_crop_tiles) n_units = ws.cap_crop_tiles;
    const int wave = wave_uniform((int)(threadIdx.x >> 6));
    // crop units differ in size (rows, and the follow-up of their start lists): waves pull them from a ticket counter.
    // (The ticket test uses an opaque copy of the lane id: see follow.hip::ticket_lane for the compiler hazard.)
    int* ticket = ws.counters + CNT_TICKET_BC;
    // a wave can be handed at most n_units tickets (+ the one that tells it to stop): anything beyond that means the loop's
    // control flow is broken; report instead of spinning
    for (int guard = n_units + 1;; guard--) {
        if (guard < 0) {
            if ((threadIdx.x & 63) == 0) atomicOr(ws.counters + CNT_ERR, ERR_TICKET_RUNAWAY);
            break;
        }
        int lane_id = (int)(threadIdx.x & 63);
        asm volatile("" : "+v"(lane_id));
        int u = 0;
        if (lane_id == 0) u = atomicAdd(ticket, 1);
        u = wave_uniform(u);
        if (u >= n_units) break;
        const TileDesc td = ws.tiles_crop[u];
        const Roi r = ws.rois_crop[td.roi];
        const int pitch = gray_pitch(ws.W);
        const uint8_t* src = ws.gray + (size_t)r.frame * gray_plane_bytes(ws.W, ws.H) + (size_t)r.y0 * pitch;   // (row y0 of the frame's panelled grey plane; the columns go through gray_col(r.x0 + column))
        const int Y1 = td.y0 + MARCH_CROP_ROWS < r.sh ? td.y0 + MARCH_CROP_ROWS : r.sh;
        MarchOut o;
        o.gray = nullptr;
        o.gray_stride = 0;
        o.nbr = ws.nbr_crop + r.nbr_off;
        o.ns = r.ns;
        o.roi = td.roi;
        o.cands = ws.cands_crop;
        o.n_cands = ws.counters + CNT_CROP_CANDS;
        o.cap_cands = ws.cap_crop_cands;
        o.err = ws.counters + CNT_ERR;
        march_unit<false, true>(src, pitch, r.x0, r.sw, r.sh, td.x0, td.y0, Y1, o, stage[wave], rowbuf[wave], tab);
    }
}

void launch_binarise_frames(const Workspace& ws, const uint8_t* d_bgr, int row_stride, size_t frame_stride, int grey_in_place,
                            hipStream_t stream) {
    const int units = ws.frame_strips * ws.frame_chunks * ws.n_frames;
    if (units <= 0) return;
    hipLaunchKernelGGL(binarise_frames_kernel, dim3((units + BW - 1) / BW), dim3(64 * BW), 0, stream, ws, d_bgr, row_stride, frame_stride);
    if (ws.W > ws.sw || ws.H > ws.sh)
        hipLaunchKernelGGL(grey_edges_kernel, dim3(8, ws.n_frames), dim3(256), 0, stream, ws, d_bgr, row_stride, frame_stride);
    if (grey_in_place)
        hipLaunchKernelGGL(grey_writeback_kernel, dim3(1024, ws.n_frames), dim3(256), 0, stream, ws, const_cast<uint8_t*>(d_bgr),
                           row_stride, frame_stride);
}

void launch_binarise_crops(const Workspace& ws, hipStream_t stream) {
    hipLaunchKernelGGL(binarise_crops_kernel, dim3(ws.crop_blocks * (4 / BW)), dim3(64 * BW), 0, stream, ws);   // (crop_blocks counts four waves each)
}

}  // namespace ocvar
