// 3x3 stride-1 pad-1 convolution by Winograd F(4x4,3x3) on the gfx950 FP32 matrix cores: 4x fewer multiplies than the
// direct form (F(2x2,3x3) of conv_wino.hip: 2.25x).  Used for the deep layers, where the matrix pipe is the bound.
//
//   tile  = 4x4 output pixels (6x6 input patch d, origin (4ty-1, 4tx-1), zero outside the image)
//   V     = Bt d B      Bt = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
//   U     = G g Gt      G  = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]  (float64, once)
//   M_p   = sum_c U_p[n][c] * V_p[c][tile]    for the 36 positions p = 6i+j -- the MFMA part
//   Y     = At M A      At = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
//   y     = act(Y*scale[n] + shift[n]) + residual
// float32 throughout; against float64 the error is ~3e-6 rms of O(1) outputs (the direct form: 2e-7; tests/
// test_winograd_algebra.py), i.e. 30x inside the 1e-4 the detections are held to.
//
// Two launches per layer (four when the K-cut tail applies, see mydet_conv2d_wino4_f32):
//  1. wino4_input_kernel: V = Bt d B once per (tile, input channel), written to a workspace in MFMA-fragment order
//     [tile block of 32][k/4][position group of 4][k%4][tile] float4 = positions 4g..4g+3.  An HBM-bound pass (x read once,
//     2.25x its size written) with 16-byte accesses on both sides: 5.3-5.7 TB/s (round 4; the dword-load form of rounds
//     2-3 ran at 2.8-4.3 TB/s).  Fusing it into the GEMM kernel was measured and lost: the patch loads per (tile, channel)
//     are repeated by every output-channel block and the ~150 transform VALU per element issue beside the MFMAs at full
//     price (profiles/r02_wino4_notes.md).
//  2. conv_wino4_kernel: a workgroup of 4 waves owns 32 output channels x 32 tiles; wave (wc, wt) owns channels 16wc.. x
//     tiles 16wt.. for all 36 positions (36 x 4 = 144 accumulator registers).  Two workgroups per CU (72 KB of LDS each,
//     one wave of each per SIMD).  K = Cin is walked 4 channels (one MFMA k-step) per stage through two LDS stages; both
//     operands of stage kt+1 -- 18 KB of V and 18 KB of U, each contiguous in memory in exactly the LDS order -- are
//     copied global -> LDS with the DMA path (buffer_load ... lds: no registers, no VALU) under the 36 MFMAs per wave of
//     stage kt; ONE barrier per stage.  One ds_read_b128 per operand feeds four MFMAs.  The loop body is DMA issue,
//     18 LDS reads and 36 MFMAs -- no VALU.
//     Output channels are MFMA rows, so a lane ends up with 4 consecutive channels of a tile: after At M A, sixteen
//     16-byte stores; the residual loads of a row are in flight before its first store.
// Replaces the same ATen chain as conv_wino.hip (models/modules.py:69-73,94-95).
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace {

constexpr unsigned OOB = 0xFFFFFFFFu;
constexpr int NW = 4;                                  // waves per workgroup of the GEMM kernel
constexpr int CH = 8 * NW, TILES = 32, KC = 4, NPG = 9; // output channels / tiles per workgroup, k per stage, position groups
constexpr int U_BYTES = NPG * KC * CH * 16;            // 18 432: one 4-channel stage of U
constexpr int UP = U_BYTES / 1024;                     // its 1 KB pieces
constexpr int V_BYTES = NPG * KC * TILES * 16;         // 18 432: one 4-channel stage of V
constexpr int STAGE = U_BYTES + V_BYTES;               // 36 864
constexpr int LDS_BYTES = 2 * STAGE;                   // 73 728: two workgroups per CU
constexpr int COUT_PAD = 64;                           // U rows are padded to this many output channels
constexpr int64_t TAIL_BYTES = (int64_t)1024 * 16 * 64 * NW * 16;   // 64 MiB behind V: partial tiles of the K-cut tail (<= 1024 pieces)

struct W4Args {
    const float *x, *u, *scale, *shift, *res;
    float *y, *v;                                      // v: transform-domain input (workspace)
    int64_t ldx, ldr, ldy;
    int B, H, W, Cin, Cout, CoutP;
    int TH, TW, MT, ntn, nblk;
    int nmb, nbn, rn_log2;                             // item order of the GEMM kernel (see there)
    // K-cut tail (see mydet_conv2d_wino4_f32): the last 64-item blocks of the item order run in groups as `splits` pieces along K whose
    // output-domain partial tiles go to `part` ([block slot][piece][16 output pixels][256 threads] float4) and are summed by
    // wino4_fixup_kernel; the main launch covers the ids below tail_id0
    // A group: blocks [tail_id0 / 64, + tail_blocks), of each the first `tail_stride` ids (a block of the ragged last row has
    // its valid items in front), `splits` pieces; `part` is this group's share of the scratch.
    int tail_id0, tail_blocks, tail_stride, splits;
    float *part;
#ifdef MYDET_DIAG
    // diagnostic build only (`make EXTRA=-DMYDET_DIAG`, tools/r04_clock.py; the results of such a build are NOT valid):
    // MYDET_W4_DBG & 7 = 1 every stage reads the same 36 KB, 2 no DMA after stage 1, 3 U always stage 0; & 8 = stamp the K
    // loop of every workgroup (s_memtime / s_memrealtime) into the consumed V workspace
    int dbg;
#endif
};

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const float *base, int64_t bytes) {
    const uint64_t a = (uint64_t)base;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    const int64_t capped = bytes > 0x7FFFFFF0ll ? 0x7FFFFFF0ll : bytes;
    const int n = __builtin_amdgcn_readfirstlane((int)capped);
    return __builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)hi << 32) | lo), 0, n, 0x00020000);
}

// Bt x (x = six values): all six rows, or only rows 0-2 / 3-5 (the column pass of a thread that owns half the rows)
#define W4_BT_LO(o, x0, x1, x2, x3, x4, x5)               \
    {                                                     \
        o[0] = fmaf(4.0f, x0, fmaf(-5.0f, x2, x4));       \
        o[1] = (x3 + x4) - 4.0f * (x1 + x2);              \
        o[2] = (x4 - x3) + 4.0f * (x1 - x2);              \
    }
#define W4_BT_HI(o, x0, x1, x2, x3, x4, x5)               \
    {                                                     \
        o[0] = (x4 - x2) + 2.0f * (x3 - x1);              \
        o[1] = (x4 - x2) - 2.0f * (x3 - x1);              \
        o[2] = fmaf(4.0f, x1, fmaf(-5.0f, x3, x5));       \
    }
#define W4_BT(o, x0, x1, x2, x3, x4, x5)                  \
    {                                                     \
        W4_BT_LO(o, x0, x1, x2, x3, x4, x5)               \
        o[3] = (x4 - x2) + 2.0f * (x3 - x1);              \
        o[4] = (x4 - x2) - 2.0f * (x3 - x1);              \
        o[5] = fmaf(4.0f, x1, fmaf(-5.0f, x3, x5));       \
    }


// ---- 1. input transform.  Workgroup = 32 tiles (one tile block) x 8 k-quads (32 input channels); thread = (tile, k-quad):
// it loads the 36 pixels of the patch as float4 -- a wave = 8 tiles x 8 quads reads eight whole 128-byte lines per load
// instruction (rounds 2-3: one channel per thread, 36 dword loads: eight 32-byte sectors per instruction, four times as
// many instructions, 2.8-4.3 TB/s) --, transforms its four channels side by side (the same expressions per element as
// before: bit-identical V) and writes 36 float4 (position group g, channel k of the quad): per store instruction eight
// whole 128-byte lines, one per k-quad.  231 VGPRs: two waves per SIMD, each with 36 KB of loads in flight -- the
// launch is bandwidth-, not occupancy-bound (tools/micro/store_pattern.hip: this store pattern alone runs at the
// rate of a linear fill).
__global__ __launch_bounds__(256, 2) void wino4_input_kernel(const W4Args p) {
    const int tid = threadIdx.x;
    const int sq = tid & 7, st = tid >> 3;             // k-quad of the 8, tile of the 32 (wave w: tiles 8w .. 8w+7)
    const int mb = blockIdx.x, q = blockIdx.y * 8 + sq, c = q * 4;
    const int tpi = p.TH * p.TW, nk = p.Cin >> 2;
    const int mt = mb * TILES + st;
    const int b0 = (mb * TILES) / tpi;
    const int64_t img = (int64_t)p.H * p.W * p.ldx;
    // the buffer starts one row + one pixel BEFORE image b0, so that the patch origin (-1, -1) of its first tile is offset 0:
    // the range check sees the voffset only, and a negative one would read as out of range (those addresses are never
    // touched: row -1 and column -1 are masked to OOB below)
    const int64_t lead = (int64_t)(p.W + 1) * p.ldx;
    const __amdgpu_buffer_rsrc_t xr = rsrc(p.x + b0 * img - lead, ((p.B - b0) * img + lead) * 4);
    const int mm = mt < p.MT ? mt : p.MT - 1;
    const int b = mm / tpi, r = mm - b * tpi, ty = r / p.TW, tx = r - ty * p.TW;
    const int iy0 = 4 * ty - 1, ix0 = 4 * tx - 1;
    const unsigned vbase = mt < p.MT && c < p.Cin
                               ? (unsigned)((((((int64_t)(b - b0) * p.H + iy0) * p.W + ix0) * p.ldx + c) + lead) * 4)
                               : OOB;                  // tiles past the end and quads past Cin transform zeros
    const unsigned colstep = (unsigned)(p.ldx * 4), rowstep = (unsigned)(p.W * p.ldx * 4);
    f32x4 gv[36];
#pragma unroll
    for (int o = 0; o < 36; ++o) {
        const bool ok = (unsigned)(iy0 + o / 6) < (unsigned)p.H && (unsigned)(ix0 + o % 6) < (unsigned)p.W;
        gv[o] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                              xr, ok ? vbase : OOB,
                                              __builtin_amdgcn_readfirstlane((unsigned)(o / 6) * rowstep + (unsigned)(o % 6) * colstep), 0));
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int m = 0; m < 6; ++m) {
            float o[6];
            W4_BT(o, gv[m][e], gv[6 + m][e], gv[12 + m][e], gv[18 + m][e], gv[24 + m][e], gv[30 + m][e])
#pragma unroll
            for (int i = 0; i < 6; ++i) gv[6 * i + m][e] = o[i];
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            float o[6];
            W4_BT(o, gv[6 * i][e], gv[6 * i + 1][e], gv[6 * i + 2][e], gv[6 * i + 3][e], gv[6 * i + 4][e], gv[6 * i + 5][e])
#pragma unroll
            for (int m = 0; m < 6; ++m) gv[6 * i + m][e] = o[m];
        }
    }
    if (c < p.Cin) {                                   // Cin % 4 == 0: a k-quad is written whole or not at all
        // non-temporal: V is read back by the next launch from HBM / the Infinity Cache, not from this XCD's L2 (in the
        // model +0.8 % images/s over plain stores; back to back on its own the plain form is the faster one)
        f32x4 *dst = reinterpret_cast<f32x4 *>(p.v) + ((int64_t)mb * nk + q) * (NPG * KC * TILES) + st;
#pragma unroll
        for (int g = 0; g < NPG; ++g)
#pragma unroll
            for (int k = 0; k < KC; ++k) {
                const f32x4 val = f32x4{gv[4 * g][k], gv[4 * g + 1][k], gv[4 * g + 2][k], gv[4 * g + 3][k]};
                __builtin_nontemporal_store(val, dst + (g * KC + k) * TILES);
            }
    }
}

// ---- 2. transform-domain GEMMs + output transform + epilogue
template <int ACT, bool RES, bool PART = false>
__global__ __launch_bounds__(64 * NW, 2) void conv_wino4_kernel(const W4Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Item order: an XCD's share of the grid is a contiguous id range (mydet_xcd_remap), and each 64 consecutive ids --
    // the workgroups its 32 CUs run together -- form a block of RM tile blocks x RN channel blocks (RM * RN = 64, RN = 8
    // from 256 output channels up).  They walk K together, so a V stage is fetched from HBM once per RN workgroups
    // and a U stage once per RM; with the tile-block-major order (RM = 2 at Cout = 1024) the deep layers re-read U
    // a dozen times over.  Ids of the padded grid that fall outside return at once.
    // PART: the grid is `splits` copies of the tail's ids, piece-major -- the 64 workgroups an XCD runs together are one block
    // at one K range, as in the main launch
    int id, piece = 0, slot = 0;
    if (PART) {
        const int per = p.tail_blocks * p.tail_stride;
        const int lid = mydet_xcd_remap(blockIdx.x, per * p.splits);
        piece = lid / per;
        slot = lid - piece * per;
        const int blk = slot / p.tail_stride;
        id = p.tail_id0 + blk * 64 + (slot - blk * p.tail_stride);
    } else {
        id = mydet_xcd_remap(blockIdx.x, p.nblk);
    }
    const int bi = id >> 6, w = id & 63;
    const int mb = (bi / p.nbn) * (64 >> p.rn_log2) + (w >> p.rn_log2);
    const int nb = (bi % p.nbn) * (1 << p.rn_log2) + (w & ((1 << p.rn_log2) - 1));
    if (mb >= p.nmb || nb >= p.ntn) return;
    const int tpi = p.TH * p.TW;
    const int nk = p.Cin >> 2;
    const int m0 = mb * TILES, n0 = nb * CH;
    const int b0 = m0 / tpi;
    const int wc = wave % (NW / 2), wt = wave / (NW / 2);
    const int fr = lane & 15, fq = lane >> 4;

    // stage = [U: 9 position groups x 4 k x 32 channels float4 | V: 9 x 4 x 32 tiles float4]; both arrive by DMA in 1 KB
    // pieces (one per wave instruction), wave w copies pieces w, w + 4, ... of each operand.
    //   U: piece = two runs of 32 float4 (the block's channels) 16*CoutP bytes apart (layout of wino4_weights_kernel)
    //   V: the stage is one contiguous 18 KB run of the workspace
    const __amdgpu_buffer_rsrc_t ur = rsrc(p.u, (int64_t)p.Cin * 36 * p.CoutP * 4);
    const unsigned urun = (unsigned)p.CoutP * 16u;
    const unsigned uoff = (unsigned)((n0 + (lane & 31)) * 16) + (lane >> 5) * urun;
#ifdef MYDET_DIAG
    const int dmode = p.dbg & 7;
    const __amdgpu_buffer_rsrc_t vr = rsrc(p.v + (int64_t)(dmode == 1 ? 0 : mb) * nk * (V_BYTES / 4), (int64_t)nk * V_BYTES);
#else
    const __amdgpu_buffer_rsrc_t vr = rsrc(p.v + (int64_t)mb * nk * (V_BYTES / 4), (int64_t)nk * V_BYTES);
#endif
    const unsigned voff = (unsigned)(lane * 16);
    auto load_stage = [&](int kt) __attribute__((always_inline)) {
        char *dst = smem + (kt & 1) * STAGE;
#ifdef MYDET_DIAG
        if (dmode == 2 && kt > 1) return;
        const unsigned su = dmode == 1 || dmode == 3 ? 0u : (unsigned)kt * 36u * urun, sv = dmode == 1 ? 0u : (unsigned)kt * V_BYTES;
#else
        const unsigned su = (unsigned)kt * 36u * urun, sv = (unsigned)kt * V_BYTES;
#endif
#pragma unroll
        for (int j = 0; j < (UP + 18 + NW - 1) / NW; ++j) {
            const int i = wave + NW * j;               // index in [U pieces | V pieces]
            if (i < UP)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ur, (__attribute__((address_space(3))) void *)(dst + i * 1024), 16, uoff,
                                                         __builtin_amdgcn_readfirstlane(su + (unsigned)(2 * i) * urun), 0, 0);
            else if (i < UP + 18)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(vr, (__attribute__((address_space(3))) void *)(dst + U_BYTES + (i - UP) * 1024), 16, voff,
                                                         __builtin_amdgcn_readfirstlane(sv + (unsigned)(i - UP) * 1024u), 0, 0);
        }
    };

    const unsigned rd_u = (unsigned)((fq * CH + wc * 16 + fr) * 16);                     // + position group * KC*CH*16
    const unsigned rd_v = (unsigned)(U_BYTES + (fq * TILES + wt * 16 + fr) * 16);         // + position group * KC*TILES*16
    f32x4 acc[36];
#pragma unroll
    for (int q = 0; q < 36; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- epilogue addressing: lane = tile m0 + 16wt + fr, components = channels n0 + 16wc + 4fq + (0..3); output pixel
    // (a, c) of the tile: voffset = the tile's first pixel (or OOB), (a*W + c) pixels ride in the scalar offset
    const int n = n0 + wc * 16 + fq * 4;
    const bool nok = n < p.Cout;                       // Cout % 4 == 0
    const int64_t oimg = (int64_t)p.H * p.W;
    const __amdgpu_buffer_rsrc_t yr = rsrc(p.y + b0 * oimg * p.ldy, (p.B - b0) * oimg * p.ldy * 4);
    const __amdgpu_buffer_rsrc_t rr = rsrc(RES ? p.res + b0 * oimg * p.ldr : p.y, (p.B - b0) * oimg * (RES ? p.ldr : p.ldy) * 4);
    unsigned ybase, rbase;
    bool orow[4], ocol[4];
    {
        const int mt = m0 + wt * 16 + fr;
        const bool tok = mt < p.MT && nok;
        const int mm = mt < p.MT ? mt : p.MT - 1;
        const int b = mm / tpi, r = mm - b * tpi, ty = r / p.TW, tx = r - ty * p.TW;
        const int oy = 4 * ty, ox = 4 * tx;
        const int64_t pix = ((int64_t)(b - b0) * p.H + oy) * p.W + ox;
        ybase = (unsigned)((pix * p.ldy + n) * 4);
        rbase = (unsigned)((pix * p.ldr + n) * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            orow[i] = tok && oy + i < p.H;
            ocol[i] = ox + i < p.W;
        }
    }
    f32x4 rv[16];                                      // residual of the 4x4 outputs: loaded under the last stage's MFMAs

    const int kt0 = PART ? (int)((unsigned)(nk * piece) / (unsigned)p.splits) : 0;
    const int kt1 = PART ? (int)((unsigned)(nk * (piece + 1)) / (unsigned)p.splits) : nk;
    load_stage(kt0);
    __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0): the DMA'd stage has landed
    __syncthreads();
#ifdef MYDET_DIAG
    const uint64_t dbg_t0 = __builtin_amdgcn_s_memtime(), dbg_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    for (int kt = kt0; kt < kt1; ++kt) {
        if (kt + 1 < kt1) load_stage(kt + 1);          // its buffer was last read before the previous barrier
        if (RES && kt == kt1 - 1) {
#pragma unroll
            for (int o = 0; o < 16; ++o)
                rv[o] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                      rr, orow[o >> 2] && ocol[o & 3] ? rbase : OOB,
                                                      __builtin_amdgcn_readfirstlane((unsigned)(((o >> 2) * p.W + (o & 3)) * p.ldr * 4)), 0));
        }
        const char *cu = smem + (kt & 1) * STAGE + rd_u;
        const char *cv = smem + (kt & 1) * STAGE + rd_v;
        // fragment reads run one position group ahead of the MFMAs (two register sets)
        f32x4 fu[2], fv[2];
        fu[0] = *reinterpret_cast<const f32x4 *>(cu);
        fv[0] = *reinterpret_cast<const f32x4 *>(cv);
#pragma unroll
        for (int g = 0; g < NPG; ++g) {
            if (g + 1 < NPG) {
                fu[(g + 1) & 1] = *reinterpret_cast<const f32x4 *>(cu + (g + 1) * (KC * CH * 16));
                fv[(g + 1) & 1] = *reinterpret_cast<const f32x4 *>(cv + (g + 1) * (KC * TILES * 16));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                acc[4 * g + e] = __builtin_amdgcn_mfma_f32_16x16x4f32(fu[g & 1][e], fv[g & 1][e], acc[4 * g + e], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // (also after the last stage: the residual loads issued under it are then complete before the epilogue.  An earlier
        // arrangement of those loads gave wrong residuals without it; in the present one hipcc's own count is right --
        // listing in profiles/r03_isa_notes.md -- and the wait is kept because it is free)
        __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0): the next stage has landed
        __syncthreads();
    }

#ifdef MYDET_DIAG
    if ((p.dbg & 8) && tid == 0) {                     // shader cycles / 100 MHz ticks of the K loop, per workgroup
        int *d = reinterpret_cast<int *>(p.v) + ((int64_t)mb * nk) * (V_BYTES / 4) + nb * 4;
        d[0] = (int)(__builtin_amdgcn_s_memtime() - dbg_t0);
        d[1] = (int)(__builtin_amdgcn_s_memrealtime() - dbg_r0);
    }
#endif
    // ---- epilogue: output transform Y = At M A in place on the accumulators: the column pass leaves
    // s[a][m] = sum_i At[a][i] M[i][m] in acc[6a + m], the row pass takes out[c] = sum_m s[a][m] At[c][m]
    const int nc = nok ? n : 0;
    const f32x4 scl = p.scale ? *reinterpret_cast<const f32x4 *>(p.scale + nc) : f32x4{1.f, 1.f, 1.f, 1.f};
    const f32x4 sft = p.shift ? *reinterpret_cast<const f32x4 *>(p.shift + nc) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < 6; ++m) {
        const f32x4 p1 = acc[6 + m] + acc[12 + m], m1 = acc[6 + m] - acc[12 + m];
        const f32x4 p2 = acc[18 + m] + acc[24 + m], m2 = acc[18 + m] - acc[24 + m];
        acc[m] = acc[m] + p1 + p2;
        acc[6 + m] = m1 + 2.0f * m2;
        acc[12 + m] = p1 + 4.0f * p2;
        acc[18 + m] = m1 + 8.0f * m2 + acc[30 + m];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const f32x4 pp = acc[6 * a + 1] + acc[6 * a + 2], qq = acc[6 * a + 3] + acc[6 * a + 4];
        const f32x4 dd = acc[6 * a + 1] - acc[6 * a + 2], ee = acc[6 * a + 3] - acc[6 * a + 4];
        f32x4 out[4];
        out[0] = acc[6 * a] + pp + qq;
        out[1] = dd + 2.0f * ee;
        out[2] = pp + 4.0f * qq;
        out[3] = dd + 8.0f * ee + acc[6 * a + 5];
        if (PART) {                                    // raw output-domain partial tile; scale / act / residual in the fixup
            f32x4 *dst = reinterpret_cast<f32x4 *>(p.part) + ((int64_t)slot * p.splits + piece) * (16 * 64 * NW) + tid;
#pragma unroll
            for (int c = 0; c < 4; ++c) dst[(4 * a + c) * (64 * NW)] = out[c];
            continue;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            f32x4 v = out[c] * scl + sft;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (ACT == MYDET_ACT_LEAKY) v[e] = v[e] > 0.0f ? v[e] : v[e] * 0.1f;
                if (ACT == MYDET_ACT_SWISH) v[e] = v[e] * mydet_sigmoid_fast(v[e]);
            }
            if (RES) v += rv[4 * a + c];
            // the pixel offset goes into the voffset, NOT the scalar offset: with an SGPR soffset hipcc leaves no wait
            // state between a 16-byte store and the VALU that next overwrites its data registers (LLVM pads that hazard
            // only for stores without an SGPR soffset; listing: profiles/r03_isa_notes.md), and on MI355X such a store
            // picked up part of the next pixel's values about once in eight launches (test_conv_winograd4_repeatable;
            // tools/check_store_hazard.py proves the pattern absent from the built library)
            const unsigned yo = ybase + (unsigned)((a * p.W + c) * p.ldy * 4);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yr, orow[a] && ocol[c] ? yo : OOB, 0, 0);
        }
    }
}

// ---- 3. K-cut tail: sums the pieces of one tail item in K order and applies the epilogue.  blockIdx.x = id slot of the
// tail (padded ids return), blockIdx.y = output row a of the 4x4 tiles; a thread has the role it had in the GEMM kernel
// (same tile, same four channels).  Everything is requested before anything is waited for.
template <int ACT, bool RES>
__global__ __launch_bounds__(64 * NW) void wino4_fixup_kernel(const W4Args p) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int blk = (int)blockIdx.x / p.tail_stride;
    const int id = p.tail_id0 + blk * 64 + ((int)blockIdx.x - blk * p.tail_stride), a = blockIdx.y;
    const int bi = id >> 6, w = id & 63;
    const int mb = (bi / p.nbn) * (64 >> p.rn_log2) + (w >> p.rn_log2);
    const int nb = (bi % p.nbn) * (1 << p.rn_log2) + (w & ((1 << p.rn_log2) - 1));
    if (mb >= p.nmb || nb >= p.ntn) return;
    const int tpi = p.TH * p.TW;
    const int m0 = mb * TILES, n0 = nb * CH, b0 = m0 / tpi;
    const int wc = wave % (NW / 2), wt = wave / (NW / 2), fr = lane & 15, fq = lane >> 4;
    const int n = n0 + wc * 16 + fq * 4;
    const bool nok = n < p.Cout;
    const int nc = nok ? n : 0;
    const f32x4 scl = p.scale ? *reinterpret_cast<const f32x4 *>(p.scale + nc) : f32x4{1.f, 1.f, 1.f, 1.f};
    const f32x4 sft = p.shift ? *reinterpret_cast<const f32x4 *>(p.shift + nc) : f32x4{0.f, 0.f, 0.f, 0.f};
    const int64_t oimg = (int64_t)p.H * p.W;
    const __amdgpu_buffer_rsrc_t yr = rsrc(p.y + b0 * oimg * p.ldy, (p.B - b0) * oimg * p.ldy * 4);
    const __amdgpu_buffer_rsrc_t rr = rsrc(RES ? p.res + b0 * oimg * p.ldr : p.y, (p.B - b0) * oimg * (RES ? p.ldr : p.ldy) * 4);
    const int mt = m0 + wt * 16 + fr;
    const int mm = mt < p.MT ? mt : p.MT - 1;
    const int b = mm / tpi, r = mm - b * tpi, ty = r / p.TW, tx = r - ty * p.TW;
    const int oy = 4 * ty + a, ox = 4 * tx;
    const bool rok = mt < p.MT && nok && oy < p.H;
    const int64_t pix = ((int64_t)(b - b0) * p.H + oy) * p.W + ox;
    unsigned yo[4];
    f32x4 rv[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const bool ok = rok && ox + c < p.W;
        yo[c] = ok ? (unsigned)(((pix + c) * p.ldy + n) * 4) : OOB;
        if (RES) rv[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rr, ok ? (unsigned)(((pix + c) * p.ldr + n) * 4) : OOB, 0, 0));
    }
    const f32x4 *src = reinterpret_cast<const f32x4 *>(p.part) + (int64_t)blockIdx.x * p.splits * (16 * 64 * NW) + (4 * a) * (64 * NW) + tid;
    f32x4 t[4][8];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int s = 0; s < 8; ++s)
            if (s < p.splits) t[c][s] = src[(int64_t)s * (16 * 64 * NW) + c * (64 * NW)];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        f32x4 sum = t[c][0];
#pragma unroll
        for (int s = 1; s < 8; ++s)
            if (s < p.splits) sum += t[c][s];          // K order
        f32x4 v = sum * scl + sft;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (ACT == MYDET_ACT_LEAKY) v[e] = v[e] > 0.0f ? v[e] : v[e] * 0.1f;
            if (ACT == MYDET_ACT_SWISH) v[e] = v[e] * mydet_sigmoid_fast(v[e]);
        }
        if (RES) v += rv[c];
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yr, yo[c], 0, 0);
    }
}

// U = G g Gt in float64, rounded once; layout [Cin/4][9 position groups][4 k][CoutP][4] with the float4 = positions
// 4g..4g+3 at channel k (CoutP = Cout rounded up to 64, zero rows): exactly the LDS slab, so a workgroup's share of a
// slab is 36 contiguous runs of 64 float4.
__global__ void wino4_weights_kernel(const float *w, int Cout, int Cin, int CoutP, float *u) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)CoutP * Cin) return;
    const int n = (int)(i / Cin), c = (int)(i - (int64_t)n * Cin);
    const double G[6][3] = {{0.25, 0.0, 0.0},          {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                            {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0.0, 0.0, 1.0}};
    double g[3][3], t[6][3];
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) g[a][b] = n < Cout ? (double)w[(((int64_t)n * 3 + a) * 3 + b) * Cin + c] : 0.0;
    for (int a = 0; a < 6; ++a)
        for (int b = 0; b < 3; ++b) t[a][b] = G[a][0] * g[0][b] + G[a][1] * g[1][b] + G[a][2] * g[2][b];
    const int ks = c >> 2, kq = c & 3;
    for (int a = 0; a < 6; ++a)
        for (int b = 0; b < 6; ++b) {
            const double v = t[a][0] * G[b][0] + t[a][1] * G[b][1] + t[a][2] * G[b][2];
            const int q = a * 6 + b;
            u[((((int64_t)ks * NPG + (q >> 2)) * KC + kq) * CoutP + n) * 4 + (q & 3)] = (float)v;
        }
}

// K-cut tail plan (host only; also exported as mydet_wino4_tail_plan so that the rule is testable without a GPU).
// Workgroups are equal, the chip holds `slots` of them, so T items cost ceil(T / slots) rounds: 1600 items (256->512 @40^2,
// batch 32) pay four rounds for 3.125.  The last blocks of the item order (the fewest that bring the rest down to whole
// rounds) are instead cut along K in up to two groups -- each about one short round -- and summed by small launches: 64 items
// 8 ways @40^2, 256 items 2 ways + 32 items 8 ways @20^2.  Not when K is short, the groups would cost about the round they
// replace, or there are more than four whole rounds.
struct TailGroup { int id0, blocks, stride, splits; int64_t part_bytes; };

// Tuning knobs of the tail plan, read from the environment ONCE per process (they used to be four getenv + atoi per
// F(4x4) layer per eager call, and a graph capture froze whatever they said at that moment anyway):
//   MYDET_W4_TAIL         minimum cut count of the last group (default 4), 0 = no tail
//   MYDET_W4_TAIL_GROUPS  most groups (default 2)      MYDET_W4_TAIL_MAXR  most whole rounds before a tail (default 4)
//   MYDET_W4_TAIL_COST    largest admissible cost in rounds (default 0.9)
// mydet_wino4_reload_tuning() reads them again (tests and sweeps that change them inside one process).
struct TailKnobs { int min_cuts, max_groups, max_rounds; double max_cost; };
TailKnobs read_tail_knobs() {
    const char *te = getenv("MYDET_W4_TAIL"), *ge = getenv("MYDET_W4_TAIL_GROUPS");
    const char *re = getenv("MYDET_W4_TAIL_MAXR"), *ce = getenv("MYDET_W4_TAIL_COST");
    return TailKnobs{te ? atoi(te) : 4, ge ? atoi(ge) : 2, re ? atoi(re) : 4, ce ? atof(ce) : 0.9};
}
TailKnobs &tail_knobs() {
    static TailKnobs k = read_tail_knobs();
    return k;
}

int plan_tail(int nmb, int ntn, int nbn, int rn_log2, int Cin, int slots, TailGroup *groups, int64_t *scratch_bytes = nullptr) {
    int ngroups = 0;
    if (scratch_bytes) *scratch_bytes = 0;
    const TailKnobs &knobs = tail_knobs();
    const int tail_on = knobs.min_cuts, max_groups = knobs.max_groups;
    const int64_t T = (int64_t)nmb * ntn;
    const int64_t whole = T / slots * slots;
    const int RM = 64 >> rn_log2, RN = 1 << rn_log2, nk = Cin >> 2;
    const int64_t nbm = (nmb + RM - 1) / RM;
    const int64_t NB = nbm * nbn;
    auto items_of = [&](int64_t blk) {
        const int64_t row = blk / nbn, col = blk % nbn;
        const int64_t rows = (row + 1) * RM <= nmb ? RM : nmb - row * RM;
        const int64_t cols = (col + 1) * RN <= ntn ? RN : ntn - col * RN;
        return rows * cols;
    };
    auto stride_of = [&](int64_t blk) {                // valid ids of a block are w = r * RN + c, r < rows: a prefix when no column is cut
        const int64_t row = blk / nbn;
        const int64_t rows = (row + 1) * RM <= nmb ? RM : nmb - row * RM;
        return (int)(rows * RN);
    };
    // (after many whole rounds the workgroups no longer finish together and the last partial round is cheap already:
    // 128->256 @80^2, 6.25 rounds, ran 0.6 % of the headline FASTER without its tail -- up to four whole rounds only)
    if (!(tail_on > 0 && whole > 0 && T > whole && nk >= 8 && whole / slots <= knobs.max_rounds)) return 0;
    int64_t cut = 0;
    int64_t first = NB;                                // first block of the tail: the fewest blocks that leave whole rounds
    while (T - cut > whole && first > 0) cut += items_of(--first);
    // groups in id order.  While what is left is more than half a round, a group of up to half a round is cut two ways;
    // the rest is one group cut slots / items ways (at most 8, at least 4 K stages per piece).
    int64_t blk = first, left = cut, used = 0;
    double cost = 0.0;                                 // in rounds: 1 / splits per group + ~0.12 for its two launches
    while (left > 0) {
        if (ngroups == 3 || ngroups == max_groups) return 0;
        int64_t n = 0;
        int nb = 0, stride = 0;
        const bool last = left * 2 <= slots;
        while (blk + nb < NB && (last || n + items_of(blk + nb) <= slots / 2)) {
            n += items_of(blk + nb);
            stride = stride_of(blk + nb) > stride ? stride_of(blk + nb) : stride;
            ++nb;
        }
        int splits = last ? (int)(slots / (n > 0 ? n : 1)) : 2;
        splits = splits > 8 ? 8 : splits;
        if (splits > nk / 4) splits = nk / 4;
        if (n == 0 || splits < 2 || (last && splits < tail_on)) return 0;
        groups[ngroups] = TailGroup{(int)(blk * 64), nb, stride, splits, used};
        used += (int64_t)nb * stride * splits * (16 * 64 * NW * 16);
        if (used > TAIL_BYTES) return 0;
        cost += 1.0 / splits + 0.12;
        ++ngroups;
        blk += nb; left -= n;
    }
    // (0.9 admits a two-way group + an eight-way group = 0.865: 512->1024 @20^2, 800 items = one round + 256 + 32)
    if (cost > knobs.max_cost) return 0;
    if (scratch_bytes) *scratch_bytes = used;
    return ngroups;
}

template <int ACT, bool RES>
int launch_w4(W4Args a, const W4Args *groups, int ngroups, hipStream_t stream) {
    static unsigned long long attr_set = 0, attr_set_p = 0;     // > 64 KiB of dynamic LDS needs the opt-in once per device
    if (const int e = mydet_lds_opt_in(attr_set, &conv_wino4_kernel<ACT, RES>, LDS_BYTES)) return e;
    if (const int e = mydet_lds_opt_in(attr_set_p, &conv_wino4_kernel<MYDET_ACT_NONE, false, true>, LDS_BYTES)) return e;
    hipLaunchKernelGGL(wino4_input_kernel, dim3((a.MT + TILES - 1) / TILES, (a.Cin + 31) / 32), dim3(256), 0, stream, a);
    if (ngroups > 0) {
        // whole rounds of whole items, then the remainder in groups cut along K (each about one short round), then their sums
        W4Args m = a;
        m.nblk = groups[0].tail_id0;                   // (the XCD remap of the main launch runs over its own grid)
        hipLaunchKernelGGL((conv_wino4_kernel<ACT, RES>), dim3(m.nblk), dim3(64 * NW), LDS_BYTES, stream, m);
        for (int g = 0; g < ngroups; ++g) {
            const W4Args &t = groups[g];
            hipLaunchKernelGGL((conv_wino4_kernel<MYDET_ACT_NONE, false, true>), dim3(t.tail_blocks * t.tail_stride * t.splits),
                               dim3(64 * NW), LDS_BYTES, stream, t);
        }
        for (int g = 0; g < ngroups; ++g) {
            const W4Args &t = groups[g];
            hipLaunchKernelGGL((wino4_fixup_kernel<ACT, RES>), dim3(t.tail_blocks * t.tail_stride, 4), dim3(64 * NW), 0, stream, t);
        }
        return mydet_launch_status();
    }
    hipLaunchKernelGGL((conv_wino4_kernel<ACT, RES>), dim3(a.nblk), dim3(64 * NW), LDS_BYTES, stream, a);
    return mydet_launch_status();
}

}  // namespace

extern "C" int mydet_wino4_reload_tuning(void) {
    tail_knobs() = read_tail_knobs();
    return 0;
}

extern "C" int64_t mydet_wino4_weights_floats(int Cout, int Cin) {
    if (Cout <= 0 || Cin <= 0 || (Cin & 3)) return 0;
    return (int64_t)36 * Cin * ((Cout + COUT_PAD - 1) / COUT_PAD * COUT_PAD);
}

extern "C" int mydet_wino4_weights_f32(const float *w, int Cout, int Cin, float *u, void *stream) {
    if (!w || !u || Cout <= 0 || Cin <= 0) return MYDET_E_BADARG;
    if (Cin & 3) return MYDET_E_UNSUPP;
    const int CoutP = (Cout + COUT_PAD - 1) / COUT_PAD * COUT_PAD;
    const int64_t n = (int64_t)CoutP * Cin;
    hipLaunchKernelGGL(wino4_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, Cout, Cin,
                       CoutP, u);
    return mydet_launch_status();
}

namespace {
// The launch geometry both the workspace size and the launch derive from the shape: item grid and tail plan.
struct W4Geom { int ntn, nmb, rn_log2, nbn; int64_t nbm, v_bytes, tail_bytes; int ngroups; TailGroup plan[3]; };
int w4_geometry(int B, int H, int W, int Cin, int Cout, int slots, W4Geom &g) {
    const int64_t MT = (int64_t)B * ((H + 3) / 4) * ((W + 3) / 4);
    if (MT > (int64_t)1 << 30) return MYDET_E_UNSUPP;
    g.ntn = (Cout + CH - 1) / CH;
    g.nmb = (int)((MT + TILES - 1) / TILES);
    g.rn_log2 = 0;
    while ((1 << g.rn_log2) < g.ntn && g.rn_log2 < 3) ++g.rn_log2;
    g.nbn = (g.ntn + (1 << g.rn_log2) - 1) >> g.rn_log2;
    g.nbm = (g.nmb + (64 >> g.rn_log2) - 1) / (64 >> g.rn_log2);
    if (g.nbm * g.nbn * 64 > 0x7FFFFFFF) return MYDET_E_UNSUPP;
    g.v_bytes = (MT + TILES - 1) / TILES * (Cin >> 2) * V_BYTES;
    g.ngroups = plan_tail(g.nmb, g.ntn, g.nbn, g.rn_log2, Cin, slots, g.plan, &g.tail_bytes);
    return 0;
}
}  // namespace

extern "C" int64_t mydet_wino4_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || (Cin & 3)) return 0;
    W4Geom g;
    if (w4_geometry(B, H, W, Cin, Cout, 2 * mydet_cu_count(), g) != 0) return 0;
    return g.v_bytes + g.tail_bytes;            // V, then the partial tiles of the K-cut tail where the plan has one
}

/* Test hook (host only, no GPU call): the K-cut tail plan mydet_conv2d_wino4_f32 would use on a chip of `slots` resident
 * workgroups.  out[0] = ids of the main launch, out[1] = groups, then per group {first id, blocks, ids per block, cuts,
 * scratch offset in KiB}; returns the number of groups or a negative MYDET_E_*. */
extern "C" int mydet_wino4_tail_plan(int B, int H, int W, int Cin, int Cout, int slots, int32_t *out) {
    if (!out || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || slots <= 0 || (Cin & 3) || (Cout & 3)) return MYDET_E_BADARG;
    W4Geom geo;
    if (const int e = w4_geometry(B, H, W, Cin, Cout, slots, geo)) return e;
    const TailGroup *plan = geo.plan;
    const int ng = geo.ngroups;
    const int64_t nbm = geo.nbm;
    const int nbn = geo.nbn;
    out[0] = ng ? plan[0].id0 : (int)(nbm * nbn * 64);
    out[1] = ng;
    for (int g = 0; g < ng; ++g) {
        out[2 + 5 * g] = plan[g].id0; out[3 + 5 * g] = plan[g].blocks; out[4 + 5 * g] = plan[g].stride;
        out[5 + 5 * g] = plan[g].splits; out[6 + 5 * g] = (int)(plan[g].part_bytes >> 10);
    }
    return ng;
}

extern "C" int mydet_conv2d_wino4_f32(const float *x, int64_t ldx, const float *u, const float *scale, const float *shift,
                                      const float *residual, int64_t ldr, float *ws, int64_t ws_bytes, float *y, int64_t ldy,
                                      int B, int H, int W, int Cin, int Cout, int act, void *stream) {
    if (!x || !u || !y || !ws || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || act < 0 || act > 2) return MYDET_E_BADARG;
    if ((ldx & 3) || ldx < Cin || ldy < Cout || (residual && ldr < Cout)) return MYDET_E_BADARG;
    if (((uintptr_t)x & 15) || ((uintptr_t)u & 15) || ((uintptr_t)y & 15) || ((uintptr_t)ws & 15) ||
        (residual && ((uintptr_t)residual & 15)) || (scale && ((uintptr_t)scale & 15)) || (shift && ((uintptr_t)shift & 15)))
        return MYDET_E_BADARG;
    if ((Cin & 3) || (Cout & 3) || (ldy & 3) || (residual && (ldr & 3))) return MYDET_E_UNSUPP;
    W4Args a;
    a.x = x; a.u = u; a.scale = scale; a.shift = shift; a.res = residual; a.y = y; a.v = ws;
    a.ldx = ldx; a.ldr = residual ? ldr : ldy; a.ldy = ldy;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.CoutP = (Cout + COUT_PAD - 1) / COUT_PAD * COUT_PAD;
    a.TH = (H + 3) / 4; a.TW = (W + 3) / 4;
    const int64_t MT = (int64_t)B * a.TH * a.TW;
    if (MT > (int64_t)1 << 30) return MYDET_E_UNSUPP;
    // 32-bit byte offsets inside a workgroup's window: the images its 32 tiles touch, its stages of V, all of U
    const int64_t span = TILES / ((int64_t)a.TH * a.TW) + 2;
    const int64_t ldmax = ldx > ldy ? (ldx > a.ldr ? ldx : a.ldr) : (ldy > a.ldr ? ldy : a.ldr);
    if ((int64_t)H * W * ldmax * 4 * span >= 0x7FFFFFF0ll || (int64_t)36 * Cin * a.CoutP * 4 >= 0x7FFFFFF0ll) return MYDET_E_UNSUPP;
    a.MT = (int)MT;
#ifdef MYDET_DIAG
    { const char *e = getenv("MYDET_W4_DBG"); a.dbg = e && *e ? atoi(e) : 0; }
#endif
    W4Geom geo;
    if (const int e = w4_geometry(B, H, W, Cin, Cout, 2 * mydet_cu_count(), geo)) return e;
    if (ws_bytes < geo.v_bytes + geo.tail_bytes) return MYDET_E_BADARG;
    a.ntn = geo.ntn; a.nmb = geo.nmb; a.rn_log2 = geo.rn_log2; a.nbn = geo.nbn;
    a.nblk = (int)(geo.nbm * geo.nbn * 64);
    if ((MT + TILES - 1) / TILES > 0x7FFFFFFF || (Cin + 31) / 32 > 65535) return MYDET_E_UNSUPP;
    // K-cut tail (plan_tail): its partial tiles live behind V
    a.tail_id0 = a.nblk; a.tail_blocks = 0; a.tail_stride = 64; a.splits = 1;
    a.part = ws + geo.v_bytes / 4;
    const TailGroup *plan = geo.plan;
    const int ngroups = geo.ngroups;
    W4Args groups[3];
    for (int g = 0; g < ngroups; ++g) {
        groups[g] = a;
        groups[g].tail_id0 = plan[g].id0; groups[g].tail_blocks = plan[g].blocks; groups[g].tail_stride = plan[g].stride;
        groups[g].splits = plan[g].splits; groups[g].part = a.part + plan[g].part_bytes / 4;
    }
    hipStream_t s = (hipStream_t)stream;
    const bool res = residual != nullptr;
    switch (act) {
        case MYDET_ACT_LEAKY: return res ? launch_w4<MYDET_ACT_LEAKY, true>(a, groups, ngroups, s) : launch_w4<MYDET_ACT_LEAKY, false>(a, groups, ngroups, s);
        case MYDET_ACT_SWISH: return res ? launch_w4<MYDET_ACT_SWISH, true>(a, groups, ngroups, s) : launch_w4<MYDET_ACT_SWISH, false>(a, groups, ngroups, s);
        default: return res ? launch_w4<MYDET_ACT_NONE, true>(a, groups, ngroups, s) : launch_w4<MYDET_ACT_NONE, false>(a, groups, ngroups, s);
    }
}
