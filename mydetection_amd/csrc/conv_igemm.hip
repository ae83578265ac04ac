// Dense convolution as implicit GEMM on the gfx950 FP32 matrix cores.
//
//   M = B*Ho*Wo output pixels, N = Cout, K = KH*KW*Cin (tap-major, channel-minor)
//   A[m][k] = x[b, oh*s - pad_t + kh, ow*s - pad_l + kw, c]   (gathered on the fly, zero outside)
//   B[n][k] = w[n][kh][kw][c]                                  (OHWI, K contiguous)
//   y[m][n] = act(acc*scale[n] + shift[n]) + residual[m][n]
//
// One 256-thread workgroup (4 waves) owns a BM x BN output tile; each wave a
// (BM/WM) x (BN/WN) sub-tile built from 32x32 v_mfma_f32_32x32x2_f32 blocks, i.e. exact
// float32 products and a k-ordered float32 fma chain (no reduced-precision path exists
// on gfx950).  K is walked in slabs of 32.
//
// Staging: global -> registers -> LDS with 16-byte accesses.  All global reads are
// buffer loads through wave-uniform descriptors with per-lane byte offsets: padding
// taps / rows past M get offset 0xFFFFFFFF and the hardware range check returns zeros,
// so the K loop has NO data-dependent branches -- one basic block in which the compiler
// interleaves address VALU, buffer loads, LDS traffic and the 64 MFMAs per wave.
// Tap validity of the <= 32 taps is a per-row bitmask built once.  LDS rows are padded to
// 36 floats: fragment ds_read_b128 and staging ds_write_b128 are bank-conflict free.
// Two LDS buffers, one barrier per slab; the next slab's loads fly under the MFMAs.
// Each lane reads 4 consecutive k of its row with one ds_read_b128; lane half h owns
// k = 8*kc + 4*h + t at MFMA step t -- a permutation of k applied to A and B alike.
//
// Replaces the ATen conv2d/batch_norm/leaky_relu chain under models/modules.py:94-95,
// the residual add of models/modules.py:69-73 and the head convs models/rpns.py:24-25.
#include <cstdlib>

#include "common.h"

namespace {

constexpr unsigned OOB = 0xFFFFFFFFu;

struct ConvArgs {
    const float *x, *w, *scale, *shift, *res, *gate;
    float *y;
    int64_t ldx, ldr, ldy;
    int B, H, W, Cin, Cout, KH, KW, stride, pad_t, pad_l, Ho, Wo, act;
    int M, K, ntiles, nblk;
    int tile0, splits;        // split-K tail: first logical tile of this launch, K slices per tile (1 = whole K)
    float *ws;                // split-K partial accumulators
    size_t ws_bytes;
    // CAT (1x1 only): channels [0, C1) of A come from x1 = a [B, H/2, W/2, ldx1] map read through nearest 2x upsampling, the
    // remaining Cin - C1 from x -- conv1x1(cat((up2x(x1), x), 1)) without the concatenated tensor
    const float *x1;
    int64_t ldx1;
    int C1;
    // split-bf16 form (conv_igemm_b3_kernel): the weights as three bfloat16 planes [3][Cout][K] with w = p0 + p1 + p2
    const unsigned short *wsplit;
};

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 buf_load16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0));
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float *base, int64_t bytes) {
    const uint64_t a = (uint64_t)base;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    const int64_t capped = bytes > 0x7FFFFFF0ll ? 0x7FFFFFF0ll : bytes;
    const int n = __builtin_amdgcn_readfirstlane((int)capped);
    return __builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)hi << 32) | lo), 0, n, 0x00020000);
}

// Epilogue: lane = output channel, register = output pixel.  Stores (and residual loads) are
// buffer ops whose per-lane offset is fixed and whose row term is a scalar: one v_add per element.
// Ragged tiles route out-of-range elements to offset 0xFFFFFFFF, which the range check drops.
template <int ACT, bool RES, bool FULL, int TM, int TN>
__device__ __forceinline__ void epilogue(const ConvArgs &p, f32x16 (&acc)[TM][TN], int m_base, int n_base,
                                         int fr, int fh, const float (&pscl)[TN], const float (&psft)[TN]) {
    const __amdgpu_buffer_rsrc_t yr = make_rsrc(p.y, (int64_t)p.M * p.ldy * 4);
    const __amdgpu_buffer_rsrc_t rr = make_rsrc(RES ? p.res : p.y, (int64_t)p.M * (RES ? p.ldr : p.ldy) * 4);
    const unsigned ldy4 = (unsigned)p.ldy * 4u, ldr4 = (unsigned)p.ldr * 4u;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n_base + j * 32 + fr;
        const bool nok = FULL || n < p.Cout;
        const float scl = pscl[j], sft = psft[j];      // requested before the K loop (a round trip per workgroup otherwise)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mrow = m_base + i * 32 + 4 * fh;          // + (r&3) + 8*(r>>2)
            const unsigned ybase = (unsigned)mrow * ldy4 + (unsigned)n * 4u;
            const unsigned rbase = (unsigned)mrow * ldr4 + (unsigned)n * 4u;
            float rv[16];
            if (RES) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dr = (r & 3) + 8 * (r >> 2);
                    const bool ok = FULL || (nok && mrow + dr < p.M);
                    rv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                          rr, ok ? rbase + (unsigned)dr * ldr4 : OOB, 0, 0));
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dr = (r & 3) + 8 * (r >> 2);
                float v = acc[i][j][r] * scl + sft;
                if (ACT == MYDET_ACT_LEAKY) v = v > 0.0f ? v : v * 0.1f;
                if (ACT == MYDET_ACT_SWISH) v = v * mydet_sigmoid_fast(v);
                if (RES) v += rv[r];
                const bool ok = FULL || (nok && mrow + dr < p.M);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yr,
                                                      ok ? ybase + (unsigned)dr * ldy4 : OOB, 0, 0);
            }
        }
    }
}

template <int BM, int BN, int WM, int WN, int BK, bool CIN32, int ACT, bool RES, bool GATE, bool SPLIT = false, bool CAT = false>
__global__ __launch_bounds__(WM * WN * 64) void conv_igemm_kernel(const ConvArgs p) {
    constexpr int NT = WM * WN * 64;                // threads: one wave per (wm, wn)
#ifndef IGEMM_PF
#define IGEMM_PF 2
#endif
    constexpr int PF = IGEMM_PF;                    // slabs prefetched into registers beyond the one staged in LDS
    constexpr int LDS_LD = BK + 4;                 // padded LDS row (floats)
    constexpr int CH = BK / 4;                      // 16-byte chunks per slab row
    constexpr int RP = NT / CH;                     // rows staged per pass of the workgroup
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int AI = BM / RP, BI = BN / RP;      // 16-byte chunks each thread stages per slab
    static_assert(BM % RP == 0 && BN % RP == 0, "tile rows must fill whole staging passes");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As = smem;                               // [2][BM][LDS_LD]
    float *Bs = smem + 2 * BM * LDS_LD;             // [2][BN][LDS_LD]

    const int tid = threadIdx.x;
    // SPLIT: block = (tail tile, K slice); otherwise one block per tile, XCD-grouped
    const int lid = SPLIT ? p.tile0 + (int)blockIdx.x / p.splits : mydet_xcd_remap(blockIdx.x, p.nblk);
    const int m0 = (lid / p.ntiles) * BM;
    const int n0 = (lid % p.ntiles) * BN;

    // ---- descriptors: A window starts at the image holding the tile's first row
    const int hwo = p.Ho * p.Wo;
    const int b0 = m0 / hwo;
    const int64_t img = (int64_t)p.H * p.W * p.ldx;
    const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x + b0 * img, (p.B - b0) * img * 4);
    const __amdgpu_buffer_rsrc_t wr = make_rsrc(p.w, (int64_t)p.Cout * p.K * 4);
    const int64_t img1 = (int64_t)(p.H >> 1) * (p.W >> 1) * p.ldx1;      // CAT: the half-resolution source
    const __amdgpu_buffer_rsrc_t xr1 = make_rsrc(CAT ? p.x1 + b0 * img1 : p.w, CAT ? (p.B - b0) * img1 * 4 : 16);
    // SE gate (1x1 convs only): A[m][k] is multiplied by gate[image(m)][k] while it is staged
    const __amdgpu_buffer_rsrc_t gr = make_rsrc(GATE ? p.gate : p.w, (int64_t)p.B * p.Cin * 4);

    // ---- staging role: chunk (4 floats) `sc` of rows sr + RP*i
    const int sc = tid % CH, sr = tid / CH;
    const int ntaps = p.KH * p.KW;
    int aoff[AI];                                   // byte offset of tap (0,0), channel 4*sc, from the window base
    unsigned amask[AI];                             // bit t: tap t of this row is inside the image
    unsigned gbase[AI];                             // GATE: byte offset of the row's image in gate[B][Cin]
    unsigned aoff1[AI];                             // CAT: byte offset of the row's pixel (oh / 2, ow / 2) in x1, channel 4*sc
    // pointwise, stride 1, unpadded (uniform): input pixel == output pixel, no index arithmetic beyond the row number (the
    // three divisions per row below are pure latency in a small-grid launch)
    const bool flat = !CAT && ntaps == 1 && p.stride == 1 && p.pad_t == 0 && p.pad_l == 0 && p.Ho == p.H && p.Wo == p.W;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const int m = m0 + sr + RP * i;
        const int mm = m < p.M ? m : p.M - 1;
        if (flat) {
            aoff[i] = (int)(((int64_t)(mm - b0 * hwo) * p.ldx + sc * 4) * 4);
            amask[i] = m < p.M ? 1u : 0u;
            if (GATE) gbase[i] = (unsigned)((mm / hwo) * p.Cin) * 4u;
            continue;
        }
        const int ow = mm % p.Wo, t = mm / p.Wo;
        const int oh = t % p.Ho, b = t / p.Ho;
        const int ih0 = oh * p.stride - p.pad_t, iw0 = ow * p.stride - p.pad_l;
        aoff[i] = (int)((((int64_t)(b - b0) * p.H + ih0) * p.W + iw0) * p.ldx + sc * 4) * 4;
        unsigned mask = 0;
        for (int tp = 0; tp < ntaps; ++tp) {
            const int kh = tp / p.KW, kw = tp - kh * p.KW;
            if ((unsigned)(ih0 + kh) < (unsigned)p.H && (unsigned)(iw0 + kw) < (unsigned)p.W) mask |= 1u << tp;
        }
        amask[i] = m < p.M ? mask : 0u;
        if (GATE) gbase[i] = (unsigned)(b * p.Cin) * 4u;
        if (CAT) aoff1[i] = (unsigned)((((int64_t)(b - b0) * (p.H >> 1) + (oh >> 1)) * (p.W >> 1) + (ow >> 1)) * p.ldx1 + sc * 4) * 4u;
    }
    unsigned boff[BI];
#pragma unroll
    for (int i = 0; i < BI; ++i) {
        int n = n0 + sr + RP * i;
        n = n < p.Cout ? n : p.Cout - 1;            // rows past Cout compute garbage that is never stored
        boff[i] = (unsigned)(n * p.K + sc * 4) * 4u;
    }

    f32x4 areg[PF][AI], breg[PF][BI];                // PF slabs in flight (see the K loop)
    const int nk_all = (p.K + BK - 1) / BK;
    int kt0 = 0, nk = nk_all;                        // this block's slab range [kt0, nk)
    if (SPLIT) {
        const int sp = (int)blockIdx.x % p.splits;
        kt0 = (int)((unsigned)(nk_all * sp) / (unsigned)p.splits);          // (nk_all * splits < 2^31: no 64-bit division)
        nk = (int)((unsigned)(nk_all * (sp + 1)) / (unsigned)p.splits);
    }
    int tap = 0, c0 = 0, tapoff = 0;                 // CIN32 path: uniform tap / channel base / byte offset
    if (SPLIT && CIN32) {
        const int k0 = kt0 * BK;
        tap = k0 / p.Cin;
        c0 = k0 - tap * p.Cin;
        const int kh = tap / p.KW, kw = tap - kh * p.KW;
        tapoff = (int)(((int64_t)kh * p.W + kw) * p.ldx) * 4;
    }

    auto load_slab = [&](int kt, f32x4 (&areg)[AI], f32x4 (&breg)[BI]) {
        if (CIN32) {
            const bool lo = CAT && c0 < p.C1;              // uniform: this slab's channels come from the upsampled map
            const unsigned uoff = (unsigned)(tapoff + (c0 - (CAT ? p.C1 : 0)) * 4);
#pragma unroll
            for (int i = 0; i < AI; ++i) {
                const bool ok = (amask[i] >> tap) & 1u;
                if (lo) areg[i] = buf_load16(xr1, ok ? aoff1[i] + (unsigned)(c0 * 4) : OOB);
                else areg[i] = buf_load16(xr, ok ? (unsigned)aoff[i] + uoff : OOB);
                if (GATE) areg[i] *= buf_load16(gr, gbase[i] + (unsigned)(c0 + sc * 4) * 4u);
            }
            c0 += BK;
            if (c0 == p.Cin) {                        // uniform: next tap
                c0 = 0;
                ++tap;
                const int kh = tap / p.KW, kw = tap - kh * p.KW;
                tapoff = (int)(((int64_t)kh * p.W + kw) * p.ldx) * 4;
            }
        } else {
            const int k = kt * BK + sc * 4;          // Cin % 4 == 0: a chunk never straddles taps
            const int tp = k / p.Cin;
            const int cc = k - tp * p.Cin;
            const int kh = tp / p.KW, kw = tp - kh * p.KW;
            const unsigned uoff = (unsigned)((int)(((int64_t)kh * p.W + kw) * p.ldx + cc - sc * 4) * 4);   // aoff holds +4*sc
            const bool kok = k < p.K;
#pragma unroll
            for (int i = 0; i < AI; ++i) {
                const bool ok = kok && ((amask[i] >> (tp & 31)) & 1u);
                areg[i] = buf_load16(xr, ok ? (unsigned)aoff[i] + uoff : OOB);
                if (GATE) areg[i] *= buf_load16(gr, kok ? gbase[i] + (unsigned)cc * 4u : OOB);
            }
        }
        const unsigned koff = (unsigned)kt * (BK * 4);
#pragma unroll
        for (int i = 0; i < BI; ++i) breg[i] = buf_load16(wr, boff[i] + koff);
    };
    auto store_slab = [&](int buf, const f32x4 (&areg)[AI], const f32x4 (&breg)[BI]) {
        float *a = As + buf * BM * LDS_LD, *b = Bs + buf * BN * LDS_LD;
#pragma unroll
        for (int i = 0; i < AI; ++i)
            *reinterpret_cast<f32x4 *>(a + (sr + RP * i) * LDS_LD + sc * 4) = areg[i];
#pragma unroll
        for (int i = 0; i < BI; ++i)
            *reinterpret_cast<f32x4 *>(b + (sr + RP * i) * LDS_LD + sc * 4) = breg[i];
    };

    // ---- compute role
    const int wave = tid >> 6, lane = tid & 63;
    const int wm = wave / WN, wn = wave % WN;
    const int fr = lane & 31, fh = lane >> 5;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int a_off = (wm * TM * 32 + fr) * LDS_LD + fh * 4;
    const int b_off = (wn * TN * 32 + fr) * LDS_LD + fh * 4;
    float pscl[TN], psft[TN];                          // this lane's output channels' scale / shift: in flight under the K loop
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * TN * 32 + j * 32 + fr;
        const int nc = n < p.Cout ? n : 0;
        pscl[j] = (!SPLIT && p.scale) ? p.scale[nc] : 1.0f;
        psft[j] = (!SPLIT && p.shift) ? p.shift[nc] : 0.0f;
    }

    auto compute = [&](int buf) {
        const float *a = As + buf * BM * LDS_LD + a_off;
        const float *b = Bs + buf * BN * LDS_LD + b_off;
#pragma unroll
        for (int kc = 0; kc < BK / 8; ++kc) {
            f32x4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4 *>(a + i * 32 * LDS_LD + kc * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4 *>(b + j * 32 * LDS_LD + kc * 8);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][t], bf[j][t], acc[i][j], 0, 0, 0);
        }
    };

    // K loop.  LDS holds two slabs (the one being multiplied, the next one), registers PF more: the loads of slabs kt+2 ..
    // kt+1+PF are in flight while slab kt is multiplied, so a workgroup alone on its CU (small grids: batch-1 layers, the
    // MBConv 1x1 convs of one batch lane) has two MFMA phases to cover a load round trip instead of one -- with one slab
    // in flight those loops ran at the L2 latency (0.67 us per slab against 0.43 us of MFMA work).
    // Prefetches past the last slab are unconditional: the taps are masked off (A) and the weight rows run into the next row
    // or the range check (B) -- harmless, never consumed.
    load_slab(kt0, areg[0], breg[0]);
    store_slab(0, areg[0], breg[0]);
#pragma unroll
    for (int d = 0; d < PF; ++d) load_slab(kt0 + 1 + d, areg[d], breg[d]);
    __syncthreads();
    int buf = 0;
    for (int kt = kt0; kt < nk; kt += PF) {
#pragma unroll
        for (int d = 0; d < PF; ++d) {               // register set d holds slab kt + d + 1
            if (kt + d >= nk) break;
            compute(buf);
            store_slab(buf ^ 1, areg[d], breg[d]);
            load_slab(kt + d + 1 + PF, areg[d], breg[d]);
            __syncthreads();
            buf ^= 1;
        }
    }

    if (SPLIT) {     // raw partial tile, [vec4][thread] so lanes store contiguously; summed by conv_fixup_kernel
        constexpr int NV4 = TM * TN * 4;
        f32x4 *dst = reinterpret_cast<f32x4 *>(p.ws) + (int64_t)blockIdx.x * NV4 * NT + tid;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int v = 0; v < 4; ++v)
                    dst[((i * TN + j) * 4 + v) * NT] =
                        f32x4{acc[i][j][4 * v], acc[i][j][4 * v + 1], acc[i][j][4 * v + 2], acc[i][j][4 * v + 3]};
        return;
    }
    const int m_base = m0 + wm * TM * 32, n_base = n0 + wn * TN * 32;
    if ((m0 + BM <= p.M) && (n0 + BN <= p.Cout))
        epilogue<ACT, RES, true, TM, TN>(p, acc, m_base, n_base, fr, fh, pscl, psft);
    else
        epilogue<ACT, RES, false, TM, TN>(p, acc, m_base, n_base, fr, fh, pscl, psft);
}

// ------------------------------------------------------------------------------------------------------------------------
// The same implicit GEMM on the bfloat16 matrix instructions, float32-exact operands ("split-bf16").
// Every float32 operand is cut into three bfloat16 pieces by round-to-nearest remainders,
//     a = a0 + a1 + a2,   a0 = bf16(a), a1 = bf16(a - a0), a2 = bf16(a - a0 - a1)      (both remainders are exact in float32;
//                                                                                       |a - a0 - a1 - a2| <= 2^-27 |a|)
// and a product is formed from the six piece products whose weight is 2^-18 or more,
//     a b ~= a0 b0 + (a0 b1 + a1 b0) + (a0 b2 + a1 b1 + a2 b0)        (dropped: a1 b2, a2 b1 <= 2^-27 |a b| each, a2 b2 <= 2^-36)
// -- each piece product is exact in float32 (8 x 8 significant bits) and v_mfma_f32_32x32x16_bf16 accumulates in float32, so
// the result carries the multiplicands' full 24 bits: the error of a product (2^-26) is BELOW the rounding of a float32
// multiply-add (2^-24), and the sums are float32 sums as in conv_igemm_kernel.  Six bf16 MFMAs of 16 k each take 192 cycles
// against 512 for the sixteen v_mfma_f32_32x32x2_f32 they replace: 2.67 x the matrix rate of the float32 instruction.
// (tests hold it to the same 2e-5 * max|y| against float64 as the float32 kernel; measured it is as close or closer.)
//
// Workgroup = 4 waves, tile BM x BN = 128 x (128 | 64), wave tile 64 x (64 | 32), K in slabs of 16 (one MFMA k-step).
// Activations are split while they are staged (global float32 -> registers -> three bf16 planes in LDS: ~7 VALU per
// element, once per element and tile), the weights arrive pre-split from `wsplit` (mydet_split_bf16_f32, once per layer).
// LDS rows of a plane: the 32 bytes of a row's 16 k-values, unpadded, the two 16-byte halves swapped where bit 3 of the row is set
// (fragment reads AND staging writes conflict-free); two slab buffers = 48 KB at 128 x 128: two to three workgroups per CU.
// Cin % 16 == 0 (a slab never straddles taps; 1x1 layers: Cin % 4 == 0, the last slab zero-filled).
// Non-finite inputs: bf16(inf) = inf and inf - inf = NaN, so an infinite activation or weight becomes NaN (the float32 kernel and
// the reference propagate inf), and a finite |x| > 2^127 * (2 - 2^-8) ~ 3.3962e38 (the midpoint above the largest bfloat16) rounds up to a bfloat16 inf -> NaN as well;
// NaN stays NaN.  A deliberate deviation (pinned by test_split_bf16_non_finite_semantics): the kernels are for
// finite tensors, and guarding the split costs two vector-ALU instructions per element in the loop that bounds these kernels.
constexpr int B3_COUT_PAD = 256;            // rows of the weight operand are padded to this (split_bf16_kernel)
__device__ __host__ __forceinline__ int b3_unit(int rr, int h) { return 2 * rr + (h ^ ((rr >> 2) & 1)); }
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split3(const f32x4 v, bf16x4 &p0, bf16x4 &p1, bf16x4 &p2) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const __bf16 h0 = (__bf16)v[e];
        const float r1 = v[e] - (float)h0;
        const __bf16 h1 = (__bf16)r1;
        const float r2 = r1 - (float)h1;
        p0[e] = h0; p1[e] = h1; p2[e] = (__bf16)r2;
    }
}

#ifndef B3_PF
#define B3_PF 2          // slabs prefetched into registers beyond the one staged (2, 3, 4, 6 measured equal: profiles/HISTORY.md)
#endif
// GATE (1x1 layers): the A operand is x * gate[image][channel] (squeeze-excite, as conv_igemm_kernel's GATE), applied before the split
template <int BM, int BN, int ACT, bool RES, bool SPLIT = false, int PF = B3_PF, int WN = 2, bool GATE = false>
__global__ __launch_bounds__(128 * WN, 2) void conv_igemm_b3_kernel(const ConvArgs p) {
    constexpr int NT = 128 * WN, WM = 2, BK = 16;
    // LDS rows of one plane are the 32 bytes of a row's 16 k-values, unpadded; the two 16-byte halves of row r are swapped when
    // bit 3 of r is set.  Fragment reads (lane = row, 16 bytes): 16 consecutive lanes cover all sixteen 16-byte slots of the 256-byte
    // bank space; staging writes (four 8-byte or two 16-byte lanes per row): 8 consecutive rows are 256 contiguous bytes.
    // (48-byte padded rows made the reads conflict-free and left the WRITES colliding: SQ_LDS_BANK_CONFLICT was a third of
    // SQ_LDS_IDX_ACTIVE -- profiles/r05_pmc_sq_b3.txt; 24.6 instead of 36.9 KB per buffer also lets a third workgroup onto the CU.)
    constexpr int ROWB = 32;
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int AI = BM * 4 / NT;                  // 16-byte float4 chunks of A per thread and slab (4 per row)
    constexpr int BCH = BN * 2;                      // 16-byte bf16x8 chunks of one B plane per slab (2 per row)
    constexpr int PLANE_A = BM * ROWB, PLANE_B = BN * ROWB, BUF = 3 * (PLANE_A + PLANE_B);
    extern __shared__ __attribute__((aligned(16))) char smem_b3[];

    const int tid = threadIdx.x;
    const int lid = SPLIT ? p.tile0 + (int)blockIdx.x / p.splits : mydet_xcd_remap(blockIdx.x, p.nblk);
    const int m0 = (lid / p.ntiles) * BM;
    const int n0 = (lid % p.ntiles) * BN;
    const int hwo = p.Ho * p.Wo;
    const int b0 = m0 / hwo;
    const int64_t img = (int64_t)p.H * p.W * p.ldx;
    const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x + b0 * img, (p.B - b0) * img * 4);
    const int CoutP = (p.Cout + B3_COUT_PAD - 1) / B3_COUT_PAD * B3_COUT_PAD;
    const __amdgpu_buffer_rsrc_t wr = make_rsrc(reinterpret_cast<const float *>(p.wsplit), (int64_t)((p.K + 15) >> 4) * 3 * CoutP * 32);

    // ---- staging roles.  A: chunk (4 floats) sc of rows sr + RP i.  B: chunk (8 bf16) bh of row br, all three planes.
    constexpr int RP = NT / 4;
    const int sc = tid & 3, sr = tid >> 2;
    const int ntaps = p.KH * p.KW;
    int aoff[AI];
    unsigned amask[AI];
    unsigned goff[AI];                               // GATE: byte offset of the row's image (+ this thread's channel quad) in gate[B][Cin]
    const __amdgpu_buffer_rsrc_t gr = make_rsrc(GATE ? p.gate : p.x, (int64_t)p.B * p.Cin * 4);
    const bool flat = ntaps == 1 && p.stride == 1 && p.pad_t == 0 && p.pad_l == 0 && p.Ho == p.H && p.Wo == p.W;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const int m = m0 + sr + RP * i;
        const int mm = m < p.M ? m : p.M - 1;
        goff[i] = GATE ? (unsigned)((mm / hwo) * p.Cin + sc * 4) * 4u : 0u;
        if (flat) {
            aoff[i] = (int)(((int64_t)(mm - b0 * hwo) * p.ldx + sc * 4) * 4);
            amask[i] = m < p.M ? 1u : 0u;
            continue;
        }
        const int ow = mm % p.Wo, t = mm / p.Wo;
        const int oh = t % p.Ho, b = t / p.Ho;
        const int ih0 = oh * p.stride - p.pad_t, iw0 = ow * p.stride - p.pad_l;
        aoff[i] = (int)((((int64_t)(b - b0) * p.H + ih0) * p.W + iw0) * p.ldx + sc * 4) * 4;
        unsigned mask = 0;
        for (int tp = 0; tp < ntaps; ++tp) {
            const int kh = tp / p.KW, kw = tp - kh * p.KW;
            if ((unsigned)(ih0 + kh) < (unsigned)p.H && (unsigned)(iw0 + kw) < (unsigned)p.W) mask |= 1u << tp;
        }
        amask[i] = m < p.M ? mask : 0u;
    }
    // B: 16-byte unit `tid` of the tile's plane-slab (BN rows x 32 bytes, contiguous in the slab-major layout); the unit's
    // (row, k-half) follow from the layout's swizzle (split_bf16_kernel); rows past Cout are zeros there
    const bool bact = tid < BCH;
    const int bu = tid & 63, brr = bu >> 1;
    const int br = (tid >> 6) * 32 + brr, bh = (bu & 1) ^ ((brr >> 2) & 1);
    const unsigned boff = bact ? (unsigned)(n0 * 32 + tid * 16) : OOB;
    const unsigned plane_bytes = (unsigned)CoutP * 32u, slab_bytes = 3u * plane_bytes;

    f32x4 areg[PF][AI];
    u32x4 breg[PF][3];
    const int nk_all = (p.K + BK - 1) / BK;          // (1x1 layers may have Cin % 16 == 4, 8, 12: the last slab's missing channels are zeros)
    int kt0 = 0, nk = nk_all;
    if (SPLIT) {
        const int sp = (int)blockIdx.x % p.splits;
        kt0 = (int)((unsigned)(nk_all * sp) / (unsigned)p.splits);
        nk = (int)((unsigned)(nk_all * (sp + 1)) / (unsigned)p.splits);
    }
    int tap = 0, c0 = 0, tapoff = 0;                 // uniform tap / channel base / byte offset of the slab being requested
    if (SPLIT) {
        const int k0 = kt0 * BK;
        tap = k0 / p.Cin;
        c0 = k0 - tap * p.Cin;
        const int kh = tap / p.KW, kw = tap - kh * p.KW;
        tapoff = (int)(((int64_t)kh * p.W + kw) * p.ldx) * 4;
    }
    auto load_slab = [&](int kt, f32x4 (&ar)[AI], u32x4 (&brg)[3]) {
        const unsigned uoff = (unsigned)(tapoff + c0 * 4);
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            bool ok = ((amask[i] >> tap) & 1u) && c0 + sc * 4 < p.Cin;
            ar[i] = buf_load16(xr, ok ? (unsigned)aoff[i] + uoff : OOB);
            if (GATE) {                               // (1x1: one tap, c0 is the slab's first channel)
                const f32x4 gv = buf_load16(gr, ok ? goff[i] + (unsigned)c0 * 4u : OOB);
#pragma unroll
                for (int e = 0; e < 4; ++e) ar[i][e] *= gv[e];
            }
        }
        c0 += BK;
        if (c0 == p.Cin) {                            // uniform: next tap
            c0 = 0;
            ++tap;
            const int kh = tap / p.KW, kw = tap - kh * p.KW;
            tapoff = (int)(((int64_t)kh * p.W + kw) * p.ldx) * 4;
        }
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
            brg[pl] = __builtin_amdgcn_raw_buffer_load_b128(wr, kt < nk_all ? boff : OOB,      // (the range check sees the voffset only: the
                                                            // look-ahead past the last slab must not leave the planes: ADVICE r05)
                                                            __builtin_amdgcn_readfirstlane((unsigned)kt * slab_bytes + (unsigned)pl * plane_bytes), 0);
    };
    auto store_slab = [&](int buf, const f32x4 (&ar)[AI], const u32x4 (&brg)[3]) {
        char *a = smem_b3 + buf * BUF, *b = a + 3 * PLANE_A;
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            bf16x4 q0, q1, q2;
            split3(ar[i], q0, q1, q2);
            char *d = a + (sr + RP * i) * ROWB + (((sc >> 1) ^ ((sr >> 3) & 1)) * 16) + (sc & 1) * 8;      // (RP % 16 == 0)
            *reinterpret_cast<bf16x4 *>(d) = q0;
            *reinterpret_cast<bf16x4 *>(d + PLANE_A) = q1;
            *reinterpret_cast<bf16x4 *>(d + 2 * PLANE_A) = q2;
        }
        if (bact) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x4 *>(b + pl * PLANE_B + br * ROWB + ((bh ^ ((br >> 3) & 1)) * 16)) = brg[pl];
        }
    };

    // ---- compute role
    const int wave = tid >> 6, lane = tid & 63;
    const int wm = wave / WN, wn = wave % WN;
    const int fr = lane & 31, fh = lane >> 5;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int fsw = (fh ^ ((fr >> 3) & 1)) * 16;           // (block rows start at multiples of 32: bit 3 of the row is bit 3 of fr)
    const int a_off = (wm * TM * 32 + fr) * ROWB + fsw;
    const int b_off = 3 * PLANE_A + (wn * TN * 32 + fr) * ROWB + fsw;
    float pscl[TN], psft[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * TN * 32 + j * 32 + fr;
        const int nc = n < p.Cout ? n : 0;
        pscl[j] = (!SPLIT && p.scale) ? p.scale[nc] : 1.0f;
        psft[j] = (!SPLIT && p.shift) ? p.shift[nc] : 0.0f;
    }
    auto compute = [&](int buf) {
        const char *a = smem_b3 + buf * BUF + a_off;
        const char *b = smem_b3 + buf * BUF + b_off;
        bf16x8 af[TM][3], bf[TN][3];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) af[i][pl] = *reinterpret_cast<const bf16x8 *>(a + pl * PLANE_A + i * 32 * ROWB);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) bf[j][pl] = *reinterpret_cast<const bf16x8 *>(b + pl * PLANE_B + j * 32 * ROWB);
        // the six piece products, small ones first (the running sum absorbs them at its own rounding either way); a piece pair
        // sweeps all blocks of the wave tile before the next pair, so MFMAs on one accumulator are TM * TN instructions apart
        constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][PA[t]], bf[j][PB[t]], acc[i][j], 0, 0, 0);
    };

    load_slab(kt0, areg[0], breg[0]);
    store_slab(0, areg[0], breg[0]);
#pragma unroll
    for (int d = 0; d < PF; ++d) load_slab(kt0 + 1 + d, areg[d], breg[d]);
    __syncthreads();
    int buf = 0;
    for (int kt = kt0; kt < nk; kt += PF) {
#pragma unroll
        for (int d = 0; d < PF; ++d) {               // register set d holds slab kt + d + 1
            if (kt + d >= nk) break;
            compute(buf);
            store_slab(buf ^ 1, areg[d], breg[d]);
            load_slab(kt + d + 1 + PF, areg[d], breg[d]);
            __syncthreads();
            buf ^= 1;
        }
    }

    if (SPLIT) {     // raw partial tile in conv_fixup_kernel's layout
        constexpr int NV4 = TM * TN * 4;
        f32x4 *dst = reinterpret_cast<f32x4 *>(p.ws) + (int64_t)blockIdx.x * NV4 * NT + tid;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int v = 0; v < 4; ++v)
                    dst[((i * TN + j) * 4 + v) * NT] =
                        f32x4{acc[i][j][4 * v], acc[i][j][4 * v + 1], acc[i][j][4 * v + 2], acc[i][j][4 * v + 3]};
        return;
    }
    const int m_base = m0 + wm * TM * 32, n_base = n0 + wn * TN * 32;
    if ((m0 + BM <= p.M) && (n0 + BN <= p.Cout))
        epilogue<ACT, RES, true, TM, TN>(p, acc, m_base, n_base, fr, fh, pscl, psft);
    else
        epilogue<ACT, RES, false, TM, TN>(p, acc, m_base, n_base, fr, fh, pscl, psft);
}

// ---- wide form: tile 128 x 256 on 8 waves (wave tile 64 x 64), ONE workgroup per CU.
// conv_igemm_b3_kernel above moves 20 KB per 128 x 128 x 16 multiplies: at the bf16 matrix rate that is the 27 B/cycle/CU an
// L2-resident stream sustains at best, and splitting 128 x 16 activations for 128 output channels keeps the vector ALU as busy
// as the matrix pipe (measured: 0.32 of the bf16 peak, 0.53 ms of a 0.73 ms launch is staging alone).  Twice the output channels per
// tile halve the activation bytes AND the split arithmetic per multiply; the weights -- already split, slab-major and swizzled
// -- come by LDS-DMA (raw_ptr_buffer_load_lds: no registers, no VALU, 1 KB contiguous per wave instruction) into a ring of three
// slab slots, two slabs ahead, so their landing is never waited for; activations are staged as above (one 16-byte chunk per
// thread and slab).  LDS: 2 x 18 KB of activation planes + 3 x 24 KB of weight slabs = 108 KB.
template <int ACT, bool RES, bool SPLIT = false>
__global__ __launch_bounds__(512, 1) void conv_igemm_b3w_kernel(const ConvArgs p) {
    constexpr int BM = 128, BN = 256, NT = 512, WN = 4, BK = 16, PF = 2, RING = 3;
    constexpr int ROWB = 48;                         // bytes per LDS row of one activation plane
    constexpr int TM = 2, TN = 2;
    constexpr int PLANE_A = BM * ROWB, ABUF = 3 * PLANE_A;
    constexpr int PLANE_B = BN * 32, BSLOT = 3 * PLANE_B;      // weight planes are packed 32-byte rows in DMA (swizzled) order
    extern __shared__ __attribute__((aligned(16))) char smem_b3[];
    char *As = smem_b3, *Bs = smem_b3 + 2 * ABUF;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int lid = SPLIT ? p.tile0 + (int)blockIdx.x / p.splits : mydet_xcd_remap(blockIdx.x, p.nblk);
    const int m0 = (lid / p.ntiles) * BM;
    const int n0 = (lid % p.ntiles) * BN;
    const int hwo = p.Ho * p.Wo;
    const int b0 = m0 / hwo;
    const int64_t img = (int64_t)p.H * p.W * p.ldx;
    const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x + b0 * img, (p.B - b0) * img * 4);
    const int CoutP = (p.Cout + B3_COUT_PAD - 1) / B3_COUT_PAD * B3_COUT_PAD;
    const __amdgpu_buffer_rsrc_t wr = make_rsrc(reinterpret_cast<const float *>(p.wsplit), (int64_t)(p.K >> 4) * 3 * CoutP * 32);
    const unsigned plane_bytes = (unsigned)CoutP * 32u, slab_bytes = 3u * plane_bytes;

    // ---- activation staging role: chunk (4 floats) sc of row sr
    const int sc = tid & 3, sr = tid >> 2;
    const int ntaps = p.KH * p.KW;
    int aoff;
    unsigned amask;
    {
        const bool flat = ntaps == 1 && p.stride == 1 && p.pad_t == 0 && p.pad_l == 0 && p.Ho == p.H && p.Wo == p.W;
        const int m = m0 + sr;
        const int mm = m < p.M ? m : p.M - 1;
        if (flat) {
            aoff = (int)(((int64_t)(mm - b0 * hwo) * p.ldx + sc * 4) * 4);
            amask = m < p.M ? 1u : 0u;
        } else {
            const int ow = mm % p.Wo, t = mm / p.Wo;
            const int oh = t % p.Ho, b = t / p.Ho;
            const int ih0 = oh * p.stride - p.pad_t, iw0 = ow * p.stride - p.pad_l;
            aoff = (int)((((int64_t)(b - b0) * p.H + ih0) * p.W + iw0) * p.ldx + sc * 4) * 4;
            unsigned mask = 0;
            for (int tp = 0; tp < ntaps; ++tp) {
                const int kh = tp / p.KW, kw = tp - kh * p.KW;
                if ((unsigned)(ih0 + kh) < (unsigned)p.H && (unsigned)(iw0 + kw) < (unsigned)p.W) mask |= 1u << tp;
            }
            amask = m < p.M ? mask : 0u;
        }
    }
    const int nk_all = p.K / BK;
    int kt0 = 0, nk = nk_all;
    if (SPLIT) {
        const int sp = (int)blockIdx.x % p.splits;
        kt0 = (int)((unsigned)(nk_all * sp) / (unsigned)p.splits);
        nk = (int)((unsigned)(nk_all * (sp + 1)) / (unsigned)p.splits);
    }
    int tap = 0, c0 = 0, tapoff = 0;                 // uniform tap / channel base / byte offset of the activation slab being requested
    if (SPLIT) {
        const int k0 = kt0 * BK;
        tap = k0 / p.Cin;
        c0 = k0 - tap * p.Cin;
        const int kh = tap / p.KW, kw = tap - kh * p.KW;
        tapoff = (int)(((int64_t)kh * p.W + kw) * p.ldx) * 4;
    }
    auto load_a = [&](f32x4 &ar) {                   // requests past the last slab run into masked taps: harmless, never consumed
        const bool ok = (amask >> tap) & 1u;
        ar = buf_load16(xr, ok ? (unsigned)aoff + (unsigned)(tapoff + c0 * 4) : OOB);
        c0 += BK;
        if (c0 == p.Cin) {                            // uniform: next tap
            c0 = 0;
            ++tap;
            const int kh = tap / p.KW, kw = tap - kh * p.KW;
            tapoff = (int)(((int64_t)kh * p.W + kw) * p.ldx) * 4;
        }
    };
    auto store_a = [&](int buf, const f32x4 &ar) {
        bf16x4 q0, q1, q2;
        split3(ar, q0, q1, q2);
        char *d = As + buf * ABUF + sr * ROWB + sc * 8;
        *reinterpret_cast<bf16x4 *>(d) = q0;
        *reinterpret_cast<bf16x4 *>(d + PLANE_A) = q1;
        *reinterpret_cast<bf16x4 *>(d + 2 * PLANE_A) = q2;
    };
    // weight slab kt -> ring slot: 24 pieces of 1 KB (3 planes x 8 row blocks of 32), three per wave; always three requests per
    // wave and call (a slab past the end reads out of range: zeros), so the counted waits below hold in every iteration
    const unsigned dma_lane = (unsigned)(n0 * 32 + lane * 16);
    auto dma_b = [&](int kt, int slot) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int idx = wave * 3 + j, pl = idx >> 3, pc = idx & 7;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                wr, (__attribute__((address_space(3))) void *)(Bs + slot * BSLOT + pl * PLANE_B + pc * 1024), 16,
                kt < nk_all ? dma_lane : OOB,
                __builtin_amdgcn_readfirstlane((unsigned)kt * slab_bytes + (unsigned)pl * plane_bytes + (unsigned)pc * 1024u), 0, 0);
        }
    };

    // ---- compute role
    const int wm = wave / WN, wn = wave % WN;
    const int fr = lane & 31, fh = lane >> 5;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int a_off = (wm * 64 + fr) * ROWB + fh * 16;
    const int b_off = (wn * 2) * 1024 + b3_unit(fr, fh) * 16;            // + j * 1024 + plane * PLANE_B
    float pscl[TN], psft[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * 64 + j * 32 + fr;
        const int nc = n < p.Cout ? n : 0;
        pscl[j] = (!SPLIT && p.scale) ? p.scale[nc] : 1.0f;
        psft[j] = (!SPLIT && p.shift) ? p.shift[nc] : 0.0f;
    }
    auto compute = [&](int buf, int slot) {
        const char *a = As + buf * ABUF + a_off;
        const char *b = Bs + slot * BSLOT + b_off;
        bf16x8 af[TM][3], bf[TN][3];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) af[i][pl] = *reinterpret_cast<const bf16x8 *>(a + pl * PLANE_A + i * 32 * ROWB);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) bf[j][pl] = *reinterpret_cast<const bf16x8 *>(b + pl * PLANE_B + j * 1024);
        constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][PA[t]], bf[j][PB[t]], acc[i][j], 0, 0, 0);
    };

    // prologue.  Requests, oldest first: B(kt0), B(kt0+1), A(kt0), A(kt0+1), A(kt0+2)
    f32x4 areg[PF];
    dma_b(kt0, 0);
    dma_b(kt0 + 1, 1);
    {
        f32x4 first;
        load_a(first);
        store_a(0, first);                            // (the wait for `first` also covers the two weight slabs requested before it)
    }
    load_a(areg[0]);
    load_a(areg[1]);
    __syncthreads();
    int buf = 0, slot = 0;
    for (int kt = kt0; kt < nk; kt += PF) {
#pragma unroll
        for (int d = 0; d < PF; ++d) {               // register set d holds activation slab kt + d + 1
            if (kt + d >= nk) break;
            const int s2 = slot >= 1 ? slot - 1 : RING - 1;             // (slot + 2) % 3: the slot read in the previous iteration
            dma_b(kt + d + 2, s2);
            compute(buf, slot);
            store_a(buf ^ 1, areg[d]);
            load_a(areg[d]);
            // in flight, oldest first: B(kt+d+1) | A(kt+d+2) | B(kt+d+2) x3 | A(kt+d+3).  Slab kt+d+1 of the weights is read
            // after the barrier: all but the five youngest requests must have landed
            __builtin_amdgcn_s_waitcnt(0x0F70 | 5);
            __syncthreads();
            buf ^= 1;
            slot = slot == RING - 1 ? 0 : slot + 1;
        }
    }

    if (SPLIT) {     // raw partial tile in conv_fixup_kernel's layout
        constexpr int NV4 = TM * TN * 4;
        f32x4 *dst = reinterpret_cast<f32x4 *>(p.ws) + (int64_t)blockIdx.x * NV4 * NT + tid;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int v = 0; v < 4; ++v)
                    dst[((i * TN + j) * 4 + v) * NT] =
                        f32x4{acc[i][j][4 * v], acc[i][j][4 * v + 1], acc[i][j][4 * v + 2], acc[i][j][4 * v + 3]};
        return;
    }
    const int m_base = m0 + wm * 64, n_base = n0 + wn * 64;
    if ((m0 + BM <= p.M) && (n0 + BN <= p.Cout))
        epilogue<ACT, RES, true, TM, TN>(p, acc, m_base, n_base, fr, fh, pscl, psft);
    else
        epilogue<ACT, RES, false, TM, TN>(p, acc, m_base, n_base, fr, fh, pscl, psft);
}

// float32 OHWI weight [Cout][K] -> the split-bf16 kernels' operand: three bfloat16 planes (w = p0 + p1 + p2), SLAB-MAJOR so that
// the rows a workgroup needs for one 16-k slab are one contiguous run, and swizzled for the LDS-DMA path:
//     out[((kt * 3 + pl) * CoutP + 32 * blk) * 16 + 8 * u + e],   kt = k / 16, CoutP = Cout rounded up to 256 (zero rows),
//     blk = n / 32, the 1 KB piece of rows 32 blk .. 32 blk + 31 holds 16-byte unit u = 2 (n % 32) + (h ^ ((n % 32 >> 2) & 1)) for
//     row n, k-half h (k % 16 = 8 h + e): copied to LDS as it lies (raw_ptr_buffer_load_lds writes a wave's 64 units in
//     order), a fragment read of eight consecutive rows then touches eight different 16-byte bank groups.
__global__ __launch_bounds__(256) void split_bf16_kernel(const float *w, int Cout, int K, int CoutP, unsigned short *out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;          // (kt, n, k % 16) over the padded rows
    const int64_t total = (int64_t)((K + 15) >> 4) * CoutP * 16;         // the last slab of a K that is no multiple of 16 is zero-filled
    if (i >= total) return;
    const int kk = (int)(i & 15);
    const int64_t t = i >> 4;
    const int n = (int)(t % CoutP), kt = (int)(t / CoutP);
    const float v = (n < Cout && kt * 16 + kk < K) ? w[(int64_t)n * K + kt * 16 + kk] : 0.f;
    const __bf16 h0 = (__bf16)v;
    const float r1 = v - (float)h0;
    const __bf16 h1 = (__bf16)r1;
    const __bf16 h2 = (__bf16)(r1 - (float)h1);
    const int64_t pos = ((int64_t)(n >> 5) * 64 + b3_unit(n & 31, kk >> 3)) * 8 + (kk & 7);      // inside one plane-slab
    const int64_t plane = (int64_t)CoutP * 16;
    unsigned short *o = out + (int64_t)kt * 3 * plane + pos;
    o[0] = __builtin_bit_cast(unsigned short, h0);
    o[plane] = __builtin_bit_cast(unsigned short, h1);
    o[2 * plane] = __builtin_bit_cast(unsigned short, h2);
}

// Split-K tail: sums the K-slice partials of one tile in slice order and applies the fused epilogue.
template <int BM, int BN, int WM, int WN, int ACT, bool RES>
__global__ __launch_bounds__(WM * WN * 64) void conv_fixup_kernel(const ConvArgs p) {
    constexpr int NT = WM * WN * 64;
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int NV4 = TM * TN * 4;
    const int tid = threadIdx.x;
    const int lid = p.tile0 + (int)blockIdx.x;
    const int m0 = (lid / p.ntiles) * BM, n0 = (lid % p.ntiles) * BN;
    const int wave = tid >> 6, lane = tid & 63;
    const int wm = wave / WN, wn = wave % WN;
    const int fr = lane & 31, fh = lane >> 5;
    f32x16 acc[TM][TN];
    const f32x4 *src = reinterpret_cast<const f32x4 *>(p.ws) + (int64_t)blockIdx.x * p.splits * NV4 * NT + tid;
    // slice by slice, in slice order, the next slice's NV4 16-byte loads in flight under the current slice's adds: a slice is
    // NV4 * 4 registers whatever the tile, and every element is summed o[0] + o[1] + ... as before.  (All slices of a fragment
    // requested at once -- 16 x 4 registers per fragment, unrolled over the fragments -- spilled: 1-3 KB of scratch per lane,
    // and the 64-tile x 8-slice fixup of the 256->512 stride-2 layer took 89 us for 32 MB.)
    f32x4 cur[NV4], nxt[NV4];
#pragma unroll
    for (int q = 0; q < NV4; ++q) cur[q] = src[(int64_t)q * NT];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int sp = 0; sp < p.splits; ++sp) {
        const bool more = sp + 1 < p.splits;          // uniform
        if (more) {
#pragma unroll
            for (int q = 0; q < NV4; ++q) nxt[q] = src[((int64_t)(sp + 1) * NV4 + q) * NT];
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int v = 0; v < 4; ++v)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float o = cur[(i * TN + j) * 4 + v][e];
                        acc[i][j][4 * v + e] = sp == 0 ? o : acc[i][j][4 * v + e] + o;
                    }
        if (more) {
#pragma unroll
            for (int q = 0; q < NV4; ++q) cur[q] = nxt[q];
        }
    }
    const int m_base = m0 + wm * TM * 32, n_base = n0 + wn * TN * 32;
    float pscl[TN], psft[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n_base + j * 32 + fr;
        const int nc = n < p.Cout ? n : 0;
        pscl[j] = p.scale ? p.scale[nc] : 1.0f;
        psft[j] = p.shift ? p.shift[nc] : 0.0f;
    }
    if ((m0 + BM <= p.M) && (n0 + BN <= p.Cout))
        epilogue<ACT, RES, true, TM, TN>(p, acc, m_base, n_base, fr, fh, pscl, psft);
    else
        epilogue<ACT, RES, false, TM, TN>(p, acc, m_base, n_base, fr, fh, pscl, psft);
}

template <int BM, int BN, int WM, int WN, int BK, bool CIN32, int ACT, bool RES, bool GATE = false, bool SPLIT = false, bool CAT = false>
int launch_inst(const ConvArgs &a, size_t lds, hipStream_t stream) {
    auto kern = &conv_igemm_kernel<BM, BN, WM, WN, BK, CIN32, ACT, RES, GATE, SPLIT, CAT>;
    static unsigned long long attr_set = 0;                  // > 64 KiB of dynamic LDS needs the opt-in once per device
    if (const int e = mydet_lds_opt_in(attr_set, kern, (int)lds)) return e;
    hipLaunchKernelGGL(kern, dim3(a.nblk), dim3(WM * WN * 64), lds, stream, a);
    return mydet_launch_status();
}

template <int BM, int BN, int WM, int WN, int BK, bool CIN32>
int launch_act(const ConvArgs &a, size_t lds, hipStream_t stream) {
    const bool res = a.res != nullptr;
    if (a.gate)         // only the MBConv project conv is gated: no activation (checked by the caller)
        return res ? launch_inst<BM, BN, WM, WN, BK, CIN32, MYDET_ACT_NONE, true, true>(a, lds, stream)
                   : launch_inst<BM, BN, WM, WN, BK, CIN32, MYDET_ACT_NONE, false, true>(a, lds, stream);
    switch (a.act) {
        case MYDET_ACT_LEAKY:
            return res ? launch_inst<BM, BN, WM, WN, BK, CIN32, MYDET_ACT_LEAKY, true>(a, lds, stream)
                       : launch_inst<BM, BN, WM, WN, BK, CIN32, MYDET_ACT_LEAKY, false>(a, lds, stream);
        case MYDET_ACT_SWISH:
            return res ? launch_inst<BM, BN, WM, WN, BK, CIN32, MYDET_ACT_SWISH, true>(a, lds, stream)
                       : launch_inst<BM, BN, WM, WN, BK, CIN32, MYDET_ACT_SWISH, false>(a, lds, stream);
        default:
            return res ? launch_inst<BM, BN, WM, WN, BK, CIN32, MYDET_ACT_NONE, true>(a, lds, stream)
                       : launch_inst<BM, BN, WM, WN, BK, CIN32, MYDET_ACT_NONE, false>(a, lds, stream);
    }
}

template <int BM, int BN, int WM, int WN, int ACT, bool RES>
int launch_fixup(const ConvArgs &a, int ntail, hipStream_t stream) {
    hipLaunchKernelGGL((conv_fixup_kernel<BM, BN, WM, WN, ACT, RES>), dim3(ntail), dim3(WM * WN * 64), 0, stream, a);
    return mydet_launch_status();
}

template <int BM, int BN, int WM, int WN>
int launch_fixup_act(const ConvArgs &a, int ntail, hipStream_t stream) {
    const bool res = a.res != nullptr;
    switch (a.act) {
        case MYDET_ACT_LEAKY:
            return res ? launch_fixup<BM, BN, WM, WN, MYDET_ACT_LEAKY, true>(a, ntail, stream)
                       : launch_fixup<BM, BN, WM, WN, MYDET_ACT_LEAKY, false>(a, ntail, stream);
        case MYDET_ACT_SWISH:
            return res ? launch_fixup<BM, BN, WM, WN, MYDET_ACT_SWISH, true>(a, ntail, stream)
                       : launch_fixup<BM, BN, WM, WN, MYDET_ACT_SWISH, false>(a, ntail, stream);
        default:
            return res ? launch_fixup<BM, BN, WM, WN, MYDET_ACT_NONE, true>(a, ntail, stream)
                       : launch_fixup<BM, BN, WM, WN, MYDET_ACT_NONE, false>(a, ntail, stream);
    }
}

// `slots` = workgroups of this configuration the chip holds at once (256 CUs x workgroups per CU).  Workgroups
// are equal-sized, so a grid of R full rounds plus a small remainder pays a whole extra round for the
// remainder.  When that happens (and the caller gave workspace) the remainder tiles are instead cut along K into
// `splits` slices that together fill the chip once, their partial tiles go to the workspace, and a small fixup
// launch sums them in slice order and applies the epilogue: deterministic, no atomics.
template <int BM, int BN, int WM, int WN, int BK>
int launch(const ConvArgs &a0, int slots, hipStream_t stream) {
    ConvArgs a = a0;
    const int mtiles = (a.M + BM - 1) / BM;
    a.ntiles = (a.Cout + BN - 1) / BN;
    const int total = mtiles * a.ntiles;
    const size_t lds = (size_t)2 * (BM + BN) * (BK + 4) * sizeof(float);
    const bool cin = (a.Cin % BK) == 0;
    const int nk = (a.K + BK - 1) / BK;
    const int rounds = total / slots;
    int rem = total % slots;
    // a grid that fills less than a quarter of one round (batch-1 / small-map layers: M = Ho*Wo is a few tiles) is cut along K
    // as a whole, so the chip is busy instead of a handful of CUs walking all of K
    // (a quarter, not half: between the two the K slices and the fixup launch cost more than the idle CUs -- round 4: batch 1
    // 643 -> 654 images/s, EfficientDet-D1 +1.3 %, D1-FCOS2-ATSS +1.0 %; an eighth is worse again)
    const bool small = rounds == 0 && total * 4 <= slots && nk >= 16;
    int splits = rem > 0 ? slots / rem : 0;
    if (splits > 16) splits = 16;
    if (splits > nk / 4) splits = nk / 4;
    const size_t need = (size_t)rem * (splits > 0 ? splits : 0) * BM * BN * sizeof(float);
    // only long-K layers: on short ones the two extra launches cost more than the spared round
    // (shorter K -- 8 or 16 slabs, the 256->128 / 512->256 1x1 layers of YOLOv3 -- measured 1-3 % SLOWER with the tail cut)
    // and at most four whole rounds: after more, the workgroups no longer finish together and the partial last round is cheap
    // already (128->256 stride 2 @160^2, 6.25 rounds: headline 1 454 -> 1 459 images/s without its tail; the same holds for F(4x4))
    const bool split = rem > 0 && splits >= 2 && a0.ws && need <= a0.ws_bytes &&
                       (small || (nk >= 32 && rounds >= 2 && rounds <= 4 && rem * 2 <= slots));
    a.tile0 = 0; a.splits = 1;
    a.nblk = split ? total - rem : total;
    int rc = 0;
    if (a.x1) {         // upsample + concat read on the fly: instantiated for the YOLOv3 pyramid's use (64 x 64 tile, LeakyReLU)
        if (!(BM == 64 && BN == 64 && BK == 32) || !cin || a.act != MYDET_ACT_LEAKY || a.res || a.gate) return MYDET_E_UNSUPP;
        if (a.nblk > 0) rc = launch_inst<64, 64, 2, 2, 32, true, MYDET_ACT_LEAKY, false, false, false, true>(a, lds, stream);
        if (rc || !split) return rc;
        a.tile0 = total - rem; a.splits = splits; a.nblk = rem * splits;
        rc = launch_inst<64, 64, 2, 2, 32, true, MYDET_ACT_NONE, false, false, true, true>(a, lds, stream);
        if (rc) return rc;
        return launch_fixup_act<BM, BN, WM, WN>(a, rem, stream);
    }
    if (a.nblk > 0)
        rc = cin ? launch_act<BM, BN, WM, WN, BK, true>(a, lds, stream) : launch_act<BM, BN, WM, WN, BK, false>(a, lds, stream);
    if (rc || !split) return rc;
    a.tile0 = total - rem; a.splits = splits; a.nblk = rem * splits;
    if (a.gate) {       // SE-gated project conv: the gate rides on the A operand, so the K slices sum like any others
        if (cin) rc = launch_inst<BM, BN, WM, WN, BK, true, MYDET_ACT_NONE, false, true, true>(a, lds, stream);
        else rc = launch_inst<BM, BN, WM, WN, BK, false, MYDET_ACT_NONE, false, true, true>(a, lds, stream);
    } else {
        if (cin) rc = launch_inst<BM, BN, WM, WN, BK, true, MYDET_ACT_NONE, false, false, true>(a, lds, stream);
        else rc = launch_inst<BM, BN, WM, WN, BK, false, MYDET_ACT_NONE, false, false, true>(a, lds, stream);
    }
    if (rc) return rc;
    return launch_fixup_act<BM, BN, WM, WN>(a, rem, stream);
}

// ---- split-bf16 launches (conv_igemm_b3_kernel): the same round / split-K-tail rule as `launch`, two workgroups per CU
template <int BM, int BN, int ACT, bool RES, bool SPLIT, int PF = B3_PF, int WN = 2, bool GATE = false>
int launch_b3_inst(const ConvArgs &a, hipStream_t stream) {
    auto kern = &conv_igemm_b3_kernel<BM, BN, ACT, RES, SPLIT, PF, WN, GATE>;
    constexpr int lds = 2 * 3 * (BM + BN) * 32;
    static unsigned long long attr_set = 0;
    if (const int e = mydet_lds_opt_in(attr_set, kern, lds)) return e;
    hipLaunchKernelGGL(kern, dim3(a.nblk), dim3(128 * WN), lds, stream, a);
    return mydet_launch_status();
}

template <int BM, int BN, int WN = 2>
int launch_b3(const ConvArgs &a0, hipStream_t stream) {
    ConvArgs a = a0;
    const int slots = 2 * mydet_cu_count();
    const int mtiles = (a.M + BM - 1) / BM;
    a.ntiles = (a.Cout + BN - 1) / BN;
    const int total = mtiles * a.ntiles;
    const int nk = (a.K + 15) / 16;
    const int rounds = total / slots;
    const int rem = total % slots;
    const bool small = rounds == 0 && total * 4 <= slots && nk >= 32;
    int splits = rem > 0 ? slots / rem : 0;
    if (splits > 16) splits = 16;
    if (splits > nk / 8) splits = nk / 8;
    const size_t need = (size_t)rem * (splits > 0 ? splits : 0) * BM * BN * sizeof(float);
    const bool split = rem > 0 && splits >= 2 && a0.ws && need <= a0.ws_bytes &&
                       (small || (nk >= 64 && rounds >= 2 && rounds <= 4 && rem * 2 <= slots));
    a.tile0 = 0; a.splits = 1;
    a.nblk = split ? total - rem : total;
    const bool res = a.res != nullptr;
    int rc = 0;
    if (a.gate) {           // squeeze-excite project convs: no activation (MYDET_E_UNSUPP otherwise: the caller's float32 kernel takes it)
        if (a.act != MYDET_ACT_NONE || WN != 2) return MYDET_E_UNSUPP;
        if (a.nblk > 0)
            rc = res ? launch_b3_inst<BM, BN, MYDET_ACT_NONE, true, false, B3_PF, 2, true>(a, stream)
                     : launch_b3_inst<BM, BN, MYDET_ACT_NONE, false, false, B3_PF, 2, true>(a, stream);
        if (rc || !split) return rc;
        a.tile0 = total - rem; a.splits = splits; a.nblk = rem * splits;
        rc = launch_b3_inst<BM, BN, MYDET_ACT_NONE, false, true, B3_PF, 2, true>(a, stream);
        if (rc) return rc;
        return launch_fixup_act<BM, BN, 2, WN>(a, rem, stream);
    }
    if (a.nblk > 0) {
        switch (a.act) {
            case MYDET_ACT_LEAKY: rc = res ? launch_b3_inst<BM, BN, MYDET_ACT_LEAKY, true, false, (WN == 4 ? 2 : B3_PF), WN>(a, stream) : launch_b3_inst<BM, BN, MYDET_ACT_LEAKY, false, false, (WN == 4 ? 2 : B3_PF), WN>(a, stream); break;
            case MYDET_ACT_SWISH: rc = res ? launch_b3_inst<BM, BN, MYDET_ACT_SWISH, true, false, (WN == 4 ? 2 : B3_PF), WN>(a, stream) : launch_b3_inst<BM, BN, MYDET_ACT_SWISH, false, false, (WN == 4 ? 2 : B3_PF), WN>(a, stream); break;
            default: rc = res ? launch_b3_inst<BM, BN, MYDET_ACT_NONE, true, false, (WN == 4 ? 2 : B3_PF), WN>(a, stream) : launch_b3_inst<BM, BN, MYDET_ACT_NONE, false, false, (WN == 4 ? 2 : B3_PF), WN>(a, stream); break;
        }
    }
    if (rc || !split) return rc;
    a.tile0 = total - rem; a.splits = splits; a.nblk = rem * splits;
    rc = launch_b3_inst<BM, BN, MYDET_ACT_NONE, false, true, (WN == 4 ? 2 : B3_PF), WN>(a, stream);
    if (rc) return rc;
    return launch_fixup_act<BM, BN, 2, WN>(a, rem, stream);
}

template <int ACT, bool RES, bool SPLIT>
int launch_b3w_inst(const ConvArgs &a, hipStream_t stream) {
    auto kern = &conv_igemm_b3w_kernel<ACT, RES, SPLIT>;
    constexpr int lds = 2 * 3 * 128 * 48 + 3 * 3 * 256 * 32;
    static unsigned long long attr_set = 0;
    if (const int e = mydet_lds_opt_in(attr_set, kern, lds)) return e;
    hipLaunchKernelGGL(kern, dim3(a.nblk), dim3(512), lds, stream, a);
    return mydet_launch_status();
}

// the wide form: one workgroup per CU, so a round is `cus` tiles; same split-K-tail rule otherwise
int launch_b3w(const ConvArgs &a0, hipStream_t stream) {
    constexpr int BM = 128, BN = 256;
    ConvArgs a = a0;
    const int slots = mydet_cu_count();
    const int mtiles = (a.M + BM - 1) / BM;
    a.ntiles = (a.Cout + BN - 1) / BN;
    const int total = mtiles * a.ntiles;
    const int nk = a.K / 16;
    const int rounds = total / slots;
    const int rem = total % slots;
    const bool small = rounds == 0 && total * 4 <= slots && nk >= 32;
    int splits = rem > 0 ? slots / rem : 0;
    if (splits > 16) splits = 16;
    if (splits > nk / 8) splits = nk / 8;
    const size_t need = (size_t)rem * (splits > 0 ? splits : 0) * BM * BN * sizeof(float);
    const bool split = rem > 0 && splits >= 2 && a0.ws && need <= a0.ws_bytes &&
                       (small || (nk >= 64 && rounds >= 2 && rounds <= 4 && rem * 2 <= slots));
    a.tile0 = 0; a.splits = 1;
    a.nblk = split ? total - rem : total;
    const bool res = a.res != nullptr;
    int rc = 0;
    if (a.nblk > 0) {
        switch (a.act) {
            case MYDET_ACT_LEAKY: rc = res ? launch_b3w_inst<MYDET_ACT_LEAKY, true, false>(a, stream) : launch_b3w_inst<MYDET_ACT_LEAKY, false, false>(a, stream); break;
            case MYDET_ACT_SWISH: rc = res ? launch_b3w_inst<MYDET_ACT_SWISH, true, false>(a, stream) : launch_b3w_inst<MYDET_ACT_SWISH, false, false>(a, stream); break;
            default: rc = res ? launch_b3w_inst<MYDET_ACT_NONE, true, false>(a, stream) : launch_b3w_inst<MYDET_ACT_NONE, false, false>(a, stream); break;
        }
    }
    if (rc || !split) return rc;
    a.tile0 = total - rem; a.splits = splits; a.nblk = rem * splits;
    rc = launch_b3w_inst<MYDET_ACT_NONE, false, true>(a, stream);
    if (rc) return rc;
    return launch_fixup_act<BM, BN, 2, 4>(a, rem, stream);
}

// Workgroups per CU assumed for tile configuration 6: the runtime reports 4 (mydet_conv_igemm_occupancy; the two-slab
// register prefetch of round 4 took the fifth); MYDET_CFG6_PER_CU overrides (A/B), read once.
int cfg6_per_cu() {
    static const int v = [] { const char *e = getenv("MYDET_CFG6_PER_CU"); return e && atoi(e) > 0 ? atoi(e) : 4; }();
    return v;
}

// Tile configurations (id -> BM x BN, wave grid, BK).  MYDET_CONV_CFG=<id> forces one (tuning only).
int launch_cfg(int id, const ConvArgs &a, hipStream_t s) {
    const int cus = mydet_cu_count();
    switch (id) {
        // last argument: resident workgroups = CUs (256 on MI355X) x (LDS / register limited workgroups per CU)
        case 0: return launch<128, 128, 2, 2, 32>(a, 2 * cus, s);
        case 1: return launch<128, 64, 2, 2, 32>(a, 2 * cus, s);
        case 2: return launch<128, 32, 4, 1, 32>(a, 3 * cus, s);
        case 3: return launch<64, 64, 2, 2, 32>(a, 4 * cus, s);
        case 6: return launch<128, 64, 2, 2, 16>(a, cfg6_per_cu() * cus, s);
        case 9: return launch<128, 96, 4, 1, 32>(a, 2 * cus, s);       // Cout in (64, 96]: 80 / 88 channels
        case 8: return launch<128, 128, 2, 4, 32>(a, 2 * cus, s);     // 8 waves, wave tile 64x32
        default: return MYDET_E_BADARG;
    }
}

int forced_cfg() {      // MYDET_CONV_CFG=<id>: tuning only; read per call so that one process can sweep the configurations
    const char *e = getenv("MYDET_CONV_CFG");
    return e && *e ? atoi(e) : -1;
}

bool pw_wide_on() {     // MYDET_PW_WIDE=1: every 1x1 conv with Cin <= 240 goes to pointwise.hip (tuning only)
    const char *e = getenv("MYDET_PW_WIDE");
    return e && *e == '1';
}

bool pw_skinny_on() {   // MYDET_PW_SKINNY=0: tuning only (A/B against the tiled kernel)
    const char *e = getenv("MYDET_PW_SKINNY");
    return !(e && *e == '0');
}

// Tile configuration of a layer (ids of launch_cfg) -- one rule for mydet_conv2d_igemm_f32 and for the fused
// upsample-concat entry point, which only exists for configuration 3 and must say so for every other shape.
int choose_cfg(int64_t M64, int Cin, int Cout, int taps) {
    // Tile choice, from the per-shape sweep in profiles/ (tools/sweep_conv_cfg.sh):
    //   narrow outputs take a narrow N tile; short-K layers (1x1 convs, prologue/epilogue-bound) and
    //   grids under ~4 blocks per CU do best with 64x64 tiles at 4 workgroups/CU; the long-K 3x3
    //   layers with big grids take 128x128 tiles on 8 waves (wave tile 64x32, 4 waves per SIMD).
    const int64_t blocks128 = ((M64 + 127) / 128) * ((Cout + 127) / 128);
    const int K = taps * Cin;
    // short generic-K pointwise convs (EfficientNet expand/project, BiFPN, heads): BK = 16 wastes no staging on
    // K = 16/24/40/88... and five 30 KB workgroups fit a CU
    if ((Cin % 32) != 0 && K <= 256) return 6;
    if (Cout <= 32) return 2;
    // 80 / 88 output channels behind a long K (480->80 project convs): a 96-wide tile instead of two 64-wide ones
    // (tools/sweep_pointwise.py, batch 32: 67 -> 57 us; a tie at batch 16 and a loss at batch 8 -- 12 800 rows:
    // 36 vs 24 us -- where the 64 x 64 tiles with the split-K tail fill the chip better)
    if (taps == 1 && Cout > 64 && Cout <= 96 && K >= 384 && M64 >= 40000) return 9;
    // short-K, very wide outputs (EfficientNet expand convs at 20^2: 192->1152, 320->1920): the 8-wave 128x128 tile
    // (tools/sweep_pointwise.py: 79 -> 73 us, 35.7 -> 32.9 us at batch 16; under 4 800 rows -- batch 8 at 20^2 -- the
    // 64 x 64 tiles are ahead again: 320->1920 47 -> 44 us)
    if (taps == 1 && K <= 512 && Cout >= 1024 && (M64 >= 4800 || K <= 256)) return 8;
    // (1x1 with 33..64 outputs behind a long K -- YOLOv3's 128->64 @160^2 -- also does better with BK = 16 and five workgroups
    // per CU: 0.172 -> 0.156 ms in the model, round 4)
    if (Cout <= 64) return taps > 1 || K >= 96 ? 6 : 1;
    // (3x3 layers with K >= 512 and a big grid already prefer the 8-wave tile: 64->128 stride 2 @320^2 +7 %)
    if ((K <= 1024 && !(taps > 1 && K >= 512)) || blocks128 < 1024) return 3;
    return 8;
}

}  // namespace

extern "C" int mydet_conv2d_igemm_f32(const float *x, int64_t ldx, const float *w, const float *scale,
                                      const float *shift, const float *residual, int64_t ldr, const float *a_gate,
                                      void *workspace, int64_t workspace_bytes, float *y,
                                      int64_t ldy, int B, int H, int W, int Cin, int Cout, int KH, int KW,
                                      int stride, int pad_t, int pad_l, int Ho, int Wo, int act, void *stream) {
    if (!x || !w || !y || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || KH <= 0 || KW <= 0 ||
        stride <= 0 || Ho <= 0 || Wo <= 0)
        return MYDET_E_BADARG;
    if ((Cin & 3) || (ldx & 3) || ldx < Cin || ldy < Cout || (residual && ldr < Cout)) return MYDET_E_BADARG;
    if (((uintptr_t)x & 15) || ((uintptr_t)w & 15) || ((uintptr_t)y & 15) || (residual && ((uintptr_t)residual & 15)) ||
        (scale && ((uintptr_t)scale & 15)) || (shift && ((uintptr_t)shift & 15)))
        return MYDET_E_BADARG;                       // the epilogue uses 16-byte buffer operations on all of these
    if ((ldy & 3) || (residual && (ldr & 3))) return MYDET_E_UNSUPP;
    const int64_t M64 = (int64_t)B * Ho * Wo;
    if (M64 > (int64_t)1 << 30 || (int64_t)KH * KW * Cin > (int64_t)1 << 30) return MYDET_E_BADARG;
    // 32-bit byte offsets inside a block's window (the images its 128 rows touch, +1) and the weights
    const int64_t img_bytes = (int64_t)H * W * ldx * 4;
    const int64_t span_imgs = 256 / ((int64_t)Ho * Wo) + 2;      // tiles are at most 256 rows
    if ((int64_t)M64 * ldy * 4 >= 0x7FFFFFF0ll || (residual && (int64_t)M64 * ldr * 4 >= 0x7FFFFFF0ll))
        return MYDET_E_UNSUPP;
    if (act < 0 || act > 2) return MYDET_E_BADARG;
    if (KH * KW > 31 || img_bytes * span_imgs >= 0x7FFFFFF0ll || (int64_t)Cout * KH * KW * Cin * 4 >= 0x7FFFFFF0ll)
        return MYDET_E_UNSUPP;
    ConvArgs a;
    if (a_gate && (KH != 1 || KW != 1 || stride != 1 || act != MYDET_ACT_NONE || ((uintptr_t)a_gate & 15)))
        return MYDET_E_UNSUPP;
    a.x = x; a.w = w; a.scale = scale; a.shift = shift; a.res = residual; a.gate = a_gate; a.y = y;
    a.ldx = ldx; a.ldr = ldr; a.ldy = ldy;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.KH = KH; a.KW = KW; a.stride = stride;
    a.pad_t = pad_t; a.pad_l = pad_l; a.Ho = Ho; a.Wo = Wo; a.act = act;
    a.M = (int)M64; a.K = KH * KW * Cin; a.ntiles = 0; a.nblk = 0;
    a.tile0 = 0; a.splits = 1;
    a.x1 = nullptr; a.ldx1 = 0; a.C1 = 0; a.wsplit = nullptr;
    a.ws = ((uintptr_t)workspace & 15) ? nullptr : (float *)workspace;
    a.ws_bytes = workspace_bytes > 0 ? (size_t)workspace_bytes : 0;
    hipStream_t s = (hipStream_t)stream;
    if (forced_cfg() >= 0) return launch_cfg(forced_cfg(), a, s);
    // few output channels behind a big map (MBConv project convs of the high-resolution stages): the LDS-free skinny
    // kernel of pointwise.hip (tools/sweep_pointwise.py)
    if (KH * KW == 1 && stride == 1 && Ho == H && Wo == W && pw_skinny_on() &&
        ((Cout <= 48 && M64 >= 65536) || pw_wide_on())) {
        const int rc = mydet_pw_skinny(x, ldx, w, scale, shift, residual, ldr, a_gate, y, ldy, B, H * W, Cin, Cout, act, stream);
        if (rc != MYDET_E_UNSUPP) return rc;
    }
    return launch_cfg(choose_cfg(M64, Cin, Cout, KH * KW), a, s);
}

extern "C" int64_t mydet_split_bf16_elems(int Cout, int K) {
    if (Cout <= 0 || K <= 0 || (K & 3)) return 0;
    return (int64_t)((K + 15) >> 4) * 3 * ((Cout + B3_COUT_PAD - 1) / B3_COUT_PAD * B3_COUT_PAD) * 16;
}

extern "C" int mydet_split_bf16_f32(const float *w, int Cout, int K, uint16_t *planes, void *stream) {
    if (!w || !planes || Cout <= 0 || K <= 0) return MYDET_E_BADARG;
    if (K & 3) return MYDET_E_UNSUPP;
    const int CoutP = (Cout + B3_COUT_PAD - 1) / B3_COUT_PAD * B3_COUT_PAD;
    const int64_t total = (int64_t)((K + 15) >> 4) * CoutP * 16;
    if ((total + 255) / 256 > 0x7fffffff) return MYDET_E_UNSUPP;
    hipLaunchKernelGGL(split_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, Cout, K,
                       CoutP, planes);
    return mydet_launch_status();
}

static int g_b3_form = -1;
static int b3_form() {
    if (g_b3_form < 0) {
        const char *w = getenv("MYDET_B3_WIDE"), *v = getenv("MYDET_B3_WAVES");
        g_b3_form = ((w && *w == '1') ? 1 : 0) | ((v && atoi(v) == 8) ? 2 : 0);
    }
    return g_b3_form;
}
/* Test hook: read MYDET_B3_WIDE / MYDET_B3_WAVES again (they are read once per process otherwise). */
extern "C" int mydet_conv_b3_reload_tuning(void) {
    g_b3_form = -1;
    return b3_form();
}

extern "C" int mydet_conv2d_igemm_b3_f32(const float *x, int64_t ldx, const uint16_t *w_planes, const float *scale,
                                         const float *shift, const float *residual, int64_t ldr, const float *a_gate, void *workspace,
                                         int64_t workspace_bytes, float *y, int64_t ldy, int B, int H, int W, int Cin, int Cout,
                                         int KH, int KW, int stride, int pad_t, int pad_l, int Ho, int Wo, int act, void *stream) {
    if (!x || !w_planes || !y || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || KH <= 0 || KW <= 0 || stride <= 0 ||
        Ho <= 0 || Wo <= 0)
        return MYDET_E_BADARG;
    if ((ldx & 3) || ldx < Cin || ldy < Cout || (residual && ldr < Cout)) return MYDET_E_BADARG;
    if (((uintptr_t)x & 15) || ((uintptr_t)w_planes & 15) || ((uintptr_t)y & 15) || (residual && ((uintptr_t)residual & 15)) ||
        (scale && ((uintptr_t)scale & 15)) || (shift && ((uintptr_t)shift & 15)))
        return MYDET_E_BADARG;
    if (act < 0 || act > 2) return MYDET_E_BADARG;
    if (a_gate && (KH != 1 || KW != 1 || ((uintptr_t)a_gate & 15))) return MYDET_E_BADARG;
    // a 16-channel slab never straddles taps: Cin % 16 == 0, or ONE tap (1x1) with Cin % 4 == 0 (the last slab's missing channels: zeros)
    if (((Cin & 15) && !(KH == 1 && KW == 1 && !(Cin & 3))) || (ldy & 3) || (residual && (ldr & 3))) return MYDET_E_UNSUPP;
    if ((Cin & 15) && (Cout > 192 ? b3_form() & 1 : 0)) return MYDET_E_UNSUPP;          // (the opt-in wide form has no padded-K instance)
    const int64_t M64 = (int64_t)B * Ho * Wo, K64 = (int64_t)KH * KW * Cin;
    if (M64 > (int64_t)1 << 30 || K64 > (int64_t)1 << 30) return MYDET_E_BADARG;
    const int64_t img_bytes = (int64_t)H * W * ldx * 4;
    const int64_t span_imgs = 256 / ((int64_t)Ho * Wo) + 2;
    if (M64 * ldy * 4 >= 0x7FFFFFF0ll || (residual && M64 * ldr * 4 >= 0x7FFFFFF0ll)) return MYDET_E_UNSUPP;
    if (KH * KW > 31 || img_bytes * span_imgs >= 0x7FFFFFF0ll || mydet_split_bf16_elems(Cout, (int)K64) * 2 >= 0x7FFFFFF0ll) return MYDET_E_UNSUPP;
    ConvArgs a;
    a.x = x; a.w = nullptr; a.scale = scale; a.shift = shift; a.res = residual; a.gate = a_gate; a.y = y;
    a.ldx = ldx; a.ldr = ldr; a.ldy = ldy;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.KH = KH; a.KW = KW; a.stride = stride;
    a.pad_t = pad_t; a.pad_l = pad_l; a.Ho = Ho; a.Wo = Wo; a.act = act;
    a.M = (int)M64; a.K = (int)K64; a.ntiles = 0; a.nblk = 0; a.tile0 = 0; a.splits = 1;
    a.x1 = nullptr; a.ldx1 = 0; a.C1 = 0; a.wsplit = w_planes;
    a.ws = ((uintptr_t)workspace & 15) ? nullptr : (float *)workspace;
    a.ws_bytes = workspace_bytes > 0 ? (size_t)workspace_bytes : 0;
    if (Cout <= 64) return launch_b3<128, 64>(a, (hipStream_t)stream);
    // two other forms stay selectable (read once; mydet_conv_b3_reload_tuning re-reads), both correct and tested, neither faster:
    //   MYDET_B3_WIDE=1  from 192 output channels a 128 x 256 tile, one 8-wave workgroup per CU, weights by LDS-DMA -- 0.66 / 0.75
    //                    / 0.66 ms vs 0.65 / 0.69 / 0.69 on the three deep stride-2 layers, 1 538 vs 1 545 images/s in the model;
    //   MYDET_B3_WAVES=8 8-wave workgroups (wave tile 64 x 32, four waves per SIMD) -- 5-10 % faster back to back, 0.7 % slower
    //                    in the model (1 510 vs 1 520 images/s, twice each in one call)
    // at most half as many 128-row tiles as CUs (the EfficientNet project convs on a lane of 8-16 images): 64-row tiles, twice the
    // workgroups.  MYDET_B3_HALF_TILES overrides the limit (0 = never; tuning)
    static const int half_tiles = [] { const char *e = getenv("MYDET_B3_HALF_TILES"); return e ? atoi(e) : mydet_cu_count() / 2; }();
    if (((M64 + 127) / 128) * ((Cout + 127) / 128) <= half_tiles) return launch_b3<64, 128>(a, (hipStream_t)stream);
    const int form = a_gate ? 0 : b3_form();            // (the opt-in forms have no gated instances)
    if ((form & 1) && Cout > 192) return launch_b3w(a, (hipStream_t)stream);
    if (form & 2) return launch_b3<128, 128, 4>(a, (hipStream_t)stream);
    return launch_b3<128, 128>(a, (hipStream_t)stream);
}

/* Test / tuning hook: workgroups per CU the runtime reports for the base instance of tile configuration `cfg`
 * (hipOccupancyMaxActiveBlocksPerMultiprocessor), next to the count launch_cfg assumes; < 0 = MYDET_E_*. */
extern "C" int mydet_conv_igemm_occupancy(int cfg, int *assumed) {
    int n = 0, as = 0;
    hipError_t e = hipErrorInvalidValue;
#define MYDET_OCC(BM, BN, WM, WN, BK, PER_CU)                                                                                       \
    {                                                                                                                                \
        auto kern = &conv_igemm_kernel<BM, BN, WM, WN, BK, true, MYDET_ACT_LEAKY, false, false, false, false>;                        \
        const size_t lds = (size_t)2 * (BM + BN) * (BK + 4) * sizeof(float);                                                          \
        static unsigned long long m = 0;                                                                                             \
        if (const int rc = mydet_lds_opt_in(m, kern, (int)lds)) return rc;                                                            \
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, WM * WN * 64, lds);                                                \
        as = PER_CU;                                                                                                                 \
    }
    switch (cfg) {
        case 0: MYDET_OCC(128, 128, 2, 2, 32, 2) break;
        case 1: MYDET_OCC(128, 64, 2, 2, 32, 2) break;
        case 2: MYDET_OCC(128, 32, 4, 1, 32, 3) break;
        case 3: MYDET_OCC(64, 64, 2, 2, 32, 4) break;
        case 6: MYDET_OCC(128, 64, 2, 2, 16, cfg6_per_cu()) break;
        case 9: MYDET_OCC(128, 96, 4, 1, 32, 2) break;
        case 8: MYDET_OCC(128, 128, 2, 4, 32, 2) break;
        default: return MYDET_E_BADARG;
    }
#undef MYDET_OCC
    if (assumed) *assumed = as;
    return e == hipSuccess ? n : MYDET_E_UNSUPP;
}

extern "C" int mydet_conv1x1_upcat_f32(const float *x_lo, int64_t ld_lo, int C_lo, const float *x_hi, int64_t ld_hi, int C_hi,
                                       const float *w, const float *scale, const float *shift, void *workspace,
                                       int64_t workspace_bytes, float *y, int64_t ldy, int B, int H, int W, int Cout, int act,
                                       void *stream) {
    if (!x_lo || !x_hi || !w || !y || B <= 0 || H <= 0 || W <= 0 || C_lo <= 0 || C_hi <= 0 || Cout <= 0) return MYDET_E_BADARG;
    if ((ld_lo & 3) || (ld_hi & 3) || ld_lo < C_lo || ld_hi < C_hi || ldy < Cout || (ldy & 3)) return MYDET_E_BADARG;
    if (((uintptr_t)x_lo & 15) || ((uintptr_t)x_hi & 15) || ((uintptr_t)w & 15) || ((uintptr_t)y & 15) ||
        (scale && ((uintptr_t)scale & 15)) || (shift && ((uintptr_t)shift & 15)))
        return MYDET_E_BADARG;
    if ((H & 1) || (W & 1) || (C_lo & 31) || (C_hi & 31) || act != MYDET_ACT_LEAKY) return MYDET_E_UNSUPP;
    const int64_t M64 = (int64_t)B * H * W;
    const int Cin = C_lo + C_hi;
    // only where the two-launch path would run the same 64 x 64 x 32 tile (same k order, same cuts: bit-identical results);
    // a shape the regular rule sends to another tile -- or to the skinny kernel -- is the caller's two launches
    if (forced_cfg() >= 0 ? forced_cfg() != 3 : (choose_cfg(M64, Cin, Cout, 1) != 3 || (Cout <= 48 && M64 >= 65536 && pw_skinny_on())))
        return MYDET_E_UNSUPP;
    if (M64 > (int64_t)1 << 30 || M64 * ldy * 4 >= 0x7FFFFFF0ll) return MYDET_E_UNSUPP;
    const int64_t span_imgs = 256 / ((int64_t)H * W) + 2;
    if ((int64_t)H * W * ld_hi * 4 * span_imgs >= 0x7FFFFFF0ll || (int64_t)(H >> 1) * (W >> 1) * ld_lo * 4 * span_imgs >= 0x7FFFFFF0ll ||
        (int64_t)Cout * Cin * 4 >= 0x7FFFFFF0ll)
        return MYDET_E_UNSUPP;
    ConvArgs a;
    a.x = x_hi; a.w = w; a.scale = scale; a.shift = shift; a.res = nullptr; a.gate = nullptr; a.y = y;
    a.ldx = ld_hi; a.ldr = 0; a.ldy = ldy;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.KH = 1; a.KW = 1; a.stride = 1;
    a.pad_t = 0; a.pad_l = 0; a.Ho = H; a.Wo = W; a.act = act;
    a.M = (int)M64; a.K = Cin; a.ntiles = 0; a.nblk = 0;
    a.tile0 = 0; a.splits = 1;
    a.x1 = x_lo; a.ldx1 = ld_lo; a.C1 = C_lo; a.wsplit = nullptr;
    a.ws = ((uintptr_t)workspace & 15) ? nullptr : (float *)workspace;
    a.ws_bytes = workspace_bytes > 0 ? (size_t)workspace_bytes : 0;
    return launch_cfg(3, a, (hipStream_t)stream);          // the tile the 1x1 layers of this shape take anyway (same k order)
}
