// Fused MBConv front half for the shallow EfficientNet blocks (Cin <= 40):
//
//     y = swish(BN1( depthwise_kxk_stride_s( swish(BN0( expand1x1(x) )) ) ))      + per-tile channel sums of y (SE squeeze)
// (the caller folds each BatchNorm's scale into the conv weights; the shifts initialise the accumulators, so the
// normalisation costs no instruction)
//
// external/efficientnet/model.py:71-79 (MBConvBlock.forward: _expand_conv -> _bn0 -> swish -> _depthwise_conv -> _bn1 ->
// swish) with the static "SAME" padding of Conv2dStaticSamePadding (utils.py:122-145) applied to the EXPANDED tensor.
// Unfused, the 6x-wide expanded tensor is written by the expand conv and read back by the depthwise conv: at 160^2 /
// 80^2 that round trip is most of the block's HBM traffic (for a stride-2 block the expanded map is 4x the output).
// Here it only ever exists as a tile in LDS.
//
// One 256-thread workgroup = one output tile (8x16 at stride 1, 4x8 at stride 2) of one image; it loads the input
// patch once and then walks the expanded channels 32 at a time:
//   1. the input patch (tile + halo, all Cin channels) -> LDS rows of Cin+1 floats (odd stride: conflict-free
//      ds_read_b32 fragments); a per-pixel validity flag marks pixels outside the image
//   2. expand on FP32 MFMA (v_mfma_f32_32x32x2_f32): [patch pixels] x [Cin] x [32 channels]; weights straight from
//      L2 into fragment registers; BN0 + swish on the accumulators; pixels outside the image become the zeros the
//      depthwise conv pads with; result -> LDS [pixel][32]
//   3. depthwise k x k on the VALU from LDS (taps in (kh, kw) order, one fmaf chain per channel, as dwconv_kernel),
//      BN1 + swish, 128-B-per-pixel stores, and the tile's channel sums for the SE squeeze (fixed order, no atomics).
// The constants of the next 32 channels (weight fragments, BatchNorm terms, depthwise taps) are requested while the
// current 32 are in their depthwise phase.
// Built with -ffp-contract=off like the other EfficientNet kernels.
#include "common.h"
#include "se_tail.h"

namespace {

struct MbArgs {
    const float *x, *we, *shift0, *wd, *shift1;
    float *y, *partial;
    int64_t ldx, ldy;
    int H, W, Ho, Wo, Cin, Cexp, pad_t, pad_l, tiles_x, tiles_per_img, chunks, S;
    SeTail se;                // se.gate != NULL: the squeeze-excite gate is finished inside the launch (se_tail.h)
};

template <int K, int ST>
struct MbGeom {
    static constexpr int TH = ST == 1 ? 8 : 4, TW = ST == 1 ? 16 : 8;
    static constexpr int IH = (TH - 1) * ST + K, IW = (TW - 1) * ST + K;
    static constexpr int NPIX = IH * IW, NPIXP = (NPIX + 31) / 32 * 32, MB = NPIXP / 32;
};

template <int K, int ST, int CIN>
__global__ __launch_bounds__(256, 2) void mbconv_expand_dw_kernel(const MbArgs p) {
    using G = MbGeom<K, ST>;
    constexpr int TH = G::TH, TW = G::TW, IW = G::IW, NPIX = G::NPIX, NPIXP = G::NPIXP, MB = G::MB;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int Cin = CIN, XS = Cin + 1;
    float *sx = lds;                                  // [NPIXP][Cin + 1]   input patch
    float *se = lds + ((NPIXP * XS + 3) & ~3);        // [NPIXP][32]        expanded tile of the current channel chunk
    float *sval = se + NPIXP * 32;                    // [NPIXP]            1 inside the image, 0 outside
    float *swd = sval + NPIXP;                        // [K*K][32]          depthwise taps of the current chunk
    f32x4 *red = reinterpret_cast<f32x4 *>(swd + K * K * 32);   // [4][8]   SE partial sums per wave
    float *selds = reinterpret_cast<float *>(red + 32);         // [MYDET_SE_LDS_FLOATS] in-launch SE tail (se_tail.h)
    float *tots = selds + MYDET_SE_LDS_FLOATS;                  // [chunks * 32]  this tile's channel sums, all chunks
    const bool se_on = p.se.gate != nullptr;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t = blockIdx.x;
    const int b = t / p.tiles_per_img, r = t - b * p.tiles_per_img;
    const unsigned se_ep = se_on ? se_epoch(p.se, b) : 0u;
    const int ty = r / p.tiles_x, tx = r - ty * p.tiles_x;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int iy0 = oy0 * ST - p.pad_t, ix0 = ox0 * ST - p.pad_l;
    const int q = tid & 7;                             // channel quad inside a chunk (phase 3)
    // does the patch reach outside the image?  (uniform; interior tiles skip the padding mask)
    const bool edge = iy0 < 0 || ix0 < 0 || iy0 + G::IH > p.H || ix0 + IW > p.W;

    // per-chunk constants are requested one chunk ahead (registers), so their latency hides under the previous chunk
    float wf[Cin / 2];                                 // this lane's B fragments of the expand conv, one per k-step
    float sh0;
    f32x4 sh1, tapv;
    auto request = [&](int c0) {
        const int ch = c0 + (lane & 31);
        const bool chv = ch < p.Cexp;
        const float *wrow = p.we + (int64_t)(chv ? ch : 0) * Cin + (lane >> 5);
#pragma unroll
        for (int k = 0; k < Cin / 2; ++k) wf[k] = wrow[2 * k];
        sh0 = chv ? p.shift0[ch] : 0.f;
        const int cq = min(c0 + q * 4, p.Cexp - 4);
        sh1 = *reinterpret_cast<const f32x4 *>(p.shift1 + cq);
        const int tap = min(tid >> 3, K * K - 1);
        tapv = *reinterpret_cast<const f32x4 *>(p.wd + tap * p.Cexp + cq);
    };
    request(0);

    // 1. input patch -> LDS once per tile (all loads of a thread in flight; scalar LDS stores at the odd row stride)
    constexpr int QI = Cin >> 2;
    constexpr int NIT = (NPIXP * QI + 255) / 256;
    {
        const float *xb = p.x + (int64_t)b * p.H * p.W * p.ldx;
        f32x4 pv[NIT];
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int it = min(tid + u * 256, NPIXP * QI - 1);
            const int px = it / QI, q4 = it - px * QI;
            const int py = px / IW, pxx = px - py * IW;
            const int cy = min(max(iy0 + py, 0), p.H - 1), cx = min(max(ix0 + pxx, 0), p.W - 1);
            pv[u] = *reinterpret_cast<const f32x4 *>(xb + ((int64_t)cy * p.W + cx) * p.ldx + q4 * 4);    // unconditional, clamped
        }
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int it = tid + u * 256;
            if (it < NPIXP * QI) {
                const int px = it / QI, q4 = it - px * QI;
                const int py = px / IW, pxx = px - py * IW;
                const int iy = iy0 + py, ix = ix0 + pxx;
                const bool in = px < NPIX && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                const f32x4 v = in ? pv[u] : f32x4{0.f, 0.f, 0.f, 0.f};
                float *d = sx + px * XS + q4 * 4;
                d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
                if (q4 == 0) sval[px] = in ? 1.0f : 0.0f;
            }
        }
    }
    float *yb = p.y + (int64_t)b * p.Ho * p.Wo * p.ldy;
    for (int chunk = 0; chunk < p.chunks; ++chunk) {
        const int c0 = chunk * 32;
        const int cq = c0 + q * 4;
        const bool qv = cq < p.Cexp;                   // Cexp % 4 == 0
        // this chunk's depthwise taps -> LDS (requested during the previous chunk); the barrier below also covers the patch
        if (tid < K * K * 8) *reinterpret_cast<f32x4 *>(&swd[(tid >> 3) * 32 + q * 4]) = qv ? tapv : f32x4{0.f, 0.f, 0.f, 0.f};
        const f32x4 csh1 = sh1;
        if (chunk == 0) __syncthreads();

        // 2. expand: D[pixel][channel] = sum_k X[pixel][k] * We[channel][k]   (32x32x2: A = pixels, B = channels)
        for (int mb = wave; mb < MB; mb += 4) {
            f32x16 acc;                                 // BN0: the scale is folded into the weights, the shift starts the sum
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = sh0;
            const float *arow = sx + (mb * 32 + (lane & 31)) * XS + (lane >> 5);
#pragma unroll
            for (int k = 0; k < Cin / 2; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(arow[2 * k], wf[k], acc, 0, 0, 0);
            // accumulator row of register i: (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5); column = lane & 31
            if (edge) {                                 // pixels outside the image are the depthwise conv's zero padding
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int row = mb * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                    se[row * 32 + (lane & 31)] = acc[i] * mydet_sigmoid_fast(acc[i]) * sval[row];
                }
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int row = mb * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                    se[row * 32 + (lane & 31)] = acc[i] * mydet_sigmoid_fast(acc[i]);
                }
            }
        }
        if (chunk + 1 < p.chunks) request(c0 + 32);    // next chunk's constants fly under the depthwise phase
        __syncthreads();

        // 3. depthwise + BN1 + swish + stores + SE partial sums
        f32x4 sum = {0.f, 0.f, 0.f, 0.f};
        if constexpr (ST == 1) {
            const int row = tid >> 5, strip = (tid >> 3) & 3;          // 8 rows x 4 strips of 4 pixels
            f32x4 acc[4];                               // BN1: scale folded into the taps, shift starts the sum
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = csh1;
            if (qv) {
#pragma unroll 1
                for (int kh = 0; kh < K; ++kh) {                    // one tap row at a time: K + 3 columns + K weights in registers
                    f32x4 col[K + 3];
#pragma unroll
                    for (int c = 0; c < K + 3; ++c) col[c] = *reinterpret_cast<const f32x4 *>(&se[((row + kh) * IW + strip * 4 + c) * 32 + q * 4]);
#pragma unroll
                    for (int kw = 0; kw < K; ++kw) {
                        const f32x4 wv = *reinterpret_cast<const f32x4 *>(&swd[(kh * K + kw) * 32 + q * 4]);
#pragma unroll
                        for (int u = 0; u < 4; ++u)
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[u][e] = fmaf(col[u + kw][e], wv[e], acc[u][e]);
                    }
                }
                const int oy = oy0 + row;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int ox = ox0 + strip * 4 + u;
                    if (oy < p.Ho && ox < p.Wo) {
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[e] = acc[u][e] * mydet_sigmoid_fast(acc[u][e]);
                            sum[e] += v[e];
                        }
                        *reinterpret_cast<f32x4 *>(yb + ((int64_t)oy * p.Wo + ox) * p.ldy + cq) = v;
                    }
                }
            }
        } else {
            const int opx = tid >> 3;                                  // 32 output pixels (4 x 8)
            const int row = opx >> 3, colx = opx & 7;
            f32x4 acc = csh1;
            if (qv) {
#pragma unroll
                for (int kh = 0; kh < K; ++kh)
#pragma unroll
                    for (int kw = 0; kw < K; ++kw) {
                        const f32x4 v = *reinterpret_cast<const f32x4 *>(&se[((row * 2 + kh) * IW + colx * 2 + kw) * 32 + q * 4]);
                        const f32x4 wv = *reinterpret_cast<const f32x4 *>(&swd[(kh * K + kw) * 32 + q * 4]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[e] = fmaf(v[e], wv[e], acc[e]);
                    }
                const int oy = oy0 + row, ox = ox0 + colx;
                if (oy < p.Ho && ox < p.Wo) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = acc[e] * mydet_sigmoid_fast(acc[e]);
                        sum[e] += v[e];
                    }
                    *reinterpret_cast<f32x4 *>(yb + ((int64_t)oy * p.Wo + ox) * p.ldy + cq) = v;
                }
            }
        }
        // SE squeeze: channel sums of this tile in a fixed order (butterfly over the 8 lanes of a wave that share a
        // quad, then the four waves in order) -- deterministic, no atomics
#pragma unroll
        for (int off = 8; off < 64; off <<= 1)
#pragma unroll
            for (int e = 0; e < 4; ++e) sum[e] += __shfl_xor(sum[e], off);
        if ((p.partial || se_on) && lane < 8) red[wave * 8 + lane] = sum;
        __syncthreads();                               // expanded tile and taps are free for the next chunk; red is visible
        if ((p.partial || se_on) && tid < 8) {
            f32x4 tot = {0.f, 0.f, 0.f, 0.f};
            if (qv) {
                tot = red[tid];
#pragma unroll
                for (int w = 1; w < 4; ++w)
#pragma unroll
                    for (int e = 0; e < 4; ++e) tot[e] += red[w * 8 + tid][e];
                if (p.partial) *reinterpret_cast<f32x4 *>(p.partial + ((int64_t)b * (p.S + 1) + r) * p.Cexp + cq) = tot;
            }
            if (se_on) *reinterpret_cast<f32x4 *>(&tots[c0 + tid * 4]) = tot;
        }
    }
    if (se_on) {        // this tile's share of W1 . sums over all its channels, once, then the per-image hand-over (se_tail.h)
        if (tid < MYDET_SE_MAX_CSE) selds[tid] = 0.f;
        __syncthreads();
        se_fc1_accumulate(p.se, p.Cexp, tots, 0, p.Cexp, selds);
        __syncthreads();
        se_tail_finish(p.se, selds, p.Cexp, p.Ho * p.Wo, b, r, p.tiles_per_img, (int)gridDim.x / p.tiles_per_img, se_ep);
    }
}

inline bool al16(const void *p) { return ((uintptr_t)p & 15) == 0; }

template <int K, int ST, int CIN>
int launch(MbArgs &p, int B, hipStream_t stream) {
    using G = MbGeom<K, ST>;
    p.tiles_x = (p.Wo + G::TW - 1) / G::TW;
    p.tiles_per_img = p.tiles_x * ((p.Ho + G::TH - 1) / G::TH);
    if ((p.partial || p.se.gate) && p.S != p.tiles_per_img) return MYDET_E_BADARG;
    p.chunks = (p.Cexp + 31) / 32;
    const size_t lds = ((size_t)((G::NPIXP * (p.Cin + 1) + 3) & ~3) + (size_t)G::NPIXP * 33 + (size_t)K * K * 32 + 128 +
                        (p.se.gate ? (size_t)MYDET_SE_LDS_FLOATS + (size_t)p.chunks * 32 : 0)) * sizeof(float);
    if (lds > 80 * 1024) return MYDET_E_UNSUPP;
    const int64_t grid = (int64_t)B * p.tiles_per_img;
    if (grid > 0x7fffffff) return MYDET_E_UNSUPP;
    hipLaunchKernelGGL((mbconv_expand_dw_kernel<K, ST, CIN>), dim3((unsigned)grid), dim3(256), lds, stream, p);
    return mydet_launch_status();
}

}  // namespace

extern "C" int mydet_mbconv_tiles(int Ho, int Wo, int stride) {
    if (Ho <= 0 || Wo <= 0 || (stride != 1 && stride != 2)) return 0;
    const int th = stride == 1 ? 8 : 4, tw = stride == 1 ? 16 : 8;
    return ((Ho + th - 1) / th) * ((Wo + tw - 1) / tw);
}

extern "C" int mydet_mbconv_expand_dw_f32(const float *x, int64_t ldx, const float *w_expand, const float *shift0,
                                          const float *w_dw, const float *shift1, float *y, int64_t ldy, int B, int H, int W, int Cin,
                                          int Cexp, int K, int stride, int pad_t, int pad_l, int Ho, int Wo,
                                          float *se_partial, int S, const mydet_se_tail *se, void *stream) {
    if (!x || !w_expand || !shift0 || !w_dw || !shift1 || !y) return MYDET_E_BADARG;
    if (B <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0 || Cin <= 0 || Cexp <= 0) return MYDET_E_BADARG;
    if ((Cin & 3) || (Cexp & 3) || (ldx & 3) || (ldy & 3) || ldx < Cin || ldy < Cexp) return MYDET_E_BADARG;
    if (!al16(x) || !al16(y) || !al16(w_dw) || !al16(shift1) || (se_partial && !al16(se_partial)))
        return MYDET_E_BADARG;
    if ((K != 3 && K != 5) || (stride != 1 && stride != 2)) return MYDET_E_UNSUPP;
    MbArgs p;
    p.x = x; p.we = w_expand; p.shift0 = shift0; p.wd = w_dw; p.shift1 = shift1;
    p.y = y; p.partial = se_partial; p.ldx = ldx; p.ldy = ldy; p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo; p.Cin = Cin;
    p.Cexp = Cexp; p.pad_t = pad_t; p.pad_l = pad_l; p.S = S;
    const SeTail none = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0};
    p.se = se ? *se : none;
    if (const int e = mydet_se_tail_check(p.se, Cexp, B, S)) return e;
    hipStream_t st = (hipStream_t)stream;
    // the (kernel, stride, Cin) combinations of EfficientNet-B0..B2 stages 2-4 (external/efficientnet/utils.py:258-263)
    if (K == 3 && stride == 2 && Cin == 16) return launch<3, 2, 16>(p, B, st);
    if (K == 3 && stride == 1 && Cin == 24) return launch<3, 1, 24>(p, B, st);
    if (K == 5 && stride == 2 && Cin == 24) return launch<5, 2, 24>(p, B, st);
    if (K == 5 && stride == 1 && Cin == 40) return launch<5, 1, 40>(p, B, st);
    if (K == 3 && stride == 2 && Cin == 40) return launch<3, 2, 40>(p, B, st);
    return MYDET_E_UNSUPP;
}
