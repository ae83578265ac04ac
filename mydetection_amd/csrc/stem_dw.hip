// EfficientNet stem conv fused with the first MBConv block's depthwise conv (the block has expand_ratio 1, so its
// depthwise conv reads the stem's output directly):
//
//     y = swish(BN1( depthwise3x3_s1_pad1( swish(BN0( conv3x3_s2_SAME(image) )) ) ))     + per-tile channel sums of y (SE squeeze)
//
// external/efficientnet/model.py:133-140 (_conv_stem -> _bn0 -> swish) + :76-80 of block 0 (_depthwise_conv -> _bn1 -> swish,
// adaptive_avg_pool2d) with the static "SAME" padding of utils.py:122-145.  Unfused, the 32-channel stem output -- at
// 320^2 the largest tensor of the network, 13 MB per image -- is written by the stem launch and read back by the
// depthwise launch; here it only ever exists as a 10 x 18 tile in LDS and the image is the only input.
// (BatchNorm scales are folded into the weights by the caller, the shifts initialise the accumulators.)
//
// One 256-thread workgroup = one 8 x 16 tile of outputs of one image:
//   1. the 21 x 37 x 3 image patch behind the tile's 10 x 18 stem pixels -> LDS (reads follow the image's own strides,
//      column-fastest; out-of-image values are the conv's zero padding)
//   2. stem conv on FP32 MFMA (v_mfma_f32_32x32x2_f32): [192 stem pixels] x [27 -> 28 taps] x [32 channels]; the A
//      operand is read straight from the patch (address = pixel base + a per-tap constant), the weights sit in registers;
//      swish; stem pixels outside the map become the zeros the depthwise conv pads with; result -> LDS [pixel][32]
//   3. depthwise 3x3 on the VALU from LDS exactly as mbconv_expand_dw_kernel<3, 1, .> (taps in (kh, kw) order, one fmaf
//      chain per channel), swish, 128-byte-per-pixel stores, and the tile's channel sums (fixed order, no atomics).
// Built with -ffp-contract=off like the other EfficientNet kernels.
#include "common.h"
#include "se_tail.h"

namespace {

constexpr int SD_TH = 8, SD_TW = 16, SD_SH = SD_TH + 2, SD_SW = SD_TW + 2;      // stem pixels behind a tile
constexpr int SD_NPIX = SD_SH * SD_SW, SD_NPIXP = 192, SD_MB = SD_NPIXP / 32;
constexpr int SD_IH = 2 * (SD_SH - 1) + 3, SD_IW = 2 * (SD_SW - 1) + 3;         // image patch 21 x 37
constexpr int SD_IMG = SD_IH * SD_IW * 3, SD_IMGP = (SD_IMG + 3) & ~3;
constexpr int SD_C = 32, SD_K = 27;

struct SdArgs {
    const float *x, *ws, *shift0, *wd, *shift1;
    float *y, *partial;
    int64_t sxb, sxc, sxh, sxw, ldy;
    int H, W, Hs, Ws, pad_t, pad_l, tiles_x, tiles_per_img, S;
    SeTail se;                // se.gate != NULL: block 0's squeeze-excite gate is finished inside the launch (se_tail.h)
};

__global__ __launch_bounds__(256, 2) void stem_dw_kernel(const SdArgs p) {
    __shared__ __attribute__((aligned(16))) float simg[SD_IMGP];              // [row][col][c]
    __shared__ __attribute__((aligned(16))) float se[SD_NPIXP * SD_C];         // stem tile [pixel][32]
    __shared__ __attribute__((aligned(16))) float swd[9 * SD_C];               // depthwise taps
    __shared__ float sval[SD_NPIXP];                                          // 1 inside the stem map, 0 outside
    __shared__ f32x4 red[32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t = blockIdx.x;
    const int b = t / p.tiles_per_img, r = t - b * p.tiles_per_img;
    const unsigned se_ep = p.se.gate != nullptr ? se_epoch(p.se, b) : 0u;
    const int ty = r / p.tiles_x, tx = r - ty * p.tiles_x;
    const int oy0 = ty * SD_TH, ox0 = tx * SD_TW;
    const int sy0 = oy0 - 1, sx0 = ox0 - 1;                                   // first stem pixel of the tile (may be -1)
    const int iy0 = 2 * sy0 - p.pad_t, ix0 = 2 * sx0 - p.pad_l;               // first image pixel of the patch

    // constants: this lane's stem weights (B operand: channel = lane & 31, tap = 2 i + (lane >> 5)), shifts, taps
    float wf[SD_K / 2 + 1];
    {
        const float *wrow = p.ws + (lane & 31) * SD_K;
        const int kk = lane >> 5;
#pragma unroll
        for (int i = 0; i < SD_K / 2 + 1; ++i) {
            const int k = 2 * i + kk;
            wf[i] = wrow[k < SD_K ? k : SD_K - 1];
            if (k >= SD_K) wf[i] = 0.f;
        }
    }
    const float sh0 = p.shift0[lane & 31];
    const int q = tid & 7;
    const f32x4 sh1 = *reinterpret_cast<const f32x4 *>(p.shift1 + q * 4);
    if (tid < 9 * 8) *reinterpret_cast<f32x4 *>(&swd[tid * 4]) = *reinterpret_cast<const f32x4 *>(p.wd + tid * 4);

    // 1. image patch -> LDS.  Element e = (c, row, col), col fastest: consecutive threads read consecutive image columns
    {
        constexpr int NIT = (SD_IMG + 255) / 256;
        const float *xb = p.x + (int64_t)b * p.sxb;
        float pv[NIT];
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int e = min(tid + u * 256, SD_IMG - 1);
            const int c = e / (SD_IH * SD_IW), rc = e - c * (SD_IH * SD_IW);
            const int row = rc / SD_IW, col = rc - row * SD_IW;
            const int cy = min(max(iy0 + row, 0), p.H - 1), cx = min(max(ix0 + col, 0), p.W - 1);
            pv[u] = xb[(int64_t)c * p.sxc + (int64_t)cy * p.sxh + (int64_t)cx * p.sxw];              // unconditional, clamped
        }
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int e = tid + u * 256;
            if (e < SD_IMG) {
                const int c = e / (SD_IH * SD_IW), rc = e - c * (SD_IH * SD_IW);
                const int row = rc / SD_IW, col = rc - row * SD_IW;
                const bool in = (unsigned)(iy0 + row) < (unsigned)p.H && (unsigned)(ix0 + col) < (unsigned)p.W;
                simg[(row * SD_IW + col) * 3 + c] = in ? pv[u] : 0.f;
            }
        }
        if (tid < SD_NPIXP) {
            const int sy = sy0 + tid / SD_SW, sx = sx0 + tid % SD_SW;
            sval[tid] = tid < SD_NPIX && (unsigned)sy < (unsigned)p.Hs && (unsigned)sx < (unsigned)p.Ws ? 1.0f : 0.0f;
        }
    }
    __syncthreads();

    // 2. stem conv: D[pixel][channel] = sum_k A[pixel][k] * W[channel][k]  (32x32x2: A = pixels, B = channels)
    for (int mb = wave; mb < SD_MB; mb += 4) {
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = sh0;
        const int spx = min(mb * 32 + (lane & 31), SD_NPIX - 1);
        const int sy_l = spx / SD_SW, sx_l = spx - sy_l * SD_SW;
        // tap k = (kh*3 + kw)*3 + c sits at patch offset ((kh*37 + kw)*3 + c) = k + kh * (37*3 - 9) from the pixel's base
        const float *abase = simg + ((2 * sy_l) * SD_IW + 2 * sx_l) * 3;
        const int kk = lane >> 5;
#pragma unroll
        for (int i = 0; i < SD_K / 2 + 1; ++i) {
            const int k0 = 2 * i, k1 = 2 * i + 1 < SD_K ? 2 * i + 1 : SD_K - 1;                      // compile-time
            const int off0 = k0 + (k0 / 9) * (SD_IW * 3 - 9), off1 = k1 + (k1 / 9) * (SD_IW * 3 - 9);
            const float av = abase[kk ? off1 : off0];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, wf[i], acc, 0, 0, 0);
        }
        // accumulator row of register i: (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5); column = lane & 31
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = mb * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
            se[row * SD_C + (lane & 31)] = acc[i] * mydet_sigmoid_fast(acc[i]) * sval[row];
        }
    }
    __syncthreads();

    // 3. depthwise 3x3 + BN1 + swish + stores + SE partial sums (8 rows x 4 strips of 4 pixels x 8 channel quads)
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    {
        const int row = tid >> 5, strip = (tid >> 3) & 3;
        f32x4 acc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = sh1;
#pragma unroll 1
        for (int kh = 0; kh < 3; ++kh) {
            f32x4 col[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) col[c] = *reinterpret_cast<const f32x4 *>(&se[((row + kh) * SD_SW + strip * 4 + c) * SD_C + q * 4]);
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const f32x4 wv = *reinterpret_cast<const f32x4 *>(&swd[(kh * 3 + kw) * SD_C + q * 4]);
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[u][e] = fmaf(col[u + kw][e], wv[e], acc[u][e]);
            }
        }
        const int oy = oy0 + row;
        float *yb = p.y + (int64_t)b * p.Hs * p.Ws * p.ldy;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int ox = ox0 + strip * 4 + u;
            if (oy < p.Hs && ox < p.Ws) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = acc[u][e] * mydet_sigmoid_fast(acc[u][e]);
                    sum[e] += v[e];
                }
                *reinterpret_cast<f32x4 *>(yb + ((int64_t)oy * p.Ws + ox) * p.ldy + q * 4) = v;
            }
        }
    }
    // channel sums of this tile in a fixed order (butterfly over the 8 lanes of a wave that share a quad, then the four
    // waves in order) -- deterministic, no atomics
#pragma unroll
    for (int off = 8; off < 64; off <<= 1)
#pragma unroll
        for (int e = 0; e < 4; ++e) sum[e] += __shfl_xor(sum[e], off);
    const bool se_on = p.se.gate != nullptr;
    if ((p.partial || se_on) && lane < 8) red[wave * 8 + lane] = sum;
    __syncthreads();                                   // (every thread is done with the stem tile `se`: the tail reuses it)
    float *selds = se, *tots = se + MYDET_SE_LDS_FLOATS;
    if ((p.partial || se_on) && tid < 8) {
        f32x4 tot = red[tid];
#pragma unroll
        for (int w = 1; w < 4; ++w)
#pragma unroll
            for (int e = 0; e < 4; ++e) tot[e] += red[w * 8 + tid][e];
        if (p.partial) *reinterpret_cast<f32x4 *>(p.partial + ((int64_t)b * (p.S + 1) + r) * SD_C + tid * 4) = tot;
        if (se_on) *reinterpret_cast<f32x4 *>(&tots[tid * 4]) = tot;
    }
    if (se_on) {        // this tile's share of W1 . sums, then the per-image hand-over (se_tail.h)
        if (tid >= 64 && tid < 64 + MYDET_SE_MAX_CSE) selds[tid - 64] = 0.f;
        __syncthreads();
        se_fc1_accumulate(p.se, SD_C, tots, 0, SD_C, selds);
        __syncthreads();
        se_tail_finish(p.se, selds, SD_C, p.Hs * p.Ws, b, r, p.tiles_per_img, (int)gridDim.x / p.tiles_per_img, se_ep);
    }
}

inline bool al16(const void *p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

extern "C" int mydet_stem_dw_f32(const float *x, int64_t sxb, int64_t sxc, int64_t sxh, int64_t sxw, const float *w_stem,
                                 const float *shift0, const float *w_dw, const float *shift1, float *y, int64_t ldy, int B, int H,
                                 int W, int C, int pad_t, int pad_l, int Hs, int Ws, float *se_partial, int S,
                                 const mydet_se_tail *se, void *stream) {
    if (!x || !w_stem || !shift0 || !w_dw || !shift1 || !y || B <= 0 || H <= 0 || W <= 0 || Hs <= 0 || Ws <= 0)
        return MYDET_E_BADARG;
    if (C != SD_C) return MYDET_E_UNSUPP;
    if ((ldy & 3) || ldy < C || !al16(y) || !al16(w_dw) || !al16(shift1) || (se_partial && !al16(se_partial))) return MYDET_E_BADARG;
    if (pad_t < 0 || pad_l < 0 || pad_t > 2 || pad_l > 2) return MYDET_E_BADARG;
    // the stem map the caller announces must be the one a 3x3 stride-2 conv can produce from the padded image
    if (2 * (Hs - 1) - pad_t >= H || 2 * (Ws - 1) - pad_l >= W) return MYDET_E_BADARG;
    SdArgs p;
    p.x = x; p.ws = w_stem; p.shift0 = shift0; p.wd = w_dw; p.shift1 = shift1; p.y = y; p.partial = se_partial;
    p.sxb = sxb; p.sxc = sxc; p.sxh = sxh; p.sxw = sxw; p.ldy = ldy;
    p.H = H; p.W = W; p.Hs = Hs; p.Ws = Ws; p.pad_t = pad_t; p.pad_l = pad_l;
    p.tiles_x = (Ws + SD_TW - 1) / SD_TW;
    p.tiles_per_img = p.tiles_x * ((Hs + SD_TH - 1) / SD_TH);
    p.S = S;
    const SeTail none = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0};
    p.se = se ? *se : none;
    if (const int e = mydet_se_tail_check(p.se, C, B, S)) return e;
    if ((se_partial || p.se.gate) && S != p.tiles_per_img) return MYDET_E_BADARG;
    const int64_t grid = (int64_t)B * p.tiles_per_img;
    if (grid > 0x7fffffff) return MYDET_E_UNSUPP;
    hipLaunchKernelGGL(stem_dw_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, p);
    return mydet_launch_status();
}
