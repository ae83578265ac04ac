// HBM-bound kernels of the EfficientNet / BiFPN / EfDetHead family (NHWC, 16-byte lanes).
//
//   dwconv          depthwise k x k (k = 3, 5), stride 1/2, asymmetric "static SAME" pads, optional folded
//                   BN + swish epilogue.  external/efficientnet/model.py:77 (_depthwise_conv),
//                   models/modules.py:12-19 (SeparableConv2d.depthwise)
//   squeeze_partial per-image channel sums over a slice of pixels (stage 1 of the SE global average pool)
//   se_gate         stage 2 + both SE 1x1 convs: gate = sigmoid(W2 . swish(W1 . mean + b1) + b2)
//                   external/efficientnet/model.py:80-83
//   maxpool3s2      3x3 stride-2 pad-1 max pool (-inf padding)  models/backbones.py:186,188; models/fpns.py:405-416
//   bifpn_fuse      y = swish(sum_i w_i * in_i), w = relu(weights)/(sum + 1e-4), where an input may be read
//                   through nearest 2x upsampling or the 3x3/2 max pool on the fly  models/fpns.py:398-418,433-439
// Each thread moves one float4 of channels; every tensor byte is read and written once per kernel.
// Built with -ffp-contract=off so the fusion / gate arithmetic rounds like the reference's separate ops.
#include <cstdlib>

#include "common.h"
#include "se_tail.h"

namespace {

const SeTail NO_SE_TAIL = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0};

// --------------------------------------------------------------------------------------------- depthwise
struct DwArgs {
    const float *x, *w, *scale, *shift;
    float *y, *partial;       // partial != NULL: also emit per-slice channel sums [B][S][C] (SE squeeze)
    int64_t ldx, ldy, total;
    int C, H, W, Ho, Wo, pad_t, pad_l, act, S;
    SeTail se;                // se.gate != NULL: the squeeze-excite gate is finished inside the launch (se_tail.h)
};

// TW consecutive outputs of one row for one channel quad: each input column and each weight is loaded once
// for the whole strip ((TW-1)*ST + K columns instead of TW*K).
template <int K, int ST, int TW>
__device__ __forceinline__ void dw_strip(const DwArgs &p, const float *xq, const float *wq, int oh, int ow0,
                                         f32x4 (&acc)[TW]) {
    constexpr int NC = (TW - 1) * ST + K;
#pragma unroll
    for (int t = 0; t < TW; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ih0 = oh * ST - p.pad_t, iw0 = ow0 * ST - p.pad_l;
#pragma unroll
    for (int kh = 0; kh < K; ++kh) {
        const int ih = ih0 + kh;
        if ((unsigned)ih >= (unsigned)p.H) continue;
        const float *row = xq + (int64_t)ih * p.W * p.ldx;
        f32x4 col[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int iw = iw0 + c;
            col[c] = (unsigned)iw < (unsigned)p.W ? *reinterpret_cast<const f32x4 *>(row + (int64_t)iw * p.ldx)
                                                  : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int kw = 0; kw < K; ++kw) {
            const f32x4 wv = *reinterpret_cast<const f32x4 *>(wq + (kh * K + kw) * p.C);
#pragma unroll
            for (int t = 0; t < TW; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[t][j] = fmaf(col[t * ST + kw][j], wv[j], acc[t][j]);
        }
    }
}

// Stride-1 block of TW x 2 outputs (rows oh, oh+1) for one channel quad: K+1 input rows instead of 2K, and each
// weight row is loaded once and serves both output rows (kept in registers for the next input row).
template <int K, int TW>
__device__ __forceinline__ void dw_block2(const DwArgs &p, const float *xq, const float *wq, int oh, int ow0,
                                          f32x4 (&acc)[2][TW]) {
    constexpr int NC = TW - 1 + K;
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int t = 0; t < TW; ++t) acc[r][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ih0 = oh - p.pad_t, iw0 = ow0 - p.pad_l;
    f32x4 wprev[K];
#pragma unroll
    for (int r = 0; r <= K; ++r) {                     // input row ih0 + r: tap row r of output row 0, r-1 of row 1
        f32x4 wcur[K];
        if (r < K) {
#pragma unroll
            for (int kw = 0; kw < K; ++kw) wcur[kw] = *reinterpret_cast<const f32x4 *>(wq + (r * K + kw) * p.C);
        }
        const int ih = ih0 + r;
        if ((unsigned)ih < (unsigned)p.H) {
            const float *row = xq + (int64_t)ih * p.W * p.ldx;
            f32x4 col[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int iw = iw0 + c;
                col[c] = (unsigned)iw < (unsigned)p.W ? *reinterpret_cast<const f32x4 *>(row + (int64_t)iw * p.ldx)
                                                      : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int kw = 0; kw < K; ++kw)
#pragma unroll
                for (int t = 0; t < TW; ++t)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (r < K) acc[0][t][j] = fmaf(col[t + kw][j], wcur[kw][j], acc[0][t][j]);
                        if (r > 0) acc[1][t][j] = fmaf(col[t + kw][j], wprev[kw][j], acc[1][t][j]);
                    }
        }
#pragma unroll
        for (int kw = 0; kw < K; ++kw) wprev[kw] = wcur[kw];
    }
}

__device__ __forceinline__ f32x4 dw_epilogue(const DwArgs &p, f32x4 v, int q) {
    if (p.scale) {
        const f32x4 sc = *reinterpret_cast<const f32x4 *>(p.scale + q * 4);
        const f32x4 sh = *reinterpret_cast<const f32x4 *>(p.shift + q * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = v[j] * sc[j] + sh[j];
    }
    if (p.act == MYDET_ACT_SWISH) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = v[j] * mydet_sigmoid_fast(v[j]);
    } else if (p.act == MYDET_ACT_LEAKY) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : v[j] * 0.1f;
    }
    return v;
}

// plain form: flat grid-stride over (strip, quad) items
template <int K, int ST, int TW, int TH = 1>
__global__ __launch_bounds__(256) void dwconv_kernel(const DwArgs p) {
    const int Q = p.C >> 2, WG = p.Wo / TW, HG = p.Ho / TH;     // TH = 2: items are TW x 2 blocks (stride 1, Ho even)
    for (int64_t it = (int64_t)blockIdx.x * 256 + threadIdx.x; it < p.total; it += (int64_t)gridDim.x * 256) {
        const int q = (int)(it % Q);
        const int64_t g = it / Q;
        const int owg = (int)(g % WG);
        const int64_t t = g / WG;
        const int ohg = (int)(t % HG);
        const int64_t b = t / HG;
        if constexpr (TH == 2) {
            f32x4 acc[2][TW];
            dw_block2<K, TW>(p, p.x + (b * p.H * p.W) * p.ldx + q * 4, p.w + q * 4, ohg * 2, owg * TW, acc);
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                float *yp = p.y + ((b * p.Ho + ohg * 2 + r) * p.Wo + owg * TW) * p.ldy + q * 4;
#pragma unroll
                for (int tt = 0; tt < TW; ++tt) *reinterpret_cast<f32x4 *>(yp + tt * p.ldy) = dw_epilogue(p, acc[r][tt], q);
            }
        } else {
            f32x4 acc[TW];
            dw_strip<K, ST, TW>(p, p.x + (b * p.H * p.W) * p.ldx + q * 4, p.w + q * 4, ohg, owg * TW, acc);
            float *yp = p.y + ((b * p.Ho + ohg) * p.Wo + owg * TW) * p.ldy + q * 4;
#pragma unroll
            for (int tt = 0; tt < TW; ++tt) *reinterpret_cast<f32x4 *>(yp + tt * p.ldy) = dw_epilogue(p, acc[tt], q);
        }
    }
}

// squeeze-fused form: workgroup (s, b) owns slice s of image b's strips, all channels; besides y it writes
// partial[b][s][c] = sum of its outputs (deterministic: fixed thread->strip map, fixed reduction order).
template <int K, int ST, int TW, int TH = 1>
__global__ __launch_bounds__(256) void dwconv_sum_kernel(const DwArgs p) {
    __shared__ f32x4 red[256];
    __shared__ float selds[MYDET_SE_LDS_FLOATS];
    const bool se_on = p.se.gate != nullptr;
    if (se_on && threadIdx.x < MYDET_SE_MAX_CSE) selds[threadIdx.x] = 0.f;      // (the first barrier below orders it)
    const int Q = p.C >> 2, WG = p.Wo / TW;
    const int b = blockIdx.y, s = blockIdx.x;
    const unsigned se_ep = se_on ? se_epoch(p.se, b) : 0u;
    const int NG = (p.Ho / TH) * WG;                   // TH = 2: a work item is a TW x 2 block (stride 1, Ho even)
    const int per = (NG + p.S - 1) / p.S;
    const int g0 = s * per, g1 = min(NG, g0 + per);
    const float *xb = p.x + ((int64_t)b * p.H * p.W) * p.ldx;
    float *yb = p.y + ((int64_t)b * p.Ho * p.Wo) * p.ldy;
    for (int qb = 0; qb < Q; qb += 256) {
        const int nq = min(Q - qb, 256);
        const int ph_n = 256 / nq;                       // >= 1
        const int q = qb + threadIdx.x % nq, ph = threadIdx.x / nq;
        f32x4 sum = {0.f, 0.f, 0.f, 0.f};
        if (ph < ph_n)
            for (int g = g0 + ph; g < g1; g += ph_n) {
                const int ohg = g / WG, owg = g - ohg * WG;
                if constexpr (TH == 2) {
                    f32x4 acc[2][TW];
                    dw_block2<K, TW>(p, xb + q * 4, p.w + q * 4, ohg * 2, owg * TW, acc);
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        float *yp = yb + ((int64_t)(ohg * 2 + r) * p.Wo + owg * TW) * p.ldy + q * 4;
#pragma unroll
                        for (int tt = 0; tt < TW; ++tt) {
                            const f32x4 v = dw_epilogue(p, acc[r][tt], q);
                            *reinterpret_cast<f32x4 *>(yp + tt * p.ldy) = v;
#pragma unroll
                            for (int j = 0; j < 4; ++j) sum[j] += v[j];
                        }
                    }
                } else {
                    f32x4 acc[TW];
                    dw_strip<K, ST, TW>(p, xb + q * 4, p.w + q * 4, ohg, owg * TW, acc);
                    float *yp = yb + ((int64_t)ohg * p.Wo + owg * TW) * p.ldy + q * 4;
#pragma unroll
                    for (int tt = 0; tt < TW; ++tt) {
                        const f32x4 v = dw_epilogue(p, acc[tt], q);
                        *reinterpret_cast<f32x4 *>(yp + tt * p.ldy) = v;
#pragma unroll
                        for (int j = 0; j < 4; ++j) sum[j] += v[j];
                    }
                }
            }
        red[threadIdx.x] = sum;
        __syncthreads();
        f32x4 tot = {0.f, 0.f, 0.f, 0.f};
        if (threadIdx.x < nq) {
            tot = red[threadIdx.x];
            for (int k = 1; k < ph_n; ++k) {
                const f32x4 o = red[threadIdx.x + k * nq];
#pragma unroll
                for (int j = 0; j < 4; ++j) tot[j] += o[j];
            }
            if (p.partial) *reinterpret_cast<f32x4 *>(p.partial + ((int64_t)b * (p.S + 1) + s) * p.C + (qb + threadIdx.x) * 4) = tot;
        }
        __syncthreads();
        if (se_on) {            // this slice's share of W1 . sums for the channels of this group (se_tail.h)
            if (threadIdx.x < nq) red[threadIdx.x] = tot;
            __syncthreads();
            se_fc1_accumulate(p.se, p.C, reinterpret_cast<const float *>(red), qb * 4, nq * 4, selds);
            __syncthreads();
        }
    }
    if (se_on) se_tail_finish(p.se, selds, p.C, p.Ho * p.Wo, b, s, p.S, (int)gridDim.y, se_ep);
}


// ------------------------------------------------------------------------------------ depthwise, LDS-tiled (stride 1)
// The register-blocked kernels above keep K+1 input rows x (TW+K-1) columns of one channel quad per thread: 183 VGPRs
// for k = 5 (two waves per SIMD) and one dependent L2 / HBM round trip per input row -- 1.7-2.3 TB/s on the
// EfficientNet 40^2 / 80^2 layers.  Here a workgroup owns 8 x 16 outputs x 8 channel quads of one image: the
// (8+K-1) x (16+K-1) input patch is requested in ONE batch (unconditional loads at clamped coordinates, zeros by
// select), lands in LDS, and every thread then forms a strip of four outputs of one quad from LDS (K+3 column reads per
// tap row, taps from LDS) in the same (kh, kw) fmaf order as dw_strip.  35 KB of LDS, < 64 VGPRs: four workgroups
// per CU keep ~30 KB of loads in flight each.  Squeeze sums per (image, tile, channel) in a fixed order.
struct DwTArgs {
    const float *x, *w, *scale, *shift;
    float *y, *partial;
    int64_t ldx, ldy;
    int C, H, W, Ho, Wo, pad_t, pad_l, act, S, tiles_x, nchunks;
    SeTail se;
};

// CQ channel quads per workgroup: 8 (tile 8 x 16) for C >= 32, 4 (tile 8 x 32) for the 16-channel layer
template <int K, bool SUM, int CQ = 8>
__global__ __launch_bounds__(256, 4) void dwconv_tile_kernel(const DwTArgs p) {
    constexpr int TH = 8, TW = 128 / CQ, IH = TH + K - 1, IW = TW + K - 1, NPIX = IH * IW, PS = CQ * 4 + 4;
    constexpr int NL = (NPIX * CQ + 255) / 256;
    __shared__ __attribute__((aligned(16))) float tile[NPIX * PS];
    __shared__ __attribute__((aligned(16))) float wl[K * K * CQ * 4];
    __shared__ f32x4 red[SUM ? 256 : 1];
    const int tid = threadIdx.x;
    const int chunk = blockIdx.x % p.nchunks;          // channel chunks of one tile are neighbours: 128-B runs of one pixel row
    const int t = blockIdx.x / p.nchunks;
    const int b = t / p.S, r = t - b * p.S;
    const int ty = r / p.tiles_x, tx = r - ty * p.tiles_x;
    const int oh0 = ty * TH, ow0 = tx * TW;
    const int ih0 = oh0 - p.pad_t, iw0 = ow0 - p.pad_l;
    const int Q = p.C >> 2, q0 = chunk * CQ;
    const float *xb = p.x + (int64_t)b * p.H * p.W * p.ldx;
    unsigned se_ep = 0u;
    if (SUM && p.se.gate != nullptr) se_ep = se_epoch(p.se, b);
    // taps of this chunk -> LDS (a quad past the end reads the last quad; its outputs are never stored)
    if (tid < K * K * CQ) {
        const int tap = tid / CQ, q = min(q0 + tid % CQ, Q - 1);
        *reinterpret_cast<f32x4 *>(&wl[tid * 4]) = *reinterpret_cast<const f32x4 *>(p.w + (int64_t)tap * p.C + q * 4);
    }
    // the BatchNorm terms of this thread's channel quad: requested with the patch, not after the taps loop
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (p.scale) {
        const int qs = min(q0 + tid % CQ, Q - 1);
        sc = *reinterpret_cast<const f32x4 *>(p.scale + qs * 4);
        sh = *reinterpret_cast<const f32x4 *>(p.shift + qs * 4);
    }
    // input patch -> LDS, one batch
    f32x4 v[NL];
#pragma unroll
    for (int j = 0; j < NL; ++j) {
        const int it = min(tid + j * 256, NPIX * CQ - 1);
        const int q = min(q0 + it % CQ, Q - 1), px = it / CQ;
        const int py = px / IW, pxx = px - py * IW;
        const int cy = min(max(ih0 + py, 0), p.H - 1), cx = min(max(iw0 + pxx, 0), p.W - 1);
        v[j] = *reinterpret_cast<const f32x4 *>(xb + ((int64_t)cy * p.W + cx) * p.ldx + q * 4);
    }
#pragma unroll
    for (int j = 0; j < NL; ++j) {
        const int it = tid + j * 256;
        if (it < NPIX * CQ) {
            const int px = it / CQ, qq = it % CQ;
            const int py = px / IW, pxx = px - py * IW;
            const bool in = (unsigned)(ih0 + py) < (unsigned)p.H && (unsigned)(iw0 + pxx) < (unsigned)p.W;
            *reinterpret_cast<f32x4 *>(&tile[px * PS + qq * 4]) = in ? v[j] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    __syncthreads();
    // a strip of four outputs of one quad per thread
    const int qq = tid % CQ, strip = (tid / CQ) % (TW / 4), row = tid / (CQ * (TW / 4));
    const int q = q0 + qq;
    f32x4 acc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int kh = 0; kh < K; ++kh) {                   // one tap row at a time: K + 3 columns and its K taps in registers
        f32x4 col[K + 3];
#pragma unroll
        for (int c = 0; c < K + 3; ++c) col[c] = *reinterpret_cast<const f32x4 *>(&tile[((row + kh) * IW + strip * 4 + c) * PS + qq * 4]);
#pragma unroll
        for (int kw = 0; kw < K; ++kw) {
            const f32x4 wv = *reinterpret_cast<const f32x4 *>(&wl[((kh * K + kw) * CQ + qq) * 4]);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[u][e] = fmaf(col[u + kw][e], wv[e], acc[u][e]);
        }
    }
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    const int oh = oh0 + row;
    if (q < Q && oh < p.Ho) {
        float *yp = p.y + (((int64_t)b * p.Ho + oh) * p.Wo + ow0 + strip * 4) * p.ldy + q * 4;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (ow0 + strip * 4 + u < p.Wo) {
                f32x4 o = acc[u];
                if (p.scale) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = o[e] * sc[e] + sh[e];
                }
                if (p.act == MYDET_ACT_SWISH) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = o[e] * mydet_sigmoid_fast(o[e]);
                } else if (p.act == MYDET_ACT_LEAKY) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = o[e] > 0.f ? o[e] : o[e] * 0.1f;
                }
                *reinterpret_cast<f32x4 *>(yp + (int64_t)u * p.ldy) = o;
#pragma unroll
                for (int e = 0; e < 4; ++e) sum[e] += o[e];
            }
        }
    }
    if (SUM) {          // channel sums of the tile: the 32 (row, strip) threads of a quad, added in thread order
        red[tid] = sum;
        __syncthreads();                               // (every thread is done with `tile` here: the tail reuses it)
        const bool se_on = p.se.gate != nullptr;
        float *selds = tile, *tots = tile + MYDET_SE_LDS_FLOATS;      // NPIX * PS >= 6 480 floats
        if (tid < CQ) {
            f32x4 tot = {0.f, 0.f, 0.f, 0.f};
            if (q0 + tid < Q) {
                tot = red[tid];
                for (int k = 1; k < 256 / CQ; ++k) {
                    const f32x4 o = red[tid + k * CQ];
#pragma unroll
                    for (int e = 0; e < 4; ++e) tot[e] += o[e];
                }
                if (p.partial) *reinterpret_cast<f32x4 *>(p.partial + ((int64_t)b * (p.S + 1) + r) * p.C + (q0 + tid) * 4) = tot;
            }
            if (se_on) *reinterpret_cast<f32x4 *>(&tots[tid * 4]) = tot;
        }
        if (se_on) {    // this tile's share of W1 . sums for this chunk's channels, then the per-image hand-over (se_tail.h)
            if (tid >= 64 && tid < 64 + MYDET_SE_MAX_CSE) selds[tid - 64] = 0.f;
            __syncthreads();
            se_fc1_accumulate(p.se, p.C, tots, q0 * 4, min(CQ * 4, p.C - q0 * 4), selds);
            __syncthreads();
            se_tail_finish(p.se, selds, p.C, p.Ho * p.Wo, b, r * p.nchunks + chunk, p.S * p.nchunks, (int)gridDim.x / (p.S * p.nchunks), se_ep);
        }
    }
}

// ------------------------------------------------------------------------------------ SE squeeze + gate
// standalone stage 1: partial[b][s][c] = sum over pixels of slice s of image b
__global__ __launch_bounds__(256) void squeeze_partial_kernel(const float *x, int64_t ldx, int C, int HW, int S,
                                                              float *partial) {
    __shared__ f32x4 red[256];
    const int Q = C >> 2;
    const int b = blockIdx.y, s = blockIdx.x;
    const int per = (HW + S - 1) / S;
    const int p0 = s * per, p1 = min(HW, p0 + per);
    const float *xb = x + (int64_t)b * HW * ldx;
    for (int qb = 0; qb < Q; qb += 256) {               // channel-quad groups of 256 when Q > 256
        const int nq = min(Q - qb, 256);
        const int ph_n = 256 / nq;
        const int q = threadIdx.x % nq, ph = threadIdx.x / nq;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (ph < ph_n)
            for (int px = p0 + ph; px < p1; px += ph_n) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(xb + (int64_t)px * ldx + (qb + q) * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] += v[j];
            }
        red[threadIdx.x] = acc;
        __syncthreads();
        if (threadIdx.x < nq) {
            f32x4 tot = red[threadIdx.x];
            for (int k = 1; k < ph_n; ++k) {
                const f32x4 o = red[threadIdx.x + k * nq];
#pragma unroll
                for (int j = 0; j < 4; ++j) tot[j] += o[j];
            }
            *reinterpret_cast<f32x4 *>(partial + ((int64_t)b * (S + 1) + s) * C + (qb + threadIdx.x) * 4) = tot;
        }
        __syncthreads();
    }
}

// stage 2a: mean[b][c] = sum_s partial[b][s][c] / HW, written into slice S of the partial buffer.
// workgroup (64-channel group, b): 64 lanes over channels x 4 phases over slices, fixed reduction order.
__global__ __launch_bounds__(256) void se_mean_kernel(float *partial, int S, int C, int HW) {
    __shared__ float red[4][64];
    const int b = blockIdx.y, lane = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const float *pb = partial + (int64_t)b * (S + 1) * C;
    float a0 = 0.f, a1 = 0.f;
    if (c < C) {
        int k = ph;
        for (; k + 4 < S; k += 8) {
            a0 += pb[(int64_t)k * C + c];
            a1 += pb[(int64_t)(k + 4) * C + c];
        }
        if (k < S) a0 += pb[(int64_t)k * C + c];
    }
    red[ph][lane] = a0 + a1;
    __syncthreads();
    if (ph == 0 && c < C)
        partial[((int64_t)b * (S + 1) + S) * C + c] = (((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane]) *
                                                      (1.0f / (float)HW);
}

// stage 2b: workgroup (64-channel group, b) recomputes the hidden layer of image b (cheap, L2-resident weights)
// and emits its 64 gate channels with the expand dot product split over 4 k-phases; w2t is the expand weight
// transposed to [Cse][C] so lanes read it coalesced.  Loops are unrolled so several loads are in flight.
__global__ __launch_bounds__(256) void se_gate_kernel(const float *partial, int S, int C, const float *w1,
                                                      const float *b1, int Cse, const float *w2t, const float *b2,
                                                      float *gate) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ float red[4][64];
    float *mean = sm;                 // [C]
    float *hid = sm + C;              // [Cse]
    const int b = blockIdx.y, tid = threadIdx.x;
    const float *mb = partial + ((int64_t)b * (S + 1) + S) * C;
    for (int c = tid; c < C; c += 256) mean[c] = mb[c];
    __syncthreads();
    const int wave = tid >> 6, lane = tid & 63;
    // reduce conv: a wave per output channel, four output channels in flight at a time (independent load / fma
    // streams; each channel's own summation order is what it was one at a time)
    for (int o0 = wave; o0 < Cse; o0 += 16) {
        float a[4][4];
        const float *wr[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int o = o0 + 4 * u < Cse ? o0 + 4 * u : o0;          // surplus slots redo o0, result discarded
            wr[u] = w1 + (int64_t)o * C;
#pragma unroll
            for (int v = 0; v < 4; ++v) a[u][v] = 0.f;
        }
        int c = lane;
        for (; c + 192 < C; c += 256) {
            const float m0 = mean[c], m1 = mean[c + 64], m2 = mean[c + 128], m3 = mean[c + 192];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a[u][0] = fmaf(wr[u][c], m0, a[u][0]);
                a[u][1] = fmaf(wr[u][c + 64], m1, a[u][1]);
                a[u][2] = fmaf(wr[u][c + 128], m2, a[u][2]);
                a[u][3] = fmaf(wr[u][c + 192], m3, a[u][3]);
            }
        }
        for (; c < C; c += 64) {
            const float m0 = mean[c];
#pragma unroll
            for (int u = 0; u < 4; ++u) a[u][0] = fmaf(wr[u][c], m0, a[u][0]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float acc = (a[u][0] + a[u][1]) + (a[u][2] + a[u][3]);
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
            if (lane == 0 && o0 + 4 * u < Cse) {
                acc += b1[o0 + 4 * u];
                hid[o0 + 4 * u] = acc * mydet_sigmoid(acc);
            }
        }
    }
    __syncthreads();
    const int c = blockIdx.x * 64 + lane;             // expand conv + sigmoid: 64 channels x 4 k-phases
    float acc = 0.f;
    if (c < C) {
        float e0 = 0.f, e1 = 0.f;
        int k = wave;
        for (; k + 4 < Cse; k += 8) {
            e0 = fmaf(w2t[(int64_t)k * C + c], hid[k], e0);
            e1 = fmaf(w2t[(int64_t)(k + 4) * C + c], hid[k + 4], e1);
        }
        if (k < Cse) e0 = fmaf(w2t[(int64_t)k * C + c], hid[k], e0);
        acc = e0 + e1;
    }
    red[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && c < C)
        gate[(int64_t)b * C + c] = mydet_sigmoid((((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane]) + b2[c]);
}

// Whole squeeze-excite tail of one image in ONE workgroup (1024 threads): mean over the S slice sums, reduce conv +
// swish, expand conv + sigmoid.  The two-launch form above spent 19 us per block on launch and dependency latency for a
// few hundred KB of L2-resident reads; S*C <= ~70 k floats per image on the D1 shapes, so one CU per image reads it in
// about a microsecond.  Thread (q, ph) sums slices ph, ph + P, ... of channel quad q (P = 1024 / (C/4) phases, 16-byte
// loads), the phases are added in order 0..P-1 (fixed order: deterministic), then the two small matrix-vector products
// run as in se_gate_kernel with 16 waves.
__global__ __launch_bounds__(1024) void se_fused_kernel(float *partial, int S, int C, int HW, const float *w1,
                                                        const float *b1, int Cse, const float *w2t, const float *b2,
                                                        float *gate, int P) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *mean = sm;                                  // [C]
    float *hid = sm + C;                               // [Cse] (+ pad to a multiple of 4)
    f32x4 *red = reinterpret_cast<f32x4 *>(sm + C + ((Cse + 3) & ~3));      // [P][C/4]
    const int b = blockIdx.x, tid = threadIdx.x;
    const int Q = C >> 2;
    float *pb = partial + (int64_t)b * (S + 1) * C;
    // The kernel is a chain of dependent round trips (slice sums -> mean -> reduce-conv weights -> expand-conv weights) on
    // one CU per image.  The weights do not depend on the data: the first 512 input channels of this wave's two reduce-conv
    // rows and the first eight rows of this thread's expand-conv column are requested here, before the first barrier, and
    // consumed in place of the same loads below (same values, same order of additions).
    const int pwave = tid >> 6, plane = tid & 63;
    float p0[8], p1[8], q2[8];
    {
        const int o0 = pwave < Cse ? pwave : 0, o1 = pwave + 16 < Cse ? pwave + 16 : o0;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int c = plane + 64 * t;
            p0[t] = c < C ? w1[(int64_t)o0 * C + c] : 0.f;
            p1[t] = c < C ? w1[(int64_t)o1 * C + c] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) q2[j] = (tid < C && j < Cse) ? w2t[(int64_t)j * C + tid] : 0.f;
    }
    const int q = tid % Q, ph = tid / Q;
    if (ph < P) {
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
        int sl = ph;
        for (; sl + P < S; sl += 2 * P) {
            const f32x4 v0 = *reinterpret_cast<const f32x4 *>(pb + (int64_t)sl * C + q * 4);
            const f32x4 v1 = *reinterpret_cast<const f32x4 *>(pb + (int64_t)(sl + P) * C + q * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { a0[j] += v0[j]; a1[j] += v1[j]; }
        }
        if (sl < S) {
            const f32x4 v0 = *reinterpret_cast<const f32x4 *>(pb + (int64_t)sl * C + q * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) a0[j] += v0[j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) a0[j] += a1[j];
        red[ph * Q + q] = a0;
    }
    __syncthreads();
    if (tid < Q) {
        f32x4 t = red[tid];
        for (int k = 1; k < P; ++k) {
            const f32x4 o = red[k * Q + tid];
#pragma unroll
            for (int j = 0; j < 4; ++j) t[j] += o[j];
        }
        const float inv = 1.0f / (float)HW;
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] *= inv;
        *reinterpret_cast<f32x4 *>(mean + tid * 4) = t;
        *reinterpret_cast<f32x4 *>(pb + (int64_t)S * C + tid * 4) = t;      // slice S keeps the mean, as before
    }
    __syncthreads();
    const int wave = tid >> 6, lane = tid & 63;
    // reduce conv: a wave per output channel, two output channels and four 64-channel strides in flight (eight
    // independent load / fma streams; each channel's own summation order is fixed)
    for (int o0 = wave; o0 < Cse; o0 += 32) {
        const int o1 = o0 + 16 < Cse ? o0 + 16 : o0;   // surplus slot redoes o0, result discarded
        const float *wr0 = w1 + (int64_t)o0 * C, *wr1 = w1 + (int64_t)o1 * C;
        float a[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        int c = lane;
        if (o0 == wave) {                              // first pair of rows: the two prefetched groups of four strides
            if (c + 192 < C) {
                const float m0 = mean[c], m1 = mean[c + 64], m2 = mean[c + 128], m3 = mean[c + 192];
                a[0][0] = fmaf(p0[0], m0, a[0][0]); a[1][0] = fmaf(p1[0], m0, a[1][0]);
                a[0][1] = fmaf(p0[1], m1, a[0][1]); a[1][1] = fmaf(p1[1], m1, a[1][1]);
                a[0][2] = fmaf(p0[2], m2, a[0][2]); a[1][2] = fmaf(p1[2], m2, a[1][2]);
                a[0][3] = fmaf(p0[3], m3, a[0][3]); a[1][3] = fmaf(p1[3], m3, a[1][3]);
                c += 256;
                if (c + 192 < C) {
                    const float n0 = mean[c], n1 = mean[c + 64], n2 = mean[c + 128], n3 = mean[c + 192];
                    a[0][0] = fmaf(p0[4], n0, a[0][0]); a[1][0] = fmaf(p1[4], n0, a[1][0]);
                    a[0][1] = fmaf(p0[5], n1, a[0][1]); a[1][1] = fmaf(p1[5], n1, a[1][1]);
                    a[0][2] = fmaf(p0[6], n2, a[0][2]); a[1][2] = fmaf(p1[6], n2, a[1][2]);
                    a[0][3] = fmaf(p0[7], n3, a[0][3]); a[1][3] = fmaf(p1[7], n3, a[1][3]);
                    c += 256;
                }
            }
        }
        for (; c + 192 < C; c += 256) {
            const float m0 = mean[c], m1 = mean[c + 64], m2 = mean[c + 128], m3 = mean[c + 192];
            a[0][0] = fmaf(wr0[c], m0, a[0][0]); a[1][0] = fmaf(wr1[c], m0, a[1][0]);
            a[0][1] = fmaf(wr0[c + 64], m1, a[0][1]); a[1][1] = fmaf(wr1[c + 64], m1, a[1][1]);
            a[0][2] = fmaf(wr0[c + 128], m2, a[0][2]); a[1][2] = fmaf(wr1[c + 128], m2, a[1][2]);
            a[0][3] = fmaf(wr0[c + 192], m3, a[0][3]); a[1][3] = fmaf(wr1[c + 192], m3, a[1][3]);
        }
        for (; c < C; c += 64) {
            a[0][0] = fmaf(wr0[c], mean[c], a[0][0]);
            a[1][0] = fmaf(wr1[c], mean[c], a[1][0]);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float acc = (a[u][0] + a[u][1]) + (a[u][2] + a[u][3]);
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
            const int o = o0 + 16 * u;
            if (lane == 0 && o < Cse) {
                acc += b1[o];
                hid[o] = acc * mydet_sigmoid(acc);
            }
        }
    }
    __syncthreads();
    // expand conv + sigmoid: a thread per channel, k in order over four interleaved chains, eight loads in flight
    for (int c = tid; c < C; c += 1024) {
        float e[4] = {0.f, 0.f, 0.f, 0.f};
        int k = 0;
        if (c == tid && Cse >= 8) {                    // the prefetched first eight rows
#pragma unroll
            for (int j = 0; j < 8; ++j) e[j & 3] = fmaf(q2[j], hid[j], e[j & 3]);
            k = 8;
        }
        for (; k + 7 < Cse; k += 8) {
            float w[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) w[j] = w2t[(int64_t)(k + j) * C + c];
#pragma unroll
            for (int j = 0; j < 8; ++j) e[j & 3] = fmaf(w[j], hid[k + j], e[j & 3]);
        }
        for (; k < Cse; ++k) e[k & 3] = fmaf(w2t[(int64_t)k * C + c], hid[k], e[k & 3]);
        gate[(int64_t)b * C + c] = mydet_sigmoid(((e[0] + e[1]) + (e[2] + e[3])) + b2[c]);
    }
}

// ----------------------------------------------------------------------------------------------- max pool
__global__ __launch_bounds__(256) void maxpool3s2_kernel(const float *x, int64_t ldx, float *y, int64_t ldy, int C,
                                                         int H, int W, int Ho, int Wo, int64_t total) {
    const int Q = C >> 2;
    for (int64_t it = (int64_t)blockIdx.x * 256 + threadIdx.x; it < total; it += (int64_t)gridDim.x * 256) {
        const int q = (int)(it % Q);
        const int64_t pix = it / Q;
        const int ow = (int)(pix % Wo);
        const int64_t t = pix / Wo;
        const int oh = (int)(t % Ho);
        const int64_t b = t / Ho;
        const float ninf = -__builtin_inff();
        f32x4 m = {ninf, ninf, ninf, ninf};
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int ih = oh * 2 - 1 + kh;
            if ((unsigned)ih >= (unsigned)H) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int iw = ow * 2 - 1 + kw;
                if ((unsigned)iw >= (unsigned)W) continue;
                const f32x4 v = *reinterpret_cast<const f32x4 *>(x + ((b * H + ih) * W + iw) * ldx + q * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) m[j] = fmaxf(m[j], v[j]);
            }
        }
        *reinterpret_cast<f32x4 *>(y + pix * ldy + q * 4) = m;
    }
}

// ------------------------------------------------------------------------------------------- BiFPN fusion
struct FuseArgs {
    const float *in[3];
    int64_t ld[3];
    int mode[3];              // 0 same size, 1 nearest 2x upsample of a half-size map, 2 3x3/2 max pool of a double-size map
    int n;
    const float *weights;     // raw (pre-relu) fusion weights [n]
    float *y;
    int64_t ldy, total;
    int C, H, W;
};

__device__ __forceinline__ f32x4 fuse_read(const FuseArgs &p, int i, int64_t b, int oh, int ow, int q) {
    const float *x = p.in[i];
    const int64_t ld = p.ld[i];
    if (p.mode[i] == 0) return *reinterpret_cast<const f32x4 *>(x + ((b * p.H + oh) * p.W + ow) * ld + q * 4);
    if (p.mode[i] == 1) {
        const int Hs = p.H >> 1, Ws = p.W >> 1;
        return *reinterpret_cast<const f32x4 *>(x + ((b * Hs + (oh >> 1)) * Ws + (ow >> 1)) * ld + q * 4);
    }
    const int Hb = p.H * 2, Wb = p.W * 2;
    const float ninf = -__builtin_inff();
    f32x4 m = {ninf, ninf, ninf, ninf};
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int ih = oh * 2 - 1 + kh;
        if ((unsigned)ih >= (unsigned)Hb) continue;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int iw = ow * 2 - 1 + kw;
            if ((unsigned)iw >= (unsigned)Wb) continue;
            const f32x4 v = *reinterpret_cast<const f32x4 *>(x + ((b * Hb + ih) * Wb + iw) * ld + q * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) m[j] = fmaxf(m[j], v[j]);
        }
    }
    return m;
}

__global__ __launch_bounds__(256) void bifpn_fuse_kernel(const FuseArgs p) {
    // w = relu(weights); w = w / (sum(w) + 0.0001)      (models/fpns.py:435-436)
    float w0 = fmaxf(p.weights[0], 0.0f), w1 = fmaxf(p.weights[1], 0.0f);
    float w2 = p.n > 2 ? fmaxf(p.weights[2], 0.0f) : 0.0f;
    float sum = w0 + w1;
    if (p.n > 2) sum += w2;
    sum += 0.0001f;
    w0 = w0 / sum; w1 = w1 / sum; w2 = w2 / sum;
    const int Q = p.C >> 2;
    for (int64_t it = (int64_t)blockIdx.x * 256 + threadIdx.x; it < p.total; it += (int64_t)gridDim.x * 256) {
        const int q = (int)(it % Q);
        const int64_t pix = it / Q;
        const int ow = (int)(pix % p.W);
        const int64_t t = pix / p.W;
        const int oh = (int)(t % p.H);
        const int64_t b = t / p.H;
        f32x4 acc;                                     // python sum(): 0 + w0*x0 + w1*x1 (+ w2*x2)
        const f32x4 v0 = fuse_read(p, 0, b, oh, ow, q), v1 = fuse_read(p, 1, b, oh, ow, q);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = w0 * v0[j] + w1 * v1[j];
        if (p.n > 2) {
            const f32x4 v2 = fuse_read(p, 2, b, oh, ow, q);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = acc[j] + w2 * v2[j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = acc[j] * mydet_sigmoid_fast(acc[j]);
        *reinterpret_cast<f32x4 *>(p.y + pix * p.ldy + q * 4) = acc;
    }
}

inline unsigned grid_for(int64_t total) {
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    return (unsigned)(blocks < 1 ? 1 : blocks);
}

inline bool al16(const void *p) { return ((uintptr_t)p & 15) == 0; }

inline bool dw_tiled() {      // MYDET_DW_TILED=0: the register-blocked kernels for stride 1 too (tuning / A-B)
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("MYDET_DW_TILED");
        v = e ? (atoi(e) != 0) : 1;
    }
    return v != 0;
}

inline bool block2() {       // MYDET_DW_BLOCK2=0: single-row strips everywhere (tuning / A-B)
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("MYDET_DW_BLOCK2");
        v = e ? (atoi(e) != 0) : 1;
    }
    return v != 0;
}

}  // namespace

template <int K, int ST>
int launch_dw(const DwArgs &p0, int B, hipStream_t stream) {
    DwArgs p = p0;
    const bool strip4 = (p.Wo % 4) == 0;
    const int TW = strip4 ? 4 : 1;
    if (p.partial || p.se.gate) {
        const dim3 grid(p.S, B);
        if (strip4 && ST == 1 && (p.Ho & 1) == 0 && block2())
            hipLaunchKernelGGL((dwconv_sum_kernel<K, 1, 4, 2>), grid, dim3(256), 0, stream, p);
        else if (strip4) hipLaunchKernelGGL((dwconv_sum_kernel<K, ST, 4>), grid, dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((dwconv_sum_kernel<K, ST, 1>), grid, dim3(256), 0, stream, p);
    } else {
        const bool two = strip4 && ST == 1 && (p.Ho & 1) == 0 && block2();
        p.total = (int64_t)B * (p.Ho / (two ? 2 : 1)) * (p.Wo / TW) * (p.C >> 2);
        if (two) hipLaunchKernelGGL((dwconv_kernel<K, 1, 4, 2>), dim3(grid_for(p.total)), dim3(256), 0, stream, p);
        else if (strip4) hipLaunchKernelGGL((dwconv_kernel<K, ST, 4>), dim3(grid_for(p.total)), dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((dwconv_kernel<K, ST, 1>), dim3(grid_for(p.total)), dim3(256), 0, stream, p);
    }
    return mydet_launch_status();
}

// number of squeeze slices mydet_dwconv_f32 writes for this layer (the caller sizes se_partial with it)
extern "C" int mydet_dwconv_slices(int Ho, int Wo, int C, int K, int stride) {
    if (Ho <= 0 || Wo <= 0) return 0;
    if (stride == 1 && (K == 3 || K == 5) && dw_tiled() && C >= 16)      // LDS-tiled kernel: one slice per tile (8 x 16, or 8 x 32 under 32 channels)
        return ((Ho + 7) / 8) * (C >= 32 ? (Wo + 15) / 16 : (Wo + 31) / 32);
    int s = Ho * Wo / 16;
    s = s > 128 ? 128 : s;
    return s < 1 ? 1 : s;
}

// workgroups per image of the squeeze-emitting kernel mydet_dwconv_f32 picks (sizes the hpart scratch of an in-launch SE tail)
extern "C" int mydet_dwconv_se_groups(int Ho, int Wo, int C, int K, int stride) {
    const int S = mydet_dwconv_slices(Ho, Wo, C, K, stride);
    if (S <= 0 || C <= 0 || (C & 3)) return 0;
    if (stride == 1 && (K == 3 || K == 5) && dw_tiled() && C >= 16)
        return S * (C < 32 ? ((C >> 2) + 3) / 4 : ((C >> 2) + 7) / 8);
    return S;
}

template <int K>
static int launch_dw_tile(const DwArgs &a, int B, hipStream_t stream) {
    DwTArgs p;
    p.se = a.se;
    p.x = a.x; p.w = a.w; p.scale = a.scale; p.shift = a.shift; p.y = a.y; p.partial = a.partial; p.ldx = a.ldx; p.ldy = a.ldy;
    p.C = a.C; p.H = a.H; p.W = a.W; p.Ho = a.Ho; p.Wo = a.Wo; p.pad_t = a.pad_t; p.pad_l = a.pad_l; p.act = a.act;
    const bool narrow = a.C < 32;                            // 4 quads x (8 x 32) outputs instead of 8 quads x (8 x 16)
    p.tiles_x = narrow ? (a.Wo + 31) / 32 : (a.Wo + 15) / 16;
    p.S = p.tiles_x * ((a.Ho + 7) / 8);
    p.nchunks = narrow ? ((a.C >> 2) + 3) / 4 : ((a.C >> 2) + 7) / 8;
    const int64_t grid = (int64_t)B * p.S * p.nchunks;      // one workgroup per (image, tile, chunk): a persistent form with a
    if (grid > 0x7fffffff) return MYDET_E_UNSUPP;           // register prefetch of the next tile measured 20 % slower (profiles/r03_mbconv_notes.md)
    const bool sums = a.partial || a.se.gate;
    if (narrow) {
        if (sums) hipLaunchKernelGGL((dwconv_tile_kernel<K, true, 4>), dim3((unsigned)grid), dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((dwconv_tile_kernel<K, false, 4>), dim3((unsigned)grid), dim3(256), 0, stream, p);
    } else {
        if (sums) hipLaunchKernelGGL((dwconv_tile_kernel<K, true>), dim3((unsigned)grid), dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((dwconv_tile_kernel<K, false>), dim3((unsigned)grid), dim3(256), 0, stream, p);
    }
    return mydet_launch_status();
}

extern "C" int mydet_dwconv_f32(const float *x, int64_t ldx, const float *w, const float *scale, const float *shift,
                                float *y, int64_t ldy, int B, int H, int W, int C, int K, int stride, int pad_t,
                                int pad_l, int Ho, int Wo, int act, float *se_partial, int S, const mydet_se_tail *se,
                                void *stream) {
    if (!x || !w || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0 || Ho <= 0 || Wo <= 0) return MYDET_E_BADARG;
    if ((C & 3) || (ldx & 3) || (ldy & 3) || ldx < C || ldy < C || !al16(x) || !al16(w) || !al16(y)) return MYDET_E_BADARG;
    if ((scale == nullptr) != (shift == nullptr) || (scale && (!al16(scale) || !al16(shift)))) return MYDET_E_BADARG;
    if (se_partial && (S <= 0 || S > 4096 || B > 65535 || !al16(se_partial))) return MYDET_E_BADARG;
    if ((K != 3 && K != 5) || (stride != 1 && stride != 2)) return MYDET_E_UNSUPP;
    DwArgs p;
    p.x = x; p.w = w; p.scale = scale; p.shift = shift; p.y = y; p.partial = se_partial; p.ldx = ldx; p.ldy = ldy;
    p.C = C; p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo; p.pad_t = pad_t; p.pad_l = pad_l; p.act = act; p.S = S; p.total = 0;
    p.se = se ? *se : NO_SE_TAIL;
    if (const int e = mydet_se_tail_check(p.se, C, B, p.se.gate ? mydet_dwconv_se_groups(Ho, Wo, C, K, stride) : 1)) return e;
    if (p.se.gate && (S <= 0 || S > 4096 || B > 65535)) return MYDET_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    if ((se_partial || p.se.gate) && S != mydet_dwconv_slices(Ho, Wo, C, K, stride)) return MYDET_E_BADARG;
    if (stride == 1 && dw_tiled() && C >= 16) return K == 3 ? launch_dw_tile<3>(p, B, st) : launch_dw_tile<5>(p, B, st);
    if (K == 3) return stride == 1 ? launch_dw<3, 1>(p, B, st) : launch_dw<3, 2>(p, B, st);
    return stride == 1 ? launch_dw<5, 1>(p, B, st) : launch_dw<5, 2>(p, B, st);
}

extern "C" int mydet_channel_sums_f32(const float *x, int64_t ldx, int B, int H, int W, int C, float *partial, int S,
                                      void *stream) {
    if (!x || !partial || B <= 0 || H <= 0 || W <= 0 || C <= 0 || S <= 0 || S > 4096 || B > 65535) return MYDET_E_BADARG;
    if ((C & 3) || (ldx & 3) || ldx < C || !al16(x) || !al16(partial)) return MYDET_E_BADARG;
    hipLaunchKernelGGL(squeeze_partial_kernel, dim3(S, B), dim3(256), 0, (hipStream_t)stream, x, ldx, C, H * W, S,
                       partial);
    return mydet_launch_status();
}

extern "C" int mydet_se_gate_f32(float *partial, int S, int B, int HW, int C, const float *w1, const float *b1,
                                 int Cse, const float *w2t, const float *b2, float *gate, void *stream) {
    if (!partial || !w1 || !b1 || !w2t || !b2 || !gate || B <= 0 || B > 65535 || HW <= 0 || C <= 0 || Cse <= 0 || S <= 0)
        return MYDET_E_BADARG;
    if ((C & 3) == 0 && (C >> 2) <= 1024 && C >= 4) {      // one launch: a workgroup per image
        int P = 1024 / (C >> 2);
        P = P > 32 ? 32 : P;
        P = P > S ? S : P;
        const size_t l1 = ((size_t)C + ((Cse + 3) & ~3) + (size_t)P * C) * sizeof(float);
        if (l1 <= 64 * 1024) {
            hipLaunchKernelGGL(se_fused_kernel, dim3(B), dim3(1024), l1, (hipStream_t)stream, partial, S, C, HW, w1, b1, Cse,
                               w2t, b2, gate, P);
            return mydet_launch_status();
        }
    }
    const size_t lds = (size_t)(C + Cse) * sizeof(float);
    if (lds > 64 * 1024) return MYDET_E_UNSUPP;
    hipLaunchKernelGGL(se_mean_kernel, dim3((C + 63) / 64, B), dim3(256), 0, (hipStream_t)stream, partial, S, C, HW);
    const int rc = mydet_launch_status();
    if (rc) return rc;
    hipLaunchKernelGGL(se_gate_kernel, dim3((C + 63) / 64, B), dim3(256), lds, (hipStream_t)stream, partial, S, C, w1,
                       b1, Cse, w2t, b2, gate);
    return mydet_launch_status();
}

extern "C" int mydet_maxpool3s2_f32(const float *x, int64_t ldx, float *y, int64_t ldy, int B, int H, int W, int C,
                                    int Ho, int Wo, void *stream) {
    if (!x || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0) return MYDET_E_BADARG;
    if ((C & 3) || (ldx & 3) || (ldy & 3) || ldx < C || ldy < C || !al16(x) || !al16(y)) return MYDET_E_BADARG;
    if (Ho != (H + 2 - 3) / 2 + 1 || Wo != (W + 2 - 3) / 2 + 1) return MYDET_E_BADARG;
    const int64_t total = (int64_t)B * Ho * Wo * (C >> 2);
    hipLaunchKernelGGL(maxpool3s2_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, C, H,
                       W, Ho, Wo, total);
    return mydet_launch_status();
}

extern "C" int mydet_bifpn_fuse_f32(int n, const float *in0, int64_t ld0, int mode0, const float *in1, int64_t ld1,
                                    int mode1, const float *in2, int64_t ld2, int mode2, const float *weights, float *y,
                                    int64_t ldy, int B, int H, int W, int C, void *stream) {
    if (n < 2 || n > 3 || !in0 || !in1 || (n == 3 && !in2) || !weights || !y) return MYDET_E_BADARG;
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || (ldy & 3) || ldy < C || !al16(y)) return MYDET_E_BADARG;
    FuseArgs p;
    const float *ins[3] = {in0, in1, in2};
    const int64_t lds[3] = {ld0, ld1, ld2};
    const int modes[3] = {mode0, mode1, mode2};
    for (int i = 0; i < 3; ++i) {
        p.in[i] = ins[i]; p.ld[i] = lds[i]; p.mode[i] = modes[i];
        if (i < n) {
            if (modes[i] < 0 || modes[i] > 2 || (lds[i] & 3) || lds[i] < C || !al16(ins[i])) return MYDET_E_BADARG;
            if (modes[i] == 1 && ((H & 1) || (W & 1))) return MYDET_E_BADARG;
        }
    }
    p.n = n; p.weights = weights; p.y = y; p.ldy = ldy; p.C = C; p.H = H; p.W = W;
    p.total = (int64_t)B * H * W * (C >> 2);
    hipLaunchKernelGGL(bifpn_fuse_kernel, dim3(grid_for(p.total)), dim3(256), 0, (hipStream_t)stream, p);
    return mydet_launch_status();
}
