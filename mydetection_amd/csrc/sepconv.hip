// Fused separable-conv node of the BiFPN / EfDetHead pyramid (88-channel maps):
//
//     y = act( BN( pointwise1x1( depthwise3x3( pre(x...) ) ) ) )
//
//   pre = identity                                   head towers: spconv3x3_bn_swish, models/rpns.py:199-205
//   pre = swish( sum_i w_i * in_i ), w = relu/(sum+1e-4)   LinearFusion, models/fpns.py:421-439 (an input may be read
//                                                    through nearest-2x upsampling or the 3x3/2 max pool, :398-418)
//   depthwise 3x3 pad 1 no bias -> pointwise 1x1 + bias: SeparableConv2d, models/modules.py:5-21
//
// replaces three launches (fusion, depthwise, pointwise GEMM) and two round trips of the 88-channel map through HBM
// per node.  A launch covers up to 10 nodes (all pyramid levels x both head towers) through a tile table, so the
// 5x5 / 10x10 levels do not pay a launch each.
//
// One 256-thread workgroup owns an 8x8 output tile of one image, all channels:
//   0. requests that do not depend on the halo go out first (depthwise taps -> LDS, the first weight block -> registers)
//   1. the 10x10 halo of pre(x) -> LDS (float4 per thread, inputs read once; out-of-image pixels are the conv's zeros);
//      every load is unconditional at clamped coordinates and specialised by the node's input modes, so that a batch of
//      halo entries is in flight at once (a run-time mode switch around each load made hipcc wait for every single one)
//   2. depthwise 3x3 on the VALU from LDS (same fmaf order as dwconv_kernel), results kept in registers, then stored
//      over the halo as the [64 px][C+1] operand tile (odd stride: conflict-free fragment reads)
//   3. pointwise conv on FP32 MFMA, computed transposed (A = weights, B = pixels: v_mfma_f32_16x16x4_f32), so a lane
//      ends up with 4 consecutive output channels of one pixel = one 16-byte store after the BN/act epilogue.  A wave
//      owns 16 pixels; the weights come pre-packed in operand order, one 16-channel block at a time through a two-slot
//      LDS ring shared by the workgroup's four waves (SpW below).
// LDS 38 KB -> 4 workgroups per CU: the halo loads of one overlap the MFMAs of another.
// Built with -ffp-contract=off: the fusion arithmetic rounds like the reference's separate mul / add ops.
#include <cstdlib>

#include "common.h"

namespace {

constexpr int SP_MAX = MYDET_SEPCONV_MAX_NODES;
constexpr int TS = 8, HS = TS + 2;

struct SpNode {
    const float *in[3];
    int64_t ld[3];
    int mode[3];
    int n_in;
    const float *fuse_w, *wd, *wpk, *scale, *shift;
    float *y;
    int64_t ldy;
    int H, W, Cout, act, nb, nsplit, nb_per;
    int tiles_x, tiles_per_img, tile_begin;
};
struct SpArgs {
    int n, B, total;
    SpNode p[SP_MAX];
};

__device__ __forceinline__ f32x4 sp_read(const SpNode &P, int i, int64_t b, int oh, int ow, int q) {
    const float *x = P.in[i];
    const int64_t ld = P.ld[i];
    if (P.mode[i] == 0) return *reinterpret_cast<const f32x4 *>(x + ((b * P.H + oh) * P.W + ow) * ld + q * 4);
    if (P.mode[i] == 1) {
        const int Hs = P.H >> 1, Ws = P.W >> 1;
        return *reinterpret_cast<const f32x4 *>(x + ((b * Hs + (oh >> 1)) * Ws + (ow >> 1)) * ld + q * 4);
    }
    // 3x3/2 max pool with -inf padding: all nine taps are loaded unconditionally at clamped coordinates (nine loads
    // in flight instead of nine branch-and-wait rounds); taps outside the map do not enter the max
    const int Hb = P.H * 2, Wb = P.W * 2;
    const float ninf = -__builtin_inff();
    f32x4 tap[9];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int ih = min(max(oh * 2 - 1 + kh, 0), Hb - 1), iw = min(max(ow * 2 - 1 + kw, 0), Wb - 1);
            tap[kh * 3 + kw] = *reinterpret_cast<const f32x4 *>(x + ((b * Hb + ih) * Wb + iw) * ld + q * 4);
        }
    f32x4 m = {ninf, ninf, ninf, ninf};
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const bool in = (unsigned)(oh * 2 - 1 + kh) < (unsigned)Hb && (unsigned)(ow * 2 - 1 + kw) < (unsigned)Wb;
#pragma unroll
            for (int j = 0; j < 4; ++j) m[j] = in ? fmaxf(m[j], tap[kh * 3 + kw][j]) : m[j];
        }
    return m;
}

struct SpItem {
    int pi, b, oy0, ox0, nb_begin, nb_end;
};

// one fused input at in-image (clamped) coordinates, branch-free: MODE 0 same size, 1 nearest-2x of the half-size map,
// 2 3x3/2 max pool of the double-size map (-inf padding: taps outside the map do not enter the max)
template <int MODE>
__device__ __forceinline__ f32x4 sp_read_m(const float *x, int64_t ld, int H, int W, int64_t b, int oh, int ow, int q) {
    if (MODE == 0) return *reinterpret_cast<const f32x4 *>(x + ((b * H + oh) * W + ow) * ld + q * 4);
    if (MODE == 1) {
        const int Hs = H >> 1, Ws = W >> 1;
        return *reinterpret_cast<const f32x4 *>(x + ((b * Hs + (oh >> 1)) * Ws + (ow >> 1)) * ld + q * 4);
    }
    const int Hb = H * 2, Wb = W * 2;
    const float ninf = -__builtin_inff();
    f32x4 tap[9];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int ih = min(max(oh * 2 - 1 + kh, 0), Hb - 1), iw = min(max(ow * 2 - 1 + kw, 0), Wb - 1);
            tap[kh * 3 + kw] = *reinterpret_cast<const f32x4 *>(x + ((b * Hb + ih) * Wb + iw) * ld + q * 4);
        }
    f32x4 m = {ninf, ninf, ninf, ninf};
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const bool in = (unsigned)(oh * 2 - 1 + kh) < (unsigned)Hb && (unsigned)(ow * 2 - 1 + kw) < (unsigned)Wb;
#pragma unroll
            for (int j = 0; j < 4; ++j) m[j] = in ? fmaxf(m[j], tap[kh * 3 + kw][j]) : m[j];
        }
    return m;
}

// halo of pre(x) for one item -> LDS, HB halo entries per thread in flight.  M1 < 0: single input (no fusion);
// M2 < 0: two inputs.  Every load is unconditional at clamped coordinates (a branch per load would serialise them);
// entries outside the image become the depthwise conv's zeros by a select.
template <int KS, int HB, int M0, int M1, int M2>
__device__ __forceinline__ void sp_stage_halo(const SpNode &P, float *halo, int ptid, int64_t b, int oy0, int ox0, float w0,
                                              float w1, float w2) {
    constexpr int Q = KS;
    constexpr int NH = (HS * HS * Q + 255) / 256;
    const int H = P.H, W = P.W;
#pragma unroll
    for (int j0 = 0; j0 < NH; j0 += HB) {
        f32x4 r0[HB], r1[HB], r2[HB];
        bool inside[HB];
#pragma unroll
        for (int u = 0; u < HB; ++u) {
            const int itc = min(ptid + (j0 + u) * 256, HS * HS * Q - 1);
            const int hp = itc / Q, q = itc - hp * Q;
            const int hy = hp / HS, hx = hp - hy * HS;
            const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
            inside[u] = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            const int cy = min(max(iy, 0), H - 1), cx = min(max(ix, 0), W - 1);
            r0[u] = sp_read_m<M0>(P.in[0], P.ld[0], H, W, b, cy, cx, q);
            if (M1 >= 0) r1[u] = sp_read_m<(M1 < 0 ? 0 : M1)>(P.in[1], P.ld[1], H, W, b, cy, cx, q);
            if (M2 >= 0) r2[u] = sp_read_m<(M2 < 0 ? 0 : M2)>(P.in[2], P.ld[2], H, W, b, cy, cx, q);
        }
#pragma unroll
        for (int u = 0; u < HB; ++u) {
            f32x4 v = r0[u];
            if (M1 >= 0) {                                    // python sum(): 0 + w0*x0 + w1*x1 (+ w2*x2), then swish
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = w0 * r0[u][e] + w1 * r1[u][e];
                if (M2 >= 0) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] + w2 * r2[u][e];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] * mydet_sigmoid_fast(v[e]);
            }
            if (!inside[u]) v = f32x4{0.f, 0.f, 0.f, 0.f};
            const int it = ptid + (j0 + u) * 256;
            if (j0 + u < NH && it < HS * HS * Q) *reinterpret_cast<f32x4 *>(&halo[it * 4]) = v;     // it * 4 == hp * C + q * 4
        }
    }
}

// generic (any mode combination): the branchy reader of the single-phase kernel; not used by the BiFPN / head shapes
template <int KS>
__device__ __forceinline__ void sp_stage_halo_any(const SpNode &P, float *halo, int ptid, int64_t b, int oy0, int ox0, float w0,
                                                  float w1, float w2) {
    constexpr int Q = KS;
    const int H = P.H, W = P.W;
    for (int it = ptid; it < HS * HS * Q; it += 256) {
        const int hp = it / Q, q = it - hp * Q;
        const int hy = hp / HS, hx = hp - hy * HS;
        const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
            const f32x4 v0 = sp_read(P, 0, b, iy, ix, q), v1 = sp_read(P, 1, b, iy, ix, q);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = w0 * v0[e] + w1 * v1[e];
            if (P.n_in > 2) {
                const f32x4 v2 = sp_read(P, 2, b, iy, ix, q);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] + w2 * v2[e];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] * mydet_sigmoid_fast(v[e]);
        }
        *reinterpret_cast<f32x4 *>(&halo[it * 4]) = v;
    }
}

// BUDGET: VGPRs of the calling kernel (256: everything of an item in flight; 128: fewer entries per batch)
template <int KS, int BUDGET = 256>
__device__ __forceinline__ void sp_stage_item(const SpNode &P, const SpItem &it, float *halo, int ptid) {
    constexpr int HB1 = BUDGET >= 256 ? 9 : 5, HB2 = BUDGET >= 256 ? 3 : 2;
    if (P.n_in == 1) {
        sp_stage_halo<KS, 9, 0, -1, -1>(P, halo, ptid, it.b, it.oy0, it.ox0, 0.f, 0.f, 0.f);
        return;
    }
    float w0 = fmaxf(P.fuse_w[0], 0.0f), w1 = fmaxf(P.fuse_w[1], 0.0f);       // w = relu(weights); w = w / (sum(w) + 0.0001)
    float w2 = P.n_in > 2 ? fmaxf(P.fuse_w[2], 0.0f) : 0.0f;
    float sum = w0 + w1;
    if (P.n_in > 2) sum += w2;
    sum += 0.0001f;
    w0 = w0 / sum; w1 = w1 / sum; w2 = w2 / sum;
    const int kind = P.n_in == 2 ? P.mode[0] * 3 + P.mode[1] : 9 + (P.mode[0] * 3 + P.mode[1]) * 3 + P.mode[2];
    if (kind == 1) sp_stage_halo<KS, HB1, 0, 1, -1>(P, halo, ptid, it.b, it.oy0, it.ox0, w0, w1, w2);         // same + up2x (top-down)
    else if (kind == 2) sp_stage_halo<KS, HB2, 0, 2, -1>(P, halo, ptid, it.b, it.oy0, it.ox0, w0, w1, w2);    // same + pool (coarsest out)
    else if (kind == 9 + 2) sp_stage_halo<KS, HB2, 0, 0, 2>(P, halo, ptid, it.b, it.oy0, it.ox0, w0, w1, w2); // same + same + pool (bottom-up)
    else sp_stage_halo_any<KS>(P, halo, ptid, it.b, it.oy0, it.ox0, w0, w1, w2);
}


// The pointwise weights of one 16-channel block, KQ * 64 float4 in MFMA operand order (include/mydet.h).  The four waves
// of a workgroup need the SAME fragments, so the block is fetched once per workgroup -- 1.5 coalesced 16-byte loads per
// thread -- into a two-slot LDS ring behind the operand tile and read from there (one ds_read_b128 per four MFMAs).
// (Each wave loading its own fragments, 22 dword loads per block and wave, kept the vector-memory pipe busier than the
// matrix pipe: with those loads removed the 720-channel class node ran in 0.20 instead of 0.36 ms.)
template <int KS>
struct SpW {
    static constexpr int KQ = (KS + 3) / 4;
    static constexpr int F4 = KQ * 64;               // float4 per block
    static constexpr int SECOND = F4 - 256;          // threads that carry a second float4
    f32x4 g0, g1;
    __device__ __forceinline__ void load(const SpNode &P, int nb, int tid) {
        const f32x4 *src = reinterpret_cast<const f32x4 *>(P.wpk) + (int64_t)min(nb, P.nb - 1) * F4;
        g0 = src[tid];
        g1 = src[min(tid + 256, F4 - 1)];
    }
    __device__ __forceinline__ void store(float *slot, int tid) const {
        reinterpret_cast<f32x4 *>(slot)[tid] = g0;
        if (tid < SECOND) reinterpret_cast<f32x4 *>(slot)[tid + 256] = g1;
    }
    static __device__ __forceinline__ void fragments(const float *slot, int lane, float (&af)[KQ * 4]) {
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
            const f32x4 v = reinterpret_cast<const f32x4 *>(slot)[q * 64 + lane];
#pragma unroll
            for (int e = 0; e < 4; ++e) af[q * 4 + e] = v[e];
        }
    }
};

// Phases 0-2 of a work item, shared by the node kernel and the decoding node kernel: on return the 64 x C operand tile of
// the pointwise conv lies in `lds` ([64 px][C + 1]) and `w0` holds (in registers) the weights of channel block `nb_first`.
template <int KS>
__device__ __forceinline__ void sp_front(const SpNode &P, float *lds, float *swd, int tid, int b, int oy0, int ox0,
                                         int nb_first, SpW<KS> &w0) {
    constexpr int C = KS * 4, Q = KS, XS = C + 1;
    // 0. everything that does not depend on the halo is requested first: the depthwise taps (-> LDS), the first block's
    //    weights (registers); their L2 round trip overlaps the halo's
    w0.load(P, nb_first, tid);
    if (tid < 9 * Q) *reinterpret_cast<f32x4 *>(&swd[tid * 4]) = *reinterpret_cast<const f32x4 *>(P.wd + tid * 4);
    // 1. halo of pre(x): every load unconditional at clamped coordinates, several halo entries per thread in flight
    {
        SpItem it;
        it.pi = 0; it.b = b; it.oy0 = oy0; it.ox0 = ox0; it.nb_begin = 0; it.nb_end = 0;
        sp_stage_item<KS, 128>(P, it, lds, tid);
    }
    __syncthreads();

    // 2. depthwise 3x3 (taps in (kh, kw) order, one fmaf chain per channel).  A thread keeps one channel quad (its
    //    nine weight vectors stay in registers) and walks pixels g, g + G, ...
    constexpr int G = 256 / Q, NI = (TS * TS + G - 1) / G;
    const int dq = tid % Q, dg = tid / Q;
    f32x4 dwv[NI];
    if (dg < G) {
        f32x4 wv[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) wv[k] = *reinterpret_cast<const f32x4 *>(&swd[k * C + dq * 4]);
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int px = dg + j * G;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if (px < TS * TS) {
                const int py = px >> 3, pxx = px & 7;
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        const f32x4 v = *reinterpret_cast<const f32x4 *>(&lds[((py + kh) * HS + pxx + kw) * C + dq * 4]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[e] = fmaf(v[e], wv[kh * 3 + kw][e], acc[e]);
                    }
            }
            dwv[j] = acc;
        }
    }
    __syncthreads();
    if (dg < G) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int px = dg + j * G;
            if (px < TS * TS) {
#pragma unroll
                for (int e = 0; e < 4; ++e) lds[px * XS + dq * 4 + e] = dwv[j][e];
            }
        }
    }
    __syncthreads();
}

template <int KS>
__global__ __launch_bounds__(256, 4) void sepconv_kernel(const SpArgs a) {
    constexpr int C = KS * 4, XS = C + 1;
    constexpr int LDS_FLOATS = HS * HS * C > TS * TS * XS ? HS * HS * C : TS * TS * XS;
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS + 9 * C];
    float *swd = lds + LDS_FLOATS;                            // the node's depthwise taps [9][C]
    const int tid = threadIdx.x, bid = blockIdx.x;
    int pi = 0;
    for (int i = 1; i < a.n; ++i)
        if (bid >= a.p[i].tile_begin) pi = i;                 // uniform
    const SpNode &P = a.p[pi];
    // work item = (tile, slice of the output-channel blocks): small maps are cut along the channels as well, so a
    // 5x5 level does not leave 240 CUs idle while 16 workgroups walk all of Cout
    const int t0 = bid - P.tile_begin;
    const int t = t0 / P.nsplit, ns = t0 - t * P.nsplit;
    const int nb_begin = ns * P.nb_per, nb_end = min(P.nb, nb_begin + P.nb_per);
    const int b = t / P.tiles_per_img, r = t - b * P.tiles_per_img;
    const int ty = r / P.tiles_x, tx = r - ty * P.tiles_x;
    const int oy0 = ty * TS, ox0 = tx * TS;
    const int H = P.H, W = P.W;
    const int wave = tid >> 6, lane = tid & 63;
    SpW<KS> wreg;
    sp_front<KS>(P, lds, swd, tid, b, oy0, ox0, nb_begin, wreg);
    float *ring = lds + TS * TS * XS;                         // two weight slots behind the operand tile
    constexpr int WSLOT = SpW<KS>::F4 * 4;
    wreg.store(ring, tid);
    __syncthreads();

    // 3. pointwise conv, transposed: D[channel][pixel] = sum_k W[channel][k] * X[pixel][k]
    const int m0 = wave * 16;
    float bf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) bf[ks] = lds[(m0 + (lane & 15)) * XS + ks * 4 + (lane >> 4)];
    const int px = m0 + (lane & 15);
    const int oy = oy0 + (px >> 3), ox = ox0 + (px & 7);
    const bool valid = oy < H && ox < W;
    float *yp = P.y + (((int64_t)b * H + oy) * W + ox) * P.ldy;
    const int nsub = (lane >> 4) * 4;
    const int act = P.act;
    // one 16-channel block at a time; the NEXT block's weights are in flight (registers) under the current block's
    // MFMAs and land in the other ring slot at the end of the iteration.  Two accumulator chains over the even / odd
    // k-steps keep the matrix pipe issuing (40-cycle dependent latency).
    for (int nb = nb_begin; nb < nb_end; ++nb) {
        const int cur = (nb - nb_begin) & 1;
        wreg.load(P, nb + 1, tid);
        float af[SpW<KS>::KQ * 4];
        SpW<KS>::fragments(ring + cur * WSLOT, lane, af);
        const int n = nb * 16 + nsub;
        const int nc = min(n, P.Cout - 4);
        const f32x4 sh = *reinterpret_cast<const f32x4 *>(P.shift + nc);
        f32x4 sc = {1.f, 1.f, 1.f, 1.f};
        if (P.scale) sc = *reinterpret_cast<const f32x4 *>(P.scale + nc);
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ks += 2) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ks], bf[ks], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ks + 1], bf[ks + 1], acc1, 0, 0, 0);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) acc0[e] = acc0[e] + acc1[e];
        if (valid && n < P.Cout) {
            if (P.scale) {
#pragma unroll
                for (int e = 0; e < 4; ++e) acc0[e] = acc0[e] * sc[e] + sh[e];
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) acc0[e] = acc0[e] + sh[e];
            }
            if (act == MYDET_ACT_SWISH) {
#pragma unroll
                for (int e = 0; e < 4; ++e) acc0[e] = acc0[e] * mydet_sigmoid_fast(acc0[e]);
            }
            *reinterpret_cast<f32x4 *>(yp + n) = acc0;
        }
        wreg.store(ring + (cur ^ 1) * WSLOT, tid);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------- decoding head nodes
// The LAST layer of an EfDetHead tower with RetinaLayer's decode in its epilogue (models/rpns.py:121-197 ->
// models/detlayers/retinanet.py:63-82): the class tower's 9 x 90 logits per pixel (442 MB per 16 images at 640^2)
// are never written -- a lane holds 4 consecutive classes of one pixel, keeps the running maximum / first argmax of the
// current anchor over the anchor's channel blocks, the four lanes of a pixel are merged by lane exchange, and
// score = sigmoid(max logit), class index go straight to the candidate arrays; the box tower's 4 logits per anchor are
// exactly one lane's 4 channels and leave as a decoded box.  Same arithmetic as decode_kernel (decode.hip), same
// candidate order (a, y, x).
constexpr int DEC_MAX_A = 12;
struct SpDecNode {
    int kind;                       // 0 class tower, 1 box tower
    float stride;
    int64_t n_off;
    float aw[DEC_MAX_A], ah[DEC_MAX_A];
};
struct SpDecArgs {
    int n, B, A, n_cls, img_h, img_w;
    int64_t N;
    float *bbox, *score;
    int64_t *cidx;
    SpNode p[SP_MAX];
    SpDecNode d[SP_MAX];
};

// first maximum of sigmoid(x) over increasing k.  Below 5 the logits are compared (sigmoid is monotone; decode_kernel does
// the same); from 5 up float32 sigmoids of different logits coincide more and more (all of them from 17.4), so there
// the sigmoid values decide and an equal value keeps the earlier class -- what torch.max over the sigmoids returns.
__device__ __forceinline__ void dec_update(float &best, int &bi, float x, int k) {
    if (x > best) {
        if (x < 5.0f || mydet_sigmoid(x) > mydet_sigmoid(best)) {
            best = x;
            bi = k;
        }
    }
}

template <int KS, int NBA>
__global__ __launch_bounds__(256, 4) void sepconv_decode_kernel(const SpDecArgs a) {
    constexpr int C = KS * 4, XS = C + 1;
    constexpr int LDS_FLOATS = HS * HS * C > TS * TS * XS ? HS * HS * C : TS * TS * XS;
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS + 9 * C];
    __shared__ float s_aw[DEC_MAX_A], s_ah[DEC_MAX_A];
    float *swd = lds + LDS_FLOATS;
    const int tid = threadIdx.x, bid = blockIdx.x;
    int pi = 0;
    for (int i = 1; i < a.n; ++i)
        if (bid >= a.p[i].tile_begin) pi = i;                 // uniform
    const SpNode &P = a.p[pi];
    const int kind = a.d[pi].kind;
    const float st = a.d[pi].stride;
    const int64_t n_off = a.d[pi].n_off;
    if (tid < DEC_MAX_A) { s_aw[tid] = a.d[pi].aw[tid]; s_ah[tid] = a.d[pi].ah[tid]; }
    const int t0 = bid - P.tile_begin;
    const int t = t0 / P.nsplit, ns = t0 - t * P.nsplit;
    const int nb_begin = ns * P.nb_per, nb_end = min(P.nb, nb_begin + P.nb_per);       // class nodes: whole anchors
    const int b = t / P.tiles_per_img, r = t - b * P.tiles_per_img;
    const int ty = r / P.tiles_x, tx = r - ty * P.tiles_x;
    const int oy0 = ty * TS, ox0 = tx * TS;
    const int H = P.H, W = P.W;
    const int wave = tid >> 6, lane = tid & 63;
    SpW<KS> wreg;
    sp_front<KS>(P, lds, swd, tid, b, oy0, ox0, nb_begin, wreg);
    float *ring = lds + TS * TS * XS;
    constexpr int WSLOT = SpW<KS>::F4 * 4;
    wreg.store(ring, tid);
    __syncthreads();
    int cur = 0;

    const int m0 = wave * 16;
    float bf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) bf[ks] = lds[(m0 + (lane & 15)) * XS + ks * 4 + (lane >> 4)];
    const int px = m0 + (lane & 15);
    const int oy = oy0 + (px >> 3), ox = ox0 + (px & 7);
    const bool valid = oy < H && ox < W;
    const int kk = lane >> 4, nsub = kk * 4;
    const int64_t cand0 = (int64_t)b * a.N + n_off + (int64_t)oy * W + ox;               // + a * H * W
    const int64_t hw = (int64_t)H * W;

    // one 16-channel block: acc = W[nb] . X^T + shift (the NEXT block's weights are requested first and land in the other
    // ring slot at the end, as in sepconv_kernel)
#define SP_DEC_BLOCK(nb_, v_)                                                                         \
    {                                                                                                 \
        wreg.load(P, (nb_) + 1, tid);                                                                 \
        float af[SpW<KS>::KQ * 4];                                                                    \
        SpW<KS>::fragments(ring + cur * WSLOT, lane, af);                                             \
        /* clamped like sepconv_kernel's: a box node's shift has A*4 floats, not whole 16-channel blocks */ \
        const int nq_ = min((nb_) * 16 + nsub, P.Cout - 4);                                           \
        const f32x4 sh_ = *reinterpret_cast<const f32x4 *>(P.shift + nq_);                            \
        f32x4 sc_ = {1.f, 1.f, 1.f, 1.f};                                                             \
        if (P.scale) sc_ = *reinterpret_cast<const f32x4 *>(P.scale + nq_);                           \
        f32x4 acc0_ = {0.f, 0.f, 0.f, 0.f}, acc1_ = {0.f, 0.f, 0.f, 0.f};                             \
        _Pragma("unroll") for (int ks = 0; ks < KS; ks += 2) {                                        \
            acc0_ = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ks], bf[ks], acc0_, 0, 0, 0);             \
            acc1_ = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ks + 1], bf[ks + 1], acc1_, 0, 0, 0);     \
        }                                                                                             \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) acc0_[e] = acc0_[e] + acc1_[e];                 \
        if (P.scale) {                                                                                \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) acc0_[e] = acc0_[e] * sc_[e] + sh_[e];      \
        } else {                                                                                      \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) acc0_[e] = acc0_[e] + sh_[e];               \
        }                                                                                             \
        (v_) = acc0_;                                                                                 \
        wreg.store(ring + (cur ^ 1) * WSLOT, tid);                                                    \
        __syncthreads();                                                                              \
        cur ^= 1;                                                                                     \
    }

    if (kind == 0) {
        const float ninf = -__builtin_inff();
        for (int an = nb_begin / NBA; an < nb_end / NBA; ++an) {
            float best = ninf;
            int bi = 0;
#pragma unroll
            for (int blk = 0; blk < NBA; ++blk) {
                f32x4 v;
                SP_DEC_BLOCK(an * NBA + blk, v)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = blk * 16 + nsub + e;
                    if (k < a.n_cls) dec_update(best, bi, v[e], k);
                }
            }
            // the pixel's four lanes hold interleaved class subsets: the larger value wins, an equal one only with the
            // lower class index (= the first maximum of the sequential scan)
#pragma unroll
            for (int off = 16; off <= 32; off <<= 1) {
                const float ob = __shfl_xor(best, off);
                const int obi = __shfl_xor(bi, off);
                bool gt = ob > best, eq = ob == best;
                if (fmaxf(ob, best) >= 5.0f) {
                    const float so = mydet_sigmoid(ob), sb = mydet_sigmoid(best);
                    gt = so > sb;
                    eq = so == sb;
                }
                const bool take = gt || (eq && obi < bi);
                best = take ? ob : best;
                bi = take ? obi : bi;
            }
            if (kk == 0 && valid) {
                const int64_t n = cand0 + (int64_t)an * hw;
                a.score[n] = mydet_sigmoid(best);
                a.cidx[n] = (int64_t)bi;
            }
        }
    } else {
        const float fmaxhw = (float)(a.img_h > a.img_w ? a.img_h : a.img_w);
        for (int nb = nb_begin; nb < nb_end; ++nb) {
            f32x4 v;
            SP_DEC_BLOCK(nb, v)
            const int an = nb * 4 + kk;                         // a lane's 4 channels are one anchor's (tx, ty, tw, th)
            if (valid && an < a.A) {
                const float aw = s_aw[an], ah = s_ah[an];
                const float acx = st * 0.5f + (float)ox * st;
                const float acy = st * 0.5f + (float)oy * st;
                f32x4 o;
                o[0] = acx + v[0] * aw;
                o[1] = acy + v[1] * ah;
                o[2] = expf(v[2]) * aw;
                o[3] = expf(v[3]) * ah;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = fminf(fmaxf(o[j], 1.0f), fmaxhw);
                *reinterpret_cast<f32x4 *>(a.bbox + (cand0 + (int64_t)an * hw) * 4) = o;
            }
        }
    }
#undef SP_DEC_BLOCK
}


inline bool al16(const void *p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

extern "C" int mydet_sepconv_nodes_f32(int n, const mydet_sepconv_node *nodes, int B, int C, void *stream) {
    if (n <= 0 || n > SP_MAX || !nodes || B <= 0) return MYDET_E_BADARG;
    if (C != 88) return MYDET_E_UNSUPP;                      // instantiated for the 88-channel pyramids of the D1 family
    SpArgs a;
    a.n = n; a.B = B;
    int64_t tiles = 0;
    for (int i = 0; i < n; ++i) {
        const mydet_sepconv_node &s = nodes[i];
        SpNode &p = a.p[i];
        if (s.n_in < 1 || s.n_in > 3 || s.H <= 0 || s.W <= 0 || s.Cout <= 0 || (s.Cout & 3)) return MYDET_E_BADARG;
        if (!s.w_dw || !s.w_pw_packed || !s.shift || !s.y || !al16(s.w_dw) || !al16(s.w_pw_packed) || !al16(s.shift) ||
            !al16(s.y) || (s.scale && !al16(s.scale)) || (s.ldy & 3) || s.ldy < s.Cout)
            return MYDET_E_BADARG;
        if (s.n_in > 1 && !s.fuse_weights) return MYDET_E_BADARG;
        if (s.act != MYDET_ACT_NONE && s.act != MYDET_ACT_SWISH) return MYDET_E_UNSUPP;
        for (int k = 0; k < 3; ++k) {
            p.in[k] = s.in[k]; p.ld[k] = s.ld[k]; p.mode[k] = s.mode[k];
            if (k < s.n_in) {
                if (!s.in[k] || !al16(s.in[k]) || (s.ld[k] & 3) || s.ld[k] < C || s.mode[k] < 0 || s.mode[k] > 2)
                    return MYDET_E_BADARG;
                if (s.mode[k] == 1 && ((s.H & 1) || (s.W & 1))) return MYDET_E_BADARG;
                if (s.n_in == 1 && s.mode[k] != 0) return MYDET_E_BADARG;
            } else {
                // a slot beyond n_in is never read; give it input 0's address all the same, so that no address
                // arithmetic in the kernel can ever start from a null pointer (the cause of round 2's fault)
                p.in[k] = s.in[0]; p.ld[k] = s.ld[0]; p.mode[k] = 0;
            }
        }
        p.n_in = s.n_in; p.fuse_w = s.fuse_weights; p.wd = s.w_dw; p.wpk = s.w_pw_packed; p.scale = s.scale;
        p.shift = s.shift; p.y = s.y; p.ldy = s.ldy; p.H = s.H; p.W = s.W; p.Cout = s.Cout; p.act = s.act;
        p.nb = (s.Cout + 15) / 16;
        p.tiles_x = (s.W + TS - 1) / TS;
        p.tiles_per_img = p.tiles_x * ((s.H + TS - 1) / TS);
        p.tile_begin = (int)tiles;
        tiles += (int64_t)p.tiles_per_img * B;
        if (tiles > 0x7fffffff) return MYDET_E_UNSUPP;
    }
    // launches with fewer workgroups than CUs cut their nodes along the output channels too (pairs of 16-channel blocks)
    const int64_t base_tiles = tiles;
    tiles = 0;
    for (int i = 0; i < n; ++i) {
        SpNode &p = a.p[i];
        const int pairs = (p.nb + 1) / 2;
        int split = 1;
        // (to one workgroup per CU, not three -- round 4: with two batch lanes in flight the other lane fills the chip, and every
        // extra slice repeats the node's depthwise phase: D1 +1.2 %, D1-FCOS2-ATSS +1.9 %)
        if (base_tiles < mydet_cu_count()) split = (int)((mydet_cu_count() + base_tiles - 1) / base_tiles);
        static const int forced_split = [] { const char *e = getenv("MYDET_SEPCONV_SPLIT"); return e ? atoi(e) : 0; }();     // tuning knob, read once
        if (forced_split > split) split = forced_split;
        if (split > pairs) split = pairs;
        p.nb_per = 2 * ((pairs + split - 1) / split);
        p.nsplit = (p.nb + p.nb_per - 1) / p.nb_per;
        p.tile_begin = (int)tiles;
        tiles += (int64_t)p.tiles_per_img * B * p.nsplit;
        if (tiles > 0x7fffffff) return MYDET_E_UNSUPP;           // (a forced split can exceed what the first check saw)
    }
    a.total = (int)tiles;
    hipLaunchKernelGGL((sepconv_kernel<22>), dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, a);
    return mydet_launch_status();
}

extern "C" int mydet_sepconv_decode_retina_f32(int n, const mydet_sepconv_decode_node *nodes, int B, int C, int A, int n_cls,
                                               int img_h, int img_w, float *bbox, int64_t *class_idx, float *score,
                                               int64_t N, void *stream) {
    if (n <= 0 || n > SP_MAX || !nodes || B <= 0 || A <= 0 || n_cls <= 0 || N <= 0 || !bbox || !class_idx || !score)
        return MYDET_E_BADARG;
    if (C != 88 || A > DEC_MAX_A) return MYDET_E_UNSUPP;
    const int nba = (n_cls + 15) / 16;                       // channel blocks per anchor of a class node
    if (nba != 5 && nba != 6) return MYDET_E_UNSUPP;         // 65..96 classes (COCO's 80 / 90 / 91)
    if (!al16(bbox)) return MYDET_E_BADARG;
    SpDecArgs a;
    a.n = n; a.B = B; a.A = A; a.n_cls = n_cls; a.img_h = img_h; a.img_w = img_w; a.N = N;
    a.bbox = bbox; a.score = score; a.cidx = class_idx;
    int64_t tiles = 0;
    for (int i = 0; i < n; ++i) {
        const mydet_sepconv_node &s = nodes[i].node;
        SpNode &p = a.p[i];
        SpDecNode &d = a.d[i];
        if (nodes[i].kind != 0 && nodes[i].kind != 1) return MYDET_E_BADARG;
        if (s.n_in != 1 || s.mode[0] != 0 || s.H <= 0 || s.W <= 0 || s.act != MYDET_ACT_NONE) return MYDET_E_BADARG;
        if (s.Cout != (nodes[i].kind == 0 ? A * nba * 16 : A * 4)) return MYDET_E_BADARG;
        if (!s.in[0] || !al16(s.in[0]) || (s.ld[0] & 3) || s.ld[0] < C || !s.w_dw || !s.w_pw_packed || !s.shift ||
            !al16(s.w_dw) || !al16(s.w_pw_packed) || !al16(s.shift) || (s.scale && !al16(s.scale)))
            return MYDET_E_BADARG;
        if (nodes[i].kind == 1 && !nodes[i].anchors_wh) return MYDET_E_BADARG;
        if (nodes[i].n_off < 0 || nodes[i].n_off + (int64_t)A * s.H * s.W > N) return MYDET_E_BADARG;
        for (int k = 0; k < 3; ++k) { p.in[k] = s.in[0]; p.ld[k] = s.ld[0]; p.mode[k] = 0; }
        p.n_in = 1; p.fuse_w = nullptr; p.wd = s.w_dw; p.wpk = s.w_pw_packed; p.scale = s.scale; p.shift = s.shift;
        p.y = nullptr; p.ldy = 0; p.H = s.H; p.W = s.W; p.Cout = s.Cout; p.act = MYDET_ACT_NONE;
        p.nb = (s.Cout + 15) / 16;
        p.tiles_x = (s.W + TS - 1) / TS;
        p.tiles_per_img = p.tiles_x * ((s.H + TS - 1) / TS);
        d.kind = nodes[i].kind; d.stride = nodes[i].stride; d.n_off = nodes[i].n_off;
        for (int k = 0; k < DEC_MAX_A; ++k) {
            d.aw[k] = nodes[i].kind == 1 && k < A ? nodes[i].anchors_wh[2 * k] : 0.0f;
            d.ah[k] = nodes[i].kind == 1 && k < A ? nodes[i].anchors_wh[2 * k + 1] : 0.0f;
        }
        tiles += (int64_t)p.tiles_per_img * B;
        if (tiles > 0x7fffffff) return MYDET_E_UNSUPP;
    }
    // small launches cut the class nodes along the anchors
    const int64_t base_tiles = tiles;
    tiles = 0;
    for (int i = 0; i < n; ++i) {
        SpNode &p = a.p[i];
        int split = 1;
        if (a.d[i].kind == 0 && base_tiles < 2 * mydet_cu_count()) split = (int)((3 * mydet_cu_count() + base_tiles - 1) / base_tiles);
        if (split > A) split = A;
        p.nb_per = a.d[i].kind == 0 ? nba * ((A + split - 1) / split) : p.nb;
        p.nsplit = (p.nb + p.nb_per - 1) / p.nb_per;
        p.tile_begin = (int)tiles;
        tiles += (int64_t)p.tiles_per_img * B * p.nsplit;
        if (tiles > 0x7fffffff) return MYDET_E_UNSUPP;
    }
    if (nba == 5)
        hipLaunchKernelGGL((sepconv_decode_kernel<22, 5>), dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL((sepconv_decode_kernel<22, 6>), dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, a);
    return mydet_launch_status();
}
