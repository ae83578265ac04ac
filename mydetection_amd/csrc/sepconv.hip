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
//   1. the 10x10 halo of pre(x) -> LDS (float4 per thread, inputs read once; out-of-image pixels are the conv's zeros)
//   2. depthwise 3x3 on the VALU from LDS (same fmaf order as dwconv_kernel), results kept in registers, then stored
//      over the halo as the [64 px][C+1] operand tile (odd stride: conflict-free fragment reads)
//   3. pointwise conv on FP32 MFMA, computed transposed (A = weights, B = pixels: v_mfma_f32_16x16x4_f32), so a lane
//      ends up with 4 consecutive output channels of one pixel = one 16-byte store after the BN/act epilogue.  A wave
//      owns 16 pixels; the weights come pre-packed in fragment order (one coalesced 256-B load per k-step).
// LDS 35 KB -> 4 workgroups per CU: the halo loads of one overlap the MFMAs of another.
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

template <int KS>
__device__ __forceinline__ void sp_load_pair(const SpNode &P, int nb, int lane, float (&f0)[KS], float (&f1)[KS]) {
    const float *wp0 = P.wpk + (int64_t)min(nb, P.nb - 1) * KS * 64 + lane;
    const float *wp1 = P.wpk + (int64_t)min(nb + 1, P.nb - 1) * KS * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        f0[ks] = wp0[ks * 64];
        f1[ks] = wp1[ks * 64];
    }
}

__device__ __forceinline__ void sp_finish(const SpNode &P, f32x4 acc, int n, bool valid, float *yp) {
    if (valid && n < P.Cout) {
        const f32x4 sh = *reinterpret_cast<const f32x4 *>(P.shift + n);
        if (P.scale) {
            const f32x4 sc = *reinterpret_cast<const f32x4 *>(P.scale + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = acc[e] * sc[e] + sh[e];
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = acc[e] + sh[e];
        }
        if (P.act == MYDET_ACT_SWISH) {
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = acc[e] * mydet_sigmoid_fast(acc[e]);
        }
        *reinterpret_cast<f32x4 *>(yp + n) = acc;
    }
}

struct SpItem {
    int pi, b, oy0, ox0, nb_begin, nb_end;
};

// work item = (tile, slice of the output-channel blocks)
__device__ __forceinline__ SpItem sp_decode(const SpArgs &a, int item) {
    int pi = 0;
    for (int i = 1; i < a.n; ++i)
        if (item >= a.p[i].tile_begin) pi = i;                 // uniform
    const SpNode &P = a.p[pi];
    const int t0 = item - P.tile_begin;
    const int t = t0 / P.nsplit, ns = t0 - t * P.nsplit;
    SpItem it;
    it.pi = pi;
    it.nb_begin = ns * P.nb_per;
    it.nb_end = min(P.nb, it.nb_begin + P.nb_per);
    it.b = t / P.tiles_per_img;
    const int r = t - it.b * P.tiles_per_img;
    const int ty = r / P.tiles_x;
    it.oy0 = ty * TS;
    it.ox0 = (r - ty * P.tiles_x) * TS;
    return it;
}

template <int KS>
__global__ __launch_bounds__(256, 4) void sepconv_kernel(const SpArgs a) {
    constexpr int C = KS * 4, Q = KS, XS = C + 1;
    constexpr int LDS_FLOATS = HS * HS * C > TS * TS * XS ? HS * HS * C : TS * TS * XS;
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
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

    // 1. halo of pre(x)
    float w0 = 0.f, w1 = 0.f, w2 = 0.f;
    if (P.n_in > 1) {                                         // w = relu(weights); w = w / (sum(w) + 0.0001)
        w0 = fmaxf(P.fuse_w[0], 0.0f);
        w1 = fmaxf(P.fuse_w[1], 0.0f);
        w2 = P.n_in > 2 ? fmaxf(P.fuse_w[2], 0.0f) : 0.0f;
        float sum = w0 + w1;
        if (P.n_in > 2) sum += w2;
        sum += 0.0001f;
        w0 = w0 / sum; w1 = w1 / sum; w2 = w2 / sum;
    }
    // three halo items per thread in flight (otherwise a thread waits out one HBM latency per item); the single-input
    // loads are unconditional at clamped coordinates (a branch per load would serialise them)
    constexpr int NH = (HS * HS * Q + 255) / 256, HB = 3;
#pragma unroll
    for (int j0 = 0; j0 < NH; j0 += HB) {
        f32x4 hv[HB];
#pragma unroll
        for (int u = 0; u < HB; ++u) {
            const int it = tid + (j0 + u) * 256;
            const int itc = min(it, HS * HS * Q - 1);
            const int hp = itc / Q, q = itc - hp * Q;
            const int hy = hp / HS, hx = hp - hy * HS;
            const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
            const bool in = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (P.n_in == 1) {
                const int cy = min(max(iy, 0), H - 1), cx = min(max(ix, 0), W - 1);
                const f32x4 ld = *reinterpret_cast<const f32x4 *>(P.in[0] + (((int64_t)b * H + cy) * W + cx) * P.ld[0] + q * 4);
                v = in ? ld : v;
            } else if (in) {                                  // python sum(): 0 + w0*x0 + w1*x1 (+ w2*x2), then swish
                const f32x4 v0 = sp_read(P, 0, b, iy, ix, q), v1 = sp_read(P, 1, b, iy, ix, q);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = w0 * v0[j] + w1 * v1[j];
                if (P.n_in > 2) {
                    const f32x4 v2 = sp_read(P, 2, b, iy, ix, q);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = v[j] + w2 * v2[j];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = v[j] * mydet_sigmoid_fast(v[j]);
            }
            hv[u] = v;
        }
#pragma unroll
        for (int u = 0; u < HB; ++u) {
            const int it = tid + (j0 + u) * 256;
            if (it < HS * HS * Q) *reinterpret_cast<f32x4 *>(&lds[it * 4]) = hv[u];      // it * 4 == hp * C + q * 4
        }
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
        for (int k = 0; k < 9; ++k) wv[k] = *reinterpret_cast<const f32x4 *>(P.wd + k * C + dq * 4);
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

    // 3. pointwise conv, transposed: D[channel][pixel] = sum_k W[channel][k] * X[pixel][k]
    const int wave = tid >> 6, lane = tid & 63;
    const int m0 = wave * 16;
    float bf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) bf[ks] = lds[(m0 + (lane & 15)) * XS + ks * 4 + (lane >> 4)];
    const int px = m0 + (lane & 15);
    const int oy = oy0 + (px >> 3), ox = ox0 + (px & 7);
    const bool valid = oy < H && ox < W;
    float *yp = P.y + (((int64_t)b * H + oy) * W + ox) * P.ldy;
    const int nsub = (lane >> 4) * 4;
    auto finish = [&](f32x4 acc, int n) {
        if (valid && n < P.Cout) {
            const f32x4 sh = *reinterpret_cast<const f32x4 *>(P.shift + n);
            if (P.scale) {
                const f32x4 sc = *reinterpret_cast<const f32x4 *>(P.scale + n);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = acc[e] * sc[e] + sh[e];
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = acc[e] + sh[e];
            }
            if (P.act == MYDET_ACT_SWISH) {
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = acc[e] * mydet_sigmoid_fast(acc[e]);
            }
            *reinterpret_cast<f32x4 *>(yp + n) = acc;
        }
    };
    // one 16-channel block at a time; the NEXT block's weight fragments are in flight under the current block's MFMAs
    // (same register budget as loading a pair and then computing it, without the exposed L2 latency per pair).  Two
    // accumulator chains over the even / odd k-steps keep the matrix pipe issuing (40-cycle dependent latency).
    auto load_block = [&](int nb, float (&f)[KS]) {
        const float *wp = P.wpk + (int64_t)min(nb, P.nb - 1) * KS * 64 + lane;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) f[ks] = wp[ks * 64];
    };
    float af[KS];
    load_block(nb_begin, af);
    for (int nb = nb_begin; nb < nb_end; ++nb) {
        float an[KS];
        load_block(nb + 1, an);
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ks += 2) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ks], bf[ks], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ks + 1], bf[ks + 1], acc1, 0, 0, 0);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) acc0[e] = acc0[e] + acc1[e];
        finish(acc0, nb * 16 + nsub);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) af[ks] = an[ks];
    }
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
    // launches that do not fill the chip cut their nodes along the output channels too (pairs of 16-channel blocks)
    const int64_t base_tiles = tiles;
    tiles = 0;
    for (int i = 0; i < n; ++i) {
        SpNode &p = a.p[i];
        const int pairs = (p.nb + 1) / 2;
        int split = 1;
        if (base_tiles < 2 * mydet_cu_count()) split = (int)((3 * mydet_cu_count() + base_tiles - 1) / base_tiles);
        if (const char *e = getenv("MYDET_SEPCONV_SPLIT")) split = atoi(e) > split ? atoi(e) : split;     // tuning knob
        if (split > pairs) split = pairs;
        p.nb_per = 2 * ((pairs + split - 1) / split);
        p.nsplit = (p.nb + p.nb_per - 1) / p.nb_per;
        p.tile_begin = (int)tiles;
        tiles += (int64_t)p.tiles_per_img * B * p.nsplit;
    }
    a.total = (int)tiles;
    hipLaunchKernelGGL((sepconv_kernel<22>), dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, a);
    return mydet_launch_status();
}
