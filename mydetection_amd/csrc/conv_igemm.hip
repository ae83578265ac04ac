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
// on gfx950).  K is walked in steps of 32: the A and B slabs are staged global ->
// registers -> LDS (rows padded to 36 floats so ds_read_b128 fragment reads and
// ds_write_b128 staging writes are bank-conflict free), two LDS buffers, one barrier
// per step, next slab's global loads in flight under the current slab's MFMAs.
// Each lane reads 4 consecutive k of its row with one ds_read_b128; lane half h owns
// k = 8*kc + 4*h + t at MFMA step t -- a permutation of k applied to A and B alike.
//
// Replaces the ATen conv2d/batch_norm/leaky_relu chain under models/modules.py:94-95,
// the residual add of models/modules.py:69-73 and the head convs models/rpns.py:24-25.
#include "common.h"

namespace {

constexpr int BK = 32;          // k per LDS slab
constexpr int LDS_LD = BK + 4;  // padded row (floats)

struct ConvArgs {
    const float *x, *w, *scale, *shift, *res;
    float *y;
    int64_t ldx, ldr, ldy;
    int H, W, Cin, Cout, KH, KW, stride, pad_t, pad_l, Ho, Wo, act;
    int M, K, ntiles, nblk;
};

template <int BM, int BN, int WM, int WN, bool CIN32>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvArgs p) {
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int AI = BM / 32, BI = BN / 32;      // 16-byte chunks each thread stages per slab
    static_assert(WM * WN == 4, "4 waves");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As = smem;                               // [2][BM][LDS_LD]
    float *Bs = smem + 2 * BM * LDS_LD;             // [2][BN][LDS_LD]

    const int tid = threadIdx.x;
    const int lid = mydet_xcd_remap(blockIdx.x, p.nblk);
    const int m0 = (lid / p.ntiles) * BM;
    const int n0 = (lid % p.ntiles) * BN;

    // ---- staging role: chunk (4 floats) `sc` of rows sr + 32*i
    const int sc = tid & 7, sr = tid >> 3;
    const float *aptr[AI];
    int ih0[AI], iw0[AI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const int m = m0 + sr + 32 * i;
        const int mm = m < p.M ? m : 0;
        const int ow = mm % p.Wo, t = mm / p.Wo;
        const int oh = t % p.Ho, b = t / p.Ho;
        ih0[i] = m < p.M ? oh * p.stride - p.pad_t : -(1 << 28);
        iw0[i] = ow * p.stride - p.pad_l;
        aptr[i] = p.x + ((int64_t)(b * p.H + oh * p.stride - p.pad_t) * p.W + iw0[i]) * p.ldx;
    }
    const float *bptr[BI];
    bool bok[BI];
#pragma unroll
    for (int i = 0; i < BI; ++i) {
        const int n = n0 + sr + 32 * i;
        bok[i] = n < p.Cout;
        bptr[i] = p.w + (int64_t)(bok[i] ? n : 0) * p.K + sc * 4;
    }

    f32x4 areg[AI], breg[BI];
    const int nk = (p.K + BK - 1) / BK;
    int kh = 0, kw = 0, c0 = 0;                      // CIN32 path: tap and channel base of the slab

    auto load_slab = [&](int kt) {
        int kkh, kkw, cc;
        bool kok = true;
        if (CIN32) {
            kkh = kh; kkw = kw; cc = c0 + sc * 4;
        } else {
            const int k = kt * BK + sc * 4;          // Cin % 4 == 0: a chunk never straddles taps
            kok = k < p.K;
            const int tap = k / p.Cin;
            cc = k - tap * p.Cin;
            kkh = tap / p.KW; kkw = tap - kkh * p.KW;
        }
        const int64_t tapoff = ((int64_t)kkh * p.W + kkw) * p.ldx + cc;
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const bool ok = kok && (unsigned)(ih0[i] + kkh) < (unsigned)p.H &&
                            (unsigned)(iw0[i] + kkw) < (unsigned)p.W;
            areg[i] = ok ? *reinterpret_cast<const f32x4 *>(aptr[i] + tapoff) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            const bool ok = kok && bok[i];
            breg[i] = ok ? *reinterpret_cast<const f32x4 *>(bptr[i] + (int64_t)kt * BK)
                         : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (CIN32) {                                  // advance (uniform, scalar)
            c0 += BK;
            if (c0 == p.Cin) { c0 = 0; if (++kw == p.KW) { kw = 0; ++kh; } }
        }
    };
    auto store_slab = [&](int buf) {
        float *a = As + buf * BM * LDS_LD, *b = Bs + buf * BN * LDS_LD;
#pragma unroll
        for (int i = 0; i < AI; ++i)
            *reinterpret_cast<f32x4 *>(a + (sr + 32 * i) * LDS_LD + sc * 4) = areg[i];
#pragma unroll
        for (int i = 0; i < BI; ++i)
            *reinterpret_cast<f32x4 *>(b + (sr + 32 * i) * LDS_LD + sc * 4) = breg[i];
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

    load_slab(0);
    store_slab(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_slab(kt + 1);
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
        if (kt + 1 < nk) store_slab(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: lane = output channel, register = output pixel
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + (wn * TN + j) * 32 + fr;
        const bool nok = n < p.Cout;
        const float scl = (nok && p.scale) ? p.scale[n] : 1.0f;
        const float sft = (nok && p.shift) ? p.shift[n] : 0.0f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (nok && m < p.M) {
                    float v = mydet_act(acc[i][j][r] * scl + sft, p.act);
                    if (p.res) v += p.res[(int64_t)m * p.ldr + n];
                    p.y[(int64_t)m * p.ldy + n] = v;
                }
            }
        }
    }
}

template <int BM, int BN, int WM, int WN>
int launch(const ConvArgs &a0, hipStream_t stream) {
    ConvArgs a = a0;
    const int mtiles = (a.M + BM - 1) / BM;
    a.ntiles = (a.Cout + BN - 1) / BN;
    a.nblk = mtiles * a.ntiles;
    const size_t lds = (size_t)2 * (BM + BN) * LDS_LD * sizeof(float);
    const bool cin32 = (a.Cin % 32) == 0;
    static bool attr_set = false;                  // > 64 KiB of dynamic LDS needs the opt-in once
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_igemm_kernel<BM, BN, WM, WN, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_igemm_kernel<BM, BN, WM, WN, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    if (cin32)
        hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, true>), dim3(a.nblk), dim3(256), lds, stream, a);
    else
        hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, false>), dim3(a.nblk), dim3(256), lds, stream, a);
    return mydet_launch_status();
}

}  // namespace

extern "C" int mydet_conv2d_igemm_f32(const float *x, int64_t ldx, const float *w, const float *scale,
                                      const float *shift, const float *residual, int64_t ldr, float *y,
                                      int64_t ldy, int B, int H, int W, int Cin, int Cout, int KH, int KW,
                                      int stride, int pad_t, int pad_l, int Ho, int Wo, int act, void *stream) {
    if (!x || !w || !y || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || KH <= 0 || KW <= 0 ||
        stride <= 0 || Ho <= 0 || Wo <= 0)
        return MYDET_E_BADARG;
    if ((Cin & 3) || (ldx & 3) || ldx < Cin || ldy < Cout || (residual && ldr < Cout)) return MYDET_E_BADARG;
    if (((uintptr_t)x & 15) || ((uintptr_t)w & 15)) return MYDET_E_BADARG;
    const int64_t M64 = (int64_t)B * Ho * Wo;
    if (M64 > (int64_t)1 << 30 || (int64_t)KH * KW * Cin > (int64_t)1 << 30) return MYDET_E_BADARG;
    ConvArgs a;
    a.x = x; a.w = w; a.scale = scale; a.shift = shift; a.res = residual; a.y = y;
    a.ldx = ldx; a.ldr = ldr; a.ldy = ldy;
    a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.KH = KH; a.KW = KW; a.stride = stride;
    a.pad_t = pad_t; a.pad_l = pad_l; a.Ho = Ho; a.Wo = Wo; a.act = act;
    a.M = (int)M64; a.K = KH * KW * Cin; a.ntiles = 0; a.nblk = 0;
    hipStream_t s = (hipStream_t)stream;
    // Tile choice: widest N tile the layer fills; fall back to smaller tiles when the
    // grid would leave most of the 256 CUs idle.
    const int64_t blocks128 = ((M64 + 127) / 128) * ((Cout + 127) / 128);
    if (Cout <= 32) return launch<128, 32, 4, 1>(a, s);
    if (Cout <= 64) return launch<128, 64, 2, 2>(a, s);
    if (blocks128 >= 384) return launch<128, 128, 2, 2>(a, s);
    const int64_t blocks64n = ((M64 + 127) / 128) * ((Cout + 63) / 64);
    if (blocks64n >= 256) return launch<128, 64, 2, 2>(a, s);
    return launch<64, 64, 2, 2>(a, s);
}
