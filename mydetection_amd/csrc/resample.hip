// Nearest-neighbour resize fused with channel concatenation (NHWC, 16-byte lanes).
//
//   y[b,yo,xo, 0:C1]     = a[b, src(yo), src(xo), :]      src(i) = min(floor(i * in/out), in-1)
//   y[b,yo,xo, C1:C1+C2] = b[b, yo, xo, :]
// HBM-bound: every byte is read once and written once; one thread moves one float4.
// Replaces F.interpolate(..., mode='nearest') + torch.cat((pre, x), dim=1) of
// YOLOBranch.forward (models/fpns.py:62-65) and BiFPN's upsample2x (models/fpns.py:442-444).
#include "common.h"

namespace {

struct UpcatArgs {
    const float *a, *b;
    float *y;
    int64_t lda, ldb, ldy;
    int Ha, Wa, C1, C2, Ho, Wo;
    float sh, sw;            // in/out as float, as ATen's nearest kernel computes it
    int64_t total;           // B*Ho*Wo*(C1+C2)/4
};

__global__ __launch_bounds__(256) void upsample_concat_kernel(const UpcatArgs p) {
    const int c4 = (p.C1 + p.C2) >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < p.total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % c4) * 4;
        const int64_t pix = i / c4;
        f32x4 v;
        if (c < p.C1) {
            const int xo = (int)(pix % p.Wo);
            const int64_t t = pix / p.Wo;
            const int yo = (int)(t % p.Ho);
            const int64_t b = t / p.Ho;
            int ys = (int)floorf(yo * p.sh), xs = (int)floorf(xo * p.sw);
            ys = ys < p.Ha - 1 ? ys : p.Ha - 1;
            xs = xs < p.Wa - 1 ? xs : p.Wa - 1;
            v = *reinterpret_cast<const f32x4 *>(p.a + ((b * p.Ha + ys) * p.Wa + xs) * p.lda + c);
        } else {
            v = *reinterpret_cast<const f32x4 *>(p.b + pix * p.ldb + (c - p.C1));
        }
        *reinterpret_cast<f32x4 *>(p.y + pix * p.ldy + c) = v;
    }
}

}  // namespace

extern "C" int mydet_upsample_concat_f32(const float *a, int64_t lda, int Ha, int Wa, int C1, const float *b,
                                         int64_t ldb, int C2, float *y, int64_t ldy, int B, int Ho, int Wo,
                                         void *stream) {
    if (!a || !y || B <= 0 || Ha <= 0 || Wa <= 0 || Ho <= 0 || Wo <= 0 || C1 <= 0 || C2 < 0) return MYDET_E_BADARG;
    if (C2 > 0 && !b) return MYDET_E_BADARG;
    if ((C1 & 3) || (C2 & 3) || (lda & 3) || (ldy & 3) || (C2 > 0 && (ldb & 3))) return MYDET_E_BADARG;
    if (((uintptr_t)a & 15) || ((uintptr_t)y & 15) || (b && ((uintptr_t)b & 15))) return MYDET_E_BADARG;
    UpcatArgs p;
    p.a = a; p.b = b; p.y = y; p.lda = lda; p.ldb = ldb; p.ldy = ldy;
    p.Ha = Ha; p.Wa = Wa; p.C1 = C1; p.C2 = C2; p.Ho = Ho; p.Wo = Wo;
    p.sh = (float)Ha / (float)Ho; p.sw = (float)Wa / (float)Wo;
    p.total = (int64_t)B * Ho * Wo * ((C1 + C2) >> 2);
    int64_t blocks = (p.total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;         // grid-stride the rest
    hipLaunchKernelGGL(upsample_concat_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
    return mydet_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------
// Ultralytics blocks (external/ultralytics/common.py): Focus' space-to-depth and SPP's pooled concatenation.
namespace {

struct S2dArgs {
    const float *x;
    float *y;
    int64_t sb, sc, sh, sw, ldy;
    int C, Ho, Wo;
    int64_t total;           // B*Ho*Wo*(4C/4)
};

// Focus.forward (common.py:84-86): cat([x[..., ::2, ::2], x[..., 1::2, ::2], x[..., ::2, 1::2], x[..., 1::2, 1::2]], 1)
// -> output channel g*C + c with g = 0:(dy 0, dx 0)  1:(dy 1, dx 0)  2:(dy 0, dx 1)  3:(dy 1, dx 1).  The image is read
// through its own strides (NCHW or channels-last), the result is channels-last for the conv that follows.
__global__ __launch_bounds__(256) void space_to_depth_kernel(const S2dArgs p) {
    const int c4 = p.C;                              // 4C channels = C float4 per output pixel
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < p.total; i += (int64_t)gridDim.x * 256) {
        const int q = (int)(i % c4);
        const int64_t pix = i / c4;
        const int xo = (int)(pix % p.Wo);
        const int64_t t = pix / p.Wo;
        const int yo = (int)(t % p.Ho);
        const int64_t b = t / p.Ho;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int ch = q * 4 + e, g = ch / p.C, c = ch - g * p.C;
            const int dy = g & 1, dx = g >> 1;
            v[e] = p.x[b * p.sb + c * p.sc + (int64_t)(2 * yo + dy) * p.sh + (int64_t)(2 * xo + dx) * p.sw];
        }
        *reinterpret_cast<f32x4 *>(p.y + pix * p.ldy + q * 4) = v;
    }
}

struct SppArgs {
    const float *x;
    float *y;
    int64_t ldx, ldy;
    int H, W, C, r0, r1, r2;  // window radii (k // 2), r0 <= r1 <= r2
    int64_t total;            // B*H*W*C/4
};

// SPP.forward (common.py:68-70): cat([x] + [MaxPool2d(k, 1, k // 2)(x) for k in (5, 9, 13)], 1).  One thread owns one
// float4 of channels of one pixel and walks the largest window once, keeping the three nested maxima (padding is
// -inf: taps outside the map do not take part).  The maps are the stride-32 level (20 x 20 at 640 x 640): L1-resident.
__global__ __launch_bounds__(256) void spp_concat_kernel(const SppArgs p) {
    const int c4 = p.C >> 2;
    const float ninf = -__builtin_inff();
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < p.total; i += (int64_t)gridDim.x * 256) {
        const int q = (int)(i % c4);
        const int64_t pix = i / c4;
        const int xo = (int)(pix % p.W);
        const int64_t t = pix / p.W;
        const int yo = (int)(t % p.H);
        const int64_t b = t / p.H;
        const float *xb = p.x + b * p.H * p.W * p.ldx + q * 4;
        f32x4 m0 = {ninf, ninf, ninf, ninf}, m1 = m0, m2 = m0;
        const int y_lo = max(yo - p.r2, 0), y_hi = min(yo + p.r2, p.H - 1);
        const int x_lo = max(xo - p.r2, 0), x_hi = min(xo + p.r2, p.W - 1);
        for (int yy = y_lo; yy <= y_hi; ++yy) {
            const int ady = abs(yy - yo);
            for (int xx = x_lo; xx <= x_hi; ++xx) {
                const int adx = abs(xx - xo);
                const int d = ady > adx ? ady : adx;
                const f32x4 v = *reinterpret_cast<const f32x4 *>(xb + ((int64_t)yy * p.W + xx) * p.ldx);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    m2[e] = fmaxf(m2[e], v[e]);
                    m1[e] = d <= p.r1 ? fmaxf(m1[e], v[e]) : m1[e];
                    m0[e] = d <= p.r0 ? fmaxf(m0[e], v[e]) : m0[e];
                }
            }
        }
        float *yp = p.y + pix * p.ldy + q * 4;
        *reinterpret_cast<f32x4 *>(yp) = *reinterpret_cast<const f32x4 *>(xb + ((int64_t)yo * p.W + xo) * p.ldx);
        *reinterpret_cast<f32x4 *>(yp + p.C) = m0;
        *reinterpret_cast<f32x4 *>(yp + 2 * p.C) = m1;
        *reinterpret_cast<f32x4 *>(yp + 3 * p.C) = m2;
    }
}

}  // namespace

extern "C" int mydet_space_to_depth_f32(const float *x, int64_t sb, int64_t sc, int64_t sh, int64_t sw, float *y, int64_t ldy,
                                        int B, int C, int H, int W, void *stream) {
    if (!x || !y || B <= 0 || C <= 0 || H <= 0 || W <= 0) return MYDET_E_BADARG;
    if ((H & 1) || (W & 1)) return MYDET_E_BADARG;
    if ((ldy & 3) || ldy < 4 * C || ((uintptr_t)y & 15)) return MYDET_E_BADARG;
    S2dArgs p;
    p.x = x; p.y = y; p.sb = sb; p.sc = sc; p.sh = sh; p.sw = sw; p.ldy = ldy; p.C = C; p.Ho = H / 2; p.Wo = W / 2;
    p.total = (int64_t)B * p.Ho * p.Wo * C;
    int64_t blocks = (p.total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(space_to_depth_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
    return mydet_launch_status();
}

extern "C" int mydet_spp_concat_f32(const float *x, int64_t ldx, float *y, int64_t ldy, int B, int H, int W, int C, int k0, int k1,
                                    int k2, void *stream) {
    if (!x || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0) return MYDET_E_BADARG;
    if ((C & 3) || (ldx & 3) || (ldy & 3) || ldx < C || ldy < 4 * C || ((uintptr_t)x & 15) || ((uintptr_t)y & 15)) return MYDET_E_BADARG;
    if (k0 < 1 || !(k0 & 1) || !(k1 & 1) || !(k2 & 1) || k0 > k1 || k1 > k2) return MYDET_E_BADARG;
    SppArgs p;
    p.x = x; p.y = y; p.ldx = ldx; p.ldy = ldy; p.H = H; p.W = W; p.C = C; p.r0 = k0 / 2; p.r1 = k1 / 2; p.r2 = k2 / 2;
    p.total = (int64_t)B * H * W * (C >> 2);
    int64_t blocks = (p.total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(spp_concat_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
    return mydet_launch_status();
}
