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
