// First-layer 3x3 convolution, Cin = 3 -> Cout = 32, direct form on the vector ALU.
//
// K = 27 is too short for the MFMA tile and the layer is write-bound (each output
// pixel stores 128 B and reads 108 B of image through L1/L2), so this is a plain
// per-pixel kernel: a thread owns one output pixel, keeps its 27 taps in registers and
// walks the 32 output channels; the 864 weights are wave-uniform and are fetched
// through the scalar cache (s_load), the epilogue (folded BN + activation) is fused and
// the wave's 64x32 outputs are transposed through a wave-private LDS strip so that every store
// instruction writes whole 128-byte lines (NHWC).
// It reads the image through arbitrary strides, so the NCHW tensor built at
// api/detection.py:160-163 is consumed as is (no layout pass).
// Replaces Darknet53 netlist[0] (models/backbones.py:14) and the EfficientNet stem
// (models/backbones.py:210, static-SAME pad 0/1 via pad_t/pad_l).
#include "common.h"

namespace {

struct StemArgs {
    const float *x, *w, *scale, *shift;
    float *y;
    int64_t sxb, sxc, sxh, sxw, ldy;
    int H, W, stride, pad_t, pad_l, Ho, Wo, act;
    int64_t M;
};

constexpr int ST_LD = 36;          // LDS row per pixel: 32 channels + pad (16-byte aligned, conflict-free b128)

__global__ __launch_bounds__(256) void conv_stem_kernel(const StemArgs p) {
    __shared__ __attribute__((aligned(16))) float tile[4][32 * ST_LD];     // one strip per wave: 32 pixels at a time
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t mw = (int64_t)blockIdx.x * 256 + wave * 64;               // first pixel of this wave
    const int64_t m = mw + lane;
    float *strip = tile[wave];
    f32x4 o[8];
    if (m < p.M) {
        const int ow = (int)(m % p.Wo);
        const int64_t t = m / p.Wo;
        const int oh = (int)(t % p.Ho);
        const int64_t b = t / p.Ho;
        float in[27];                               // [kh][kw][c], matches the OHWI weight rows
        const float *xb = p.x + b * p.sxb;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int ih = oh * p.stride - p.pad_t + kh;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int iw = ow * p.stride - p.pad_l + kw;
                const bool ok = (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    in[(kh * 3 + kw) * 3 + c] = ok ? xb[c * p.sxc + (int64_t)ih * p.sxh + (int64_t)iw * p.sxw] : 0.0f;
            }
        }
#pragma unroll
        for (int n4 = 0; n4 < 8; ++n4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n4 * 4 + j;
                const float *wr = p.w + n * 27;     // uniform address -> scalar loads
                float acc = 0.0f;
#pragma unroll
                for (int k = 0; k < 27; ++k) acc = fmaf(in[k], wr[k], acc);
                const float scl = p.scale ? p.scale[n] : 1.0f;
                const float sft = p.shift ? p.shift[n] : 0.0f;
                o[n4][j] = mydet_act(acc * scl + sft, p.act);
            }
        }
    }
    // transposed write-out, the wave's pixels in two halves of 32 (the strip is 4.6 KB per wave -- 18 KB per workgroup,
    // eight workgroups per CU): 8 consecutive lanes emit one pixel's 128-byte line, an instruction writes 1 KB
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if ((lane >> 5) == h) {
#pragma unroll
            for (int n4 = 0; n4 < 8; ++n4) *reinterpret_cast<f32x4 *>(strip + (lane & 31) * ST_LD + n4 * 4) = o[n4];
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = lane + 64 * j, px = q >> 3, part = q & 7;
            if (mw + 32 * h + px < p.M)
                __builtin_nontemporal_store(*reinterpret_cast<const f32x4 *>(strip + px * ST_LD + part * 4),
                                            reinterpret_cast<f32x4 *>(p.y + (mw + 32 * h + px) * p.ldy + part * 4));
        }
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace

extern "C" int mydet_conv2d_stem_f32(const float *x, int64_t sxb, int64_t sxc, int64_t sxh, int64_t sxw,
                                     const float *w, const float *scale, const float *shift, float *y,
                                     int64_t ldy, int B, int H, int W, int Cout, int stride, int pad_t,
                                     int pad_l, int Ho, int Wo, int act, void *stream) {
    if (!x || !w || !y || B <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0 || stride <= 0) return MYDET_E_BADARG;
    if (Cout != 32) return MYDET_E_UNSUPP;
    if ((ldy & 3) || ldy < Cout || ((uintptr_t)y & 15)) return MYDET_E_BADARG;
    StemArgs a;
    a.x = x; a.w = w; a.scale = scale; a.shift = shift; a.y = y;
    a.sxb = sxb; a.sxc = sxc; a.sxh = sxh; a.sxw = sxw; a.ldy = ldy;
    a.H = H; a.W = W; a.stride = stride; a.pad_t = pad_t; a.pad_l = pad_l; a.Ho = Ho; a.Wo = Wo; a.act = act;
    a.M = (int64_t)B * Ho * Wo;
    const int64_t blocks = (a.M + 255) / 256;
    if (blocks > 0x7fffffff) return MYDET_E_BADARG;
    hipLaunchKernelGGL(conv_stem_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    return mydet_launch_status();
}
