// Pairwise IoU and box rescaling -- small element-wise kernels kept in the reference's
// exact float32 operation order (built with -ffp-contract=off).
//   bboxes_iou          utils/bbox_ops.py:6-49   (chainercv form: en = prod(tl < br))
//   bboxes_to_original_ utils/structures.py:175-189
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void iou_kernel(const float *a, int Na, const float *b, int Nb, int xyxy,
                                                  float *out) {
    const int64_t total = (int64_t)Na * Nb;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int ia = (int)(i / Nb), ib = (int)(i - (int64_t)ia * Nb);
        const f32x4 p = *reinterpret_cast<const f32x4 *>(a + (int64_t)ia * 4);
        const f32x4 q = *reinterpret_cast<const f32x4 *>(b + (int64_t)ib * 4);
        float tlx, tly, brx, bry, area_a, area_b;
        if (xyxy) {
            tlx = fmaxf(p[0], q[0]); tly = fmaxf(p[1], q[1]);
            brx = fminf(p[2], q[2]); bry = fminf(p[3], q[3]);
            area_a = (p[2] - p[0]) * (p[3] - p[1]);
            area_b = (q[2] - q[0]) * (q[3] - q[1]);
        } else {
            tlx = fmaxf(p[0] - p[2] / 2.0f, q[0] - q[2] / 2.0f);
            tly = fmaxf(p[1] - p[3] / 2.0f, q[1] - q[3] / 2.0f);
            brx = fminf(p[0] + p[2] / 2.0f, q[0] + q[2] / 2.0f);
            bry = fminf(p[1] + p[3] / 2.0f, q[1] + q[3] / 2.0f);
            area_a = p[2] * p[3];
            area_b = q[2] * q[3];
        }
        const float en = ((tlx < brx) ? 1.0f : 0.0f) * ((tly < bry) ? 1.0f : 0.0f);
        const float area_i = ((brx - tlx) * (bry - tly)) * en;
        out[i] = area_i / (area_a + area_b - area_i);
    }
}

__global__ __launch_bounds__(256) void to_original_kernel(float *bbox, int64_t n, float ori_w, float ori_h,
                                                          float tl_x, float tl_y, float imw, float imh) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    f32x4 v = *reinterpret_cast<f32x4 *>(bbox + i * 4);
    v[0] = (v[0] - tl_x) / imw * ori_w;
    v[1] = (v[1] - tl_y) / imh * ori_h;
    v[2] = v[2] / imw * ori_w;
    v[3] = v[3] / imh * ori_h;
    *reinterpret_cast<f32x4 *>(bbox + i * 4) = v;
}

}  // namespace

extern "C" int mydet_abi_version(void) { return MYDET_ABI_VERSION; }

extern "C" int mydet_bboxes_iou_f32(const float *a, int Na, const float *b, int Nb, int xyxy, float *iou,
                                    void *stream) {
    if (Na < 0 || Nb < 0) return MYDET_E_BADARG;
    if (Na == 0 || Nb == 0) return 0;
    if (!a || !b || !iou || ((uintptr_t)a & 15) || ((uintptr_t)b & 15)) return MYDET_E_BADARG;
    int64_t blocks = ((int64_t)Na * Nb + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(iou_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, Na, b, Nb, xyxy, iou);
    return mydet_launch_status();
}

extern "C" int mydet_bboxes_to_original_f32(float *bbox, int64_t n, float ori_w, float ori_h, float tl_x,
                                            float tl_y, float imw, float imh, void *stream) {
    if (n < 0) return MYDET_E_BADARG;
    if (n == 0) return 0;
    if (!bbox || ((uintptr_t)bbox & 15)) return MYDET_E_BADARG;
    hipLaunchKernelGGL(to_original_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, bbox,
                       n, ori_w, ori_h, tl_x, tl_y, imw, imh);
    return mydet_launch_status();
}

// ---------------------------------------------------------------------------------------------------------
// Device-side image preparation: uint8 HWC image(s) -> float32 planar CHW,
//   zero-pad right/bottom to (Hp, Wp) in the uint8 domain   utils/image_ops.py:38-52 (pad_to_divisible)
//   x / 255                                                  tvf.to_tensor, api/detection.py:160
//   (x - mean[c]) / std[c] when norm != 0                    utils/image_ops.py:177-180 ('RGB_1_norm')
// in exactly that operation order.  One thread per output pixel, the three channels of a pixel together.
namespace {
__global__ __launch_bounds__(256) void preprocess_kernel(const unsigned char *img, int H, int W, float *out, int Hp,
                                                         int Wp, int norm, float m0, float m1, float m2, float s0,
                                                         float s1, float s2, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int x = (int)(i % Wp);
    const int64_t t = i / Wp;
    const int y = (int)(t % Hp);
    const int64_t b = t / Hp;
    float v[3] = {0.f, 0.f, 0.f};
    if (y < H && x < W) {
        const unsigned char *p = img + ((b * H + y) * W + x) * 3;
        v[0] = (float)p[0] / 255.0f; v[1] = (float)p[1] / 255.0f; v[2] = (float)p[2] / 255.0f;
    }
    if (norm) {
        v[0] = (v[0] - m0) / s0; v[1] = (v[1] - m1) / s1; v[2] = (v[2] - m2) / s2;
    }
    const int64_t plane = (int64_t)Hp * Wp;
    float *o = out + b * 3 * plane + (int64_t)y * Wp + x;
    o[0] = v[0]; o[plane] = v[1]; o[2 * plane] = v[2];
}
}  // namespace

extern "C" int mydet_preprocess_u8_f32(const unsigned char *img, int B, int H, int W, float *out, int Hp, int Wp,
                                       int norm, const float *mean3, const float *std3, void *stream) {
    if (!img || !out || B <= 0 || H <= 0 || W <= 0 || Hp < H || Wp < W) return MYDET_E_BADARG;
    if (norm && (!mean3 || !std3)) return MYDET_E_BADARG;
    const int64_t total = (int64_t)B * Hp * Wp;
    const float m0 = norm ? mean3[0] : 0.f, m1 = norm ? mean3[1] : 0.f, m2 = norm ? mean3[2] : 0.f;
    const float s0 = norm ? std3[0] : 1.f, s1 = norm ? std3[1] : 1.f, s2 = norm ? std3[2] : 1.f;
    hipLaunchKernelGGL(preprocess_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, img, H,
                       W, out, Hp, Wp, norm, m0, m1, m2, s0, s1, s2, total);
    return mydet_launch_status();
}
