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
