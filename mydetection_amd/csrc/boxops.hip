// Pairwise IoU and box rescaling -- small element-wise kernels kept in the reference's
// exact float32 operation order (built with -ffp-contract=off).
//   bboxes_iou          utils/bbox_ops.py:6-49   (chainercv form: en = prod(tl < br))
//   bboxes_to_original_ utils/structures.py:175-189
//   cxcywh_to_x1y1x2y2  utils/bbox_ops.py:309-316
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

// cxcywh -> x1y1x2y2 on rows of `width` >= 4 floats (utils/bbox_ops.py:309-316): columns 0..3 become
// (cx - w/2, cy - h/2, cx + w/2, cy + h/2) in float32, one division and one add / subtract each, in that order;
// further columns (the angle of a rotated box) are carried over unchanged.  One thread per row.
__global__ __launch_bounds__(256) void to_corners_kernel(const float *in, float *out, int64_t n, int width) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float *p = in + i * width;
    float *o = out + i * width;
    const float cx = p[0], cy = p[1], hw = p[2] / 2.0f, hh = p[3] / 2.0f;
    o[0] = cx - hw;
    o[1] = cy - hh;
    o[2] = cx + hw;
    o[3] = cy + hh;
    for (int j = 4; j < width; ++j) o[j] = p[j];
}

// Batched forms over fixed-size detection records (one row group of K slots per image, `count` valid):
//   to_original: bboxes_to_original_ with one pad_info row per image (api/detection.py:173-174 inside the per-image loop)
//   to_json:     the arithmetic of ImageObjects.to_json (utils/structures.py:221-259): Python floats, i.e. DOUBLES of the
//                float32 values: [cx - w/2, cy - h/2, w, h], float(score); category through an optional id table
__global__ __launch_bounds__(256) void to_original_batched_kernel(float *bbox, int64_t bbox_st, const int32_t *count,
                                                                  int64_t count_st, int K, const float *pad, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int64_t b = i / K;
    const int k = (int)(i - b * K);
    if (k >= count[b * count_st]) return;
    const float *pi = pad + b * 6;                    // (ori w, ori h, tl x, tl y, imw, imh)
    float *p = bbox + b * bbox_st + (int64_t)k * 4;
    f32x4 v = *reinterpret_cast<f32x4 *>(p);
    v[0] = (v[0] - pi[2]) / pi[4] * pi[0];
    v[1] = (v[1] - pi[3]) / pi[5] * pi[1];
    v[2] = v[2] / pi[4] * pi[0];
    v[3] = v[3] / pi[5] * pi[1];
    *reinterpret_cast<f32x4 *>(p) = v;
}

__global__ __launch_bounds__(256) void to_json_kernel(const float *bbox, int64_t bbox_st, const float *score, int64_t score_st,
                                                      const int64_t *cls, int64_t cls_st, const int32_t *count,
                                                      int64_t count_st, int K, const int64_t *cat_table, int n_cat,
                                                      double *out, int64_t *out_cat, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int64_t b = i / K;
    const int k = (int)(i - b * K);
    double r[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    int64_t cat = 0;
    if (!count || k < count[b * count_st]) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(bbox + b * bbox_st + (int64_t)k * 4);
        const double cx = (double)v[0], cy = (double)v[1], w = (double)v[2], h = (double)v[3];
        r[0] = cx - w / 2.0; r[1] = cy - h / 2.0; r[2] = w; r[3] = h;
        r[4] = (double)score[b * score_st + k];
        const int64_t c = cls[b * cls_st + k];
        cat = cat_table ? ((c >= 0 && c < n_cat) ? cat_table[c] : -1) : c;
    }
#pragma unroll
    for (int j = 0; j < 5; ++j) out[i * 5 + j] = r[j];
    out_cat[i] = cat;
}

}  // namespace

extern "C" int mydet_abi_version(void) { return MYDET_ABI_VERSION; }

extern "C" int mydet_bboxes_to_original_batched_f32(float *bbox, int64_t bbox_stride, const int32_t *count,
                                                    int64_t count_stride, int B, int K, const float *pad_info,
                                                    void *stream) {
    if (B <= 0 || K <= 0 || !bbox || !count || !pad_info || ((uintptr_t)bbox & 15) || (bbox_stride & 3)) return MYDET_E_BADARG;
    const int64_t total = (int64_t)B * K;
    hipLaunchKernelGGL(to_original_batched_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       bbox, bbox_stride, count, count_stride, K, pad_info, total);
    return mydet_launch_status();
}

extern "C" int mydet_detections_to_json_f64(const float *bbox, int64_t bbox_stride, const float *score,
                                            int64_t score_stride, const int64_t *cls, int64_t cls_stride,
                                            const int32_t *count, int64_t count_stride, int B, int K,
                                            const int64_t *cat_table, int n_cat, double *out, int64_t *out_cat,
                                            void *stream) {
    if (B <= 0 || K < 0) return MYDET_E_BADARG;
    if (K == 0) return 0;
    if (!bbox || !score || !cls || !out || !out_cat || ((uintptr_t)bbox & 15) || (bbox_stride & 3)) return MYDET_E_BADARG;
    if (cat_table && n_cat <= 0) return MYDET_E_BADARG;
    const int64_t total = (int64_t)B * K;
    hipLaunchKernelGGL(to_json_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, bbox,
                       bbox_stride, score, score_stride, cls, cls_stride, count, count_stride, K, cat_table, n_cat, out,
                       out_cat, total);
    return mydet_launch_status();
}

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

extern "C" int mydet_cxcywh_to_x1y1x2y2_f32(const float *cxcywh, float *x1y1x2y2, int64_t n, int width, void *stream) {
    if (n < 0 || width < 4) return MYDET_E_BADARG;
    if (n == 0) return 0;
    if (!cxcywh || !x1y1x2y2) return MYDET_E_BADARG;
    if ((n + 255) / 256 > 0x7fffffff) return MYDET_E_UNSUPP;
    hipLaunchKernelGGL(to_corners_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, cxcywh,
                       x1y1x2y2, n, width);
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


// ---------------------------------------------------------------------------------------------------------
// Device-side bilinear resize of an 8-bit RGB image, bit-exact with PIL.Image.resize(size, BILINEAR) -- what the
// reference's tvf.resize does to a PIL image (utils/image_ops.py:22-35, :55-137; api/detection.py:177-205).
// Pillow's separable two-pass filter in its 8-bit fixed-point form: horizontal pass to uint8 first
// (out = clip8((2^21 + sum in * w) >> 22), weights 22-bit integers built on the host by Pillow's own rule,
// mydetection_amd/utils/image_ops.py:resample_tables), then the vertical pass on those uint8 values.  One thread per
// output pixel; the horizontal results it needs (one per vertical tap) are recomputed in registers, so there is no
// intermediate image.  A pass whose size does not change is skipped (NULL tables), as in Pillow.
namespace {
struct ResizeArgs {
    const unsigned char *src;
    unsigned char *dst;
    int64_t src_row, dst_row;                  // bytes between rows
    int H, W, oh, ow, ksx, ksy;
    const int32_t *bx, *kx, *by, *ky;          // bounds [o][2] = (first tap, taps), weights [o][ks]
};

__device__ __forceinline__ int clip8(int v) {
    v >>= 22;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

__global__ __launch_bounds__(256) void resize_bilinear_kernel(const ResizeArgs p) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= p.oh * p.ow) return;
    const int y = i / p.ow, x = i - y * p.ow;
    const int y0 = p.by ? p.by[2 * y] : y, ny = p.by ? p.by[2 * y + 1] : 1;
    const int x0 = p.bx ? p.bx[2 * x] : x, nx = p.bx ? p.bx[2 * x + 1] : 1;
    int acc[3] = {1 << 21, 1 << 21, 1 << 21};
    int last[3] = {0, 0, 0};
    for (int j = 0; j < ny; ++j) {
        const unsigned char *row = p.src + (int64_t)(y0 + j) * p.src_row;
        int h[3];
        if (p.bx) {
            int a[3] = {1 << 21, 1 << 21, 1 << 21};
            for (int t = 0; t < nx; ++t) {
                const int w = p.kx[x * p.ksx + t];
                const unsigned char *px = row + (x0 + t) * 3;
                a[0] += px[0] * w; a[1] += px[1] * w; a[2] += px[2] * w;
            }
            h[0] = clip8(a[0]); h[1] = clip8(a[1]); h[2] = clip8(a[2]);
        } else {
            const unsigned char *px = row + x * 3;
            h[0] = px[0]; h[1] = px[1]; h[2] = px[2];
        }
        if (p.by) {
            const int w = p.ky[y * p.ksy + j];
            acc[0] += h[0] * w; acc[1] += h[1] * w; acc[2] += h[2] * w;
        }
        last[0] = h[0]; last[1] = h[1]; last[2] = h[2];
    }
    unsigned char *o = p.dst + (int64_t)y * p.dst_row + x * 3;
    if (p.by) {
        o[0] = (unsigned char)clip8(acc[0]); o[1] = (unsigned char)clip8(acc[1]); o[2] = (unsigned char)clip8(acc[2]);
    } else {
        o[0] = (unsigned char)last[0]; o[1] = (unsigned char)last[1]; o[2] = (unsigned char)last[2];
    }
}
}  // namespace

extern "C" int mydet_resize_bilinear_u8(const unsigned char *src, int H, int W, int64_t src_row_bytes,
                                        unsigned char *dst, int oh, int ow, int64_t dst_row_bytes,
                                        const int32_t *bounds_x, const int32_t *kx, int ksx, const int32_t *bounds_y,
                                        const int32_t *ky, int ksy, void *stream) {
    if (!src || !dst || H <= 0 || W <= 0 || oh <= 0 || ow <= 0 || src_row_bytes < (int64_t)W * 3 || dst_row_bytes < (int64_t)ow * 3)
        return MYDET_E_BADARG;
    if ((bounds_x == nullptr) != (kx == nullptr) || (bounds_y == nullptr) != (ky == nullptr)) return MYDET_E_BADARG;
    if ((!bounds_x && W != ow) || (!bounds_y && H != oh) || (bounds_x && ksx <= 0) || (bounds_y && ksy <= 0)) return MYDET_E_BADARG;
    if ((int64_t)oh * ow > 0x7fffffff) return MYDET_E_UNSUPP;
    ResizeArgs p;
    p.src = src; p.dst = dst; p.src_row = src_row_bytes; p.dst_row = dst_row_bytes; p.H = H; p.W = W; p.oh = oh; p.ow = ow;
    p.ksx = ksx; p.ksy = ksy; p.bx = bounds_x; p.kx = kx; p.by = bounds_y; p.ky = ky;
    hipLaunchKernelGGL(resize_bilinear_kernel, dim3((unsigned)(((int64_t)oh * ow + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, p);
    return mydet_launch_status();
}
