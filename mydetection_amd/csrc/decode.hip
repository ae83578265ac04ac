// Box decode of one pyramid level: raw head logits -> (cxcywh box, class index, score).
//
// HBM-bound (reads every head logit once: 8.57 MB/image for YOLOv3-80 @640^2, writes
// 28 B per candidate).  A wave owns one pixel at a time: its logits (all anchors) are
// pulled with 16-byte lane loads into a wave-private LDS strip, then for each anchor
// the 64 lanes take the class logits (<= 2 per lane for 80 classes), apply the
// logistic, and a 6-step butterfly picks max / first-argmax exactly as torch.max does
// on the sigmoid values; lane a then finishes anchor a's box arithmetic (all anchors of the
// pixel in one pass) in the reference's operation order (compiled with -ffp-contract=off so no product is fused into a sum).
//
//   YOLO   models/detlayers/yolov3.py:41-69    cx=(s(tx)+x)*stride, w=exp(tw)*aw, score=s(conf)*max s(cls)
//   RETINA models/detlayers/retinanet.py:63-82 cx=acx+tx*aw, w=exp(tw)*aw, clamp [1,max(H,W)], score=max s(cls)
//   FCOS   models/detlayers/fcos2.py:222-251   ltrb=exp(t)*stride, clamp to image, score=sqrt(s(conf)*max s(cls))
#include "common.h"

namespace {

constexpr int MAX_A = 16;
constexpr int WAVES = 4;

struct DecodeArgs {
    int mode;
    const float *box, *cls;
    int64_t ldbox, ldcls;
    int box_astride, box_c0, cls_astride, cls_c0, conf_c0;
    int A, C, H, W, img_h, img_w;
    int box_span, cls_span;      // floats of a pixel actually needed (multiple of 4)
    int same;                    // box and cls are the same tensor
    float stride;
    float aw[MAX_A], ah[MAX_A];
    float *bbox;
    int64_t *cidx;
    float *score;
    int64_t N, n_off, npix;
};

__global__ __launch_bounds__(64 * WAVES) void decode_kernel(const DecodeArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ float s_aw[MAX_A], s_ah[MAX_A];          // anchors, indexed per lane below
#pragma unroll
    for (int a = 0; a < MAX_A; ++a)
        if (threadIdx.x == a) { s_aw[a] = p.aw[a]; s_ah[a] = p.ah[a]; }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int strip = p.cls_span + (p.same ? 0 : p.box_span);
    float *lc = smem + wave * strip;
    float *lb = p.same ? lc : lc + p.cls_span;
    const int64_t nwaves = (int64_t)gridDim.x * WAVES;
    const int hw = p.H * p.W;
    const float fmaxhw = (float)(p.img_h > p.img_w ? p.img_h : p.img_w);

    for (int64_t pix = (int64_t)blockIdx.x * WAVES + wave; pix < p.npix; pix += nwaves) {
        const float *gc = p.cls + pix * p.ldcls;
        for (int i = lane * 4; i < p.cls_span; i += 256)
            *reinterpret_cast<f32x4 *>(lc + i) = *reinterpret_cast<const f32x4 *>(gc + i);
        if (!p.same) {
            const float *gb = p.box + pix * p.ldbox;
            for (int i = lane * 4; i < p.box_span; i += 256)
                *reinterpret_cast<f32x4 *>(lb + i) = *reinterpret_cast<const f32x4 *>(gb + i);
        }
        __builtin_amdgcn_wave_barrier();
        const int b = (int)(pix / hw);
        const int rem = (int)(pix - (int64_t)b * hw);
        const int gy = rem / p.W, gx = rem - gy * p.W;

        // class max / first-argmax per anchor; lane a keeps anchor a's result
        float mybest = 0.0f;
        int mybi = 0;
        for (int a = 0; a < p.A; ++a) {
            const float *cl = lc + a * p.cls_astride + p.cls_c0;
            float best = -1.0f;
            int bi = 0x7fffffff;
            if (lane < p.C) { best = mydet_sigmoid(cl[lane]); bi = lane; }
            if (lane + 64 < p.C) {
                const float s1 = mydet_sigmoid(cl[lane + 64]);
                if (s1 > best) { best = s1; bi = lane + 64; }
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const float ob = __shfl_xor(best, off);
                const int oi = __shfl_xor(bi, off);
                if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
            }
            if (lane == a) { mybest = best; mybi = bi; }
        }
        // box arithmetic: lane a finishes anchor a (all anchors of the pixel in one pass)
        if (lane < p.A) {
            const int a = lane;
            const float *t = lb + a * p.box_astride + p.box_c0;
            const float t0 = t[0], t1 = t[1], t2 = t[2], t3 = t[3];
            const float aw = s_aw[a], ah = s_ah[a];
            f32x4 o;
            float sc;
            if (p.mode == MYDET_DECODE_YOLO) {
                o[0] = (mydet_sigmoid(t0) + (float)gx) * p.stride;
                o[1] = (mydet_sigmoid(t1) + (float)gy) * p.stride;
                o[2] = expf(t2) * aw;
                o[3] = expf(t3) * ah;
                sc = mydet_sigmoid(lc[a * p.cls_astride + p.conf_c0]) * mybest;
            } else if (p.mode == MYDET_DECODE_RETINA) {
                const float acx = p.stride * 0.5f + (float)gx * p.stride;
                const float acy = p.stride * 0.5f + (float)gy * p.stride;
                o[0] = acx + t0 * aw;
                o[1] = acy + t1 * ah;
                o[2] = expf(t2) * aw;
                o[3] = expf(t3) * ah;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = fminf(fmaxf(o[j], 1.0f), fmaxhw);
                sc = mybest;
            } else {
                const float cx = (float)gx * p.stride + p.stride * 0.5f;
                const float cy = (float)gy * p.stride + p.stride * 0.5f;
                const float fw = (float)p.img_w, fh = (float)p.img_h;
                const float x1 = fminf(fmaxf(cx - expf(t0) * p.stride, 0.0f), fw);
                const float y1 = fminf(fmaxf(cy - expf(t1) * p.stride, 0.0f), fh);
                const float x2 = fminf(fmaxf(cx + expf(t2) * p.stride, 0.0f), fw);
                const float y2 = fminf(fmaxf(cy + expf(t3) * p.stride, 0.0f), fh);
                o[0] = (x1 + x2) / 2.0f;
                o[1] = (y1 + y2) / 2.0f;
                o[2] = x2 - x1;
                o[3] = y2 - y1;
                sc = sqrtf(mydet_sigmoid(lc[a * p.cls_astride + p.conf_c0]) * mybest);
            }
            const int64_t n = (int64_t)b * p.N + p.n_off + ((int64_t)a * p.H + gy) * p.W + gx;
            *reinterpret_cast<f32x4 *>(p.bbox + n * 4) = o;
            p.cidx[n] = (int64_t)mybi;
            p.score[n] = sc;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace

extern "C" int mydet_decode_f32(int mode, const float *box, int64_t ldbox, int box_astride, int box_c0,
                                const float *cls, int64_t ldcls, int cls_astride, int cls_c0, int conf_c0,
                                const float *anchors_wh, int A, int C, int B, int H, int W, float stride,
                                int img_h, int img_w, float *bbox, int64_t *class_idx, float *score, int64_t N,
                                int64_t n_off, void *stream) {
    if (mode < 0 || mode > 2 || !box || !cls || !bbox || !class_idx || !score) return MYDET_E_BADARG;
    if (A <= 0 || A > MAX_A || C <= 0 || C > 128 || B <= 0 || H <= 0 || W <= 0) return MYDET_E_BADARG;
    if ((ldbox & 3) || (ldcls & 3) || ((uintptr_t)box & 15) || ((uintptr_t)cls & 15) || ((uintptr_t)bbox & 15))
        return MYDET_E_BADARG;
    if (mode != MYDET_DECODE_FCOS && !anchors_wh) return MYDET_E_BADARG;
    if (n_off < 0 || n_off + (int64_t)A * H * W > N) return MYDET_E_BADARG;
    DecodeArgs p;
    p.mode = mode; p.box = box; p.cls = cls; p.ldbox = ldbox; p.ldcls = ldcls;
    p.box_astride = box_astride; p.box_c0 = box_c0; p.cls_astride = cls_astride; p.cls_c0 = cls_c0;
    p.conf_c0 = conf_c0; p.A = A; p.C = C; p.H = H; p.W = W; p.img_h = img_h; p.img_w = img_w;
    p.stride = stride; p.bbox = bbox; p.cidx = class_idx; p.score = score; p.N = N; p.n_off = n_off;
    p.npix = (int64_t)B * H * W;
    for (int a = 0; a < MAX_A; ++a) { p.aw[a] = 0.f; p.ah[a] = 0.f; }
    // anchors_wh is a HOST pointer (mydet.h): the pairs travel as kernel arguments.
    if (anchors_wh)
        for (int a = 0; a < A; ++a) { p.aw[a] = anchors_wh[2 * a]; p.ah[a] = anchors_wh[2 * a + 1]; }
    int cls_need = (A - 1) * cls_astride + cls_c0 + C;
    if (mode != MYDET_DECODE_RETINA) {
        const int cneed = (A - 1) * cls_astride + conf_c0 + 1;
        cls_need = cls_need > cneed ? cls_need : cneed;
    }
    int box_need = (A - 1) * box_astride + box_c0 + 4;
    p.same = (box == cls && ldbox == ldcls) ? 1 : 0;
    if (p.same) cls_need = cls_need > box_need ? cls_need : box_need;
    p.cls_span = (cls_need + 3) & ~3;
    p.box_span = (box_need + 3) & ~3;
    if (p.cls_span > ldcls || p.box_span > ldbox) return MYDET_E_BADARG;
    const size_t lds = (size_t)WAVES * (p.cls_span + (p.same ? 0 : p.box_span)) * sizeof(float);
    if (lds > 64 * 1024) return MYDET_E_UNSUPP;
    int64_t blocks = (p.npix + WAVES - 1) / WAVES;
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(decode_kernel, dim3((unsigned)blocks), dim3(64 * WAVES), lds, (hipStream_t)stream, p);
    return mydet_launch_status();
}
