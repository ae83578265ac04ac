// Box decode of one pyramid level: raw head logits -> (cxcywh box, class index, score).
//
// HBM-bound: every head logit is read once (8.57 MB/image for YOLOv3-80 @640^2) and 28 B are written per
// candidate.  A workgroup stages a tile of PIX consecutive pixels (all anchors, all channels) from HBM into LDS
// with coalesced 16-byte loads -- the pixel-major head rows are contiguous in memory -- into rows padded to an
// odd number of floats, so that in the compute phase, where a THREAD owns one candidate (pixel, anchor) and walks
// its 80 class logits, the 64 lanes of a wave (64 different pixels, same channel) hit 64 different banks.
// The class max / first argmax is a sequential strict-greater scan, i.e. torch.max's order: on the logits while
// the float32 logistic keeps them apart, on the sigmoid values themselves once the max logit is >= 5 (near
// saturation distinct logits collapse onto one float32 sigmoid and the reference's "first index among equal
// sigmoids" is decided by those collisions).  Box arithmetic follows the reference's operation order; the file is
// built with -ffp-contract=off so no product is fused into a sum.  Outputs of a wave are 64 consecutive
// candidates: coalesced 16-byte box stores.
//
//   YOLO   models/detlayers/yolov3.py:41-69    cx=(s(tx)+x)*stride, w=exp(tw)*aw, score=s(conf)*max s(cls)
//   RETINA models/detlayers/retinanet.py:63-82 cx=acx+tx*aw, w=exp(tw)*aw, clamp [1,max(H,W)], score=max s(cls)
//   FCOS   models/detlayers/fcos2.py:222-251   ltrb=exp(t)*stride, clamp to image, score=sqrt(s(conf)*max s(cls))
#include <cstdlib>

#include "common.h"

namespace {

constexpr int MAX_A = 16;

struct DecodeArgs {
    int mode;
    const float *box, *cls;
    int64_t ldbox, ldcls;
    int box_astride, box_c0, cls_astride, cls_c0, conf_c0;
    int A, C, H, W, img_h, img_w;
    int box_span, cls_span;      // floats of a pixel actually needed (multiple of 4)
    int same;                    // box and cls are the same tensor
    int PIX, row;                // pixels per tile, LDS row length (odd)
    float stride;
    float aw[MAX_A], ah[MAX_A];
    float *bbox;
    int64_t *cidx;
    float *score;
    int64_t N, n_off, npix;
};

__device__ __forceinline__ void stage_rows(float *sm, int row, int col0, const float *src, int64_t ld, int span,
                                           int npx) {
    const int q = span >> 2;
    for (int i = threadIdx.x; i < npx * q; i += 256) {
        const int r = i / q, c4 = i - r * q;
        const f32x4 v = *reinterpret_cast<const f32x4 *>(src + (int64_t)r * ld + c4 * 4);
        float *d = sm + r * row + col0 + c4 * 4;
        d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
    }
}

__global__ __launch_bounds__(256) void decode_kernel(const DecodeArgs p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ float s_aw[MAX_A], s_ah[MAX_A];
#pragma unroll
    for (int a = 0; a < MAX_A; ++a)
        if (threadIdx.x == a) { s_aw[a] = p.aw[a]; s_ah[a] = p.ah[a]; }
    const int hw = p.H * p.W;
    const float fmaxhw = (float)(p.img_h > p.img_w ? p.img_h : p.img_w);
    const int box_col = p.same ? 0 : p.cls_span;
    const int64_t ntiles = (p.npix + p.PIX - 1) / p.PIX;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t pix0 = tile * p.PIX;
        const int npx = (int)(p.npix - pix0 < p.PIX ? p.npix - pix0 : p.PIX);
        stage_rows(sm, p.row, 0, p.cls + pix0 * p.ldcls, p.ldcls, p.cls_span, npx);
        if (!p.same) stage_rows(sm, p.row, box_col, p.box + pix0 * p.ldbox, p.ldbox, p.box_span, npx);
        __syncthreads();
        for (int c = threadIdx.x; c < p.PIX * p.A; c += 256) {
            const int a = c / p.PIX, px = c - a * p.PIX;
            if (px >= npx) continue;
            const float *rowp = sm + px * p.row;
            const float *cl = rowp + a * p.cls_astride + p.cls_c0;
            // class max / first argmax (strict >: the first maximum wins, as torch.max)
            float best = cl[0];
            int bi = 0;
#pragma unroll 8
            for (int k = 1; k < p.C; ++k) {
                const float v = cl[k];
                if (v > best) { best = v; bi = k; }
            }
            float cmax;
            if (best < 5.0f && best > -80.0f) {
                cmax = mydet_sigmoid(best);
            } else {                                   // near saturation: compare the sigmoid values themselves
                cmax = mydet_sigmoid(cl[0]);
                bi = 0;
                for (int k = 1; k < p.C; ++k) {
                    const float sv = mydet_sigmoid(cl[k]);
                    if (sv > cmax) { cmax = sv; bi = k; }
                }
            }
            const float *t = rowp + box_col + a * p.box_astride + p.box_c0;
            const float t0 = t[0], t1 = t[1], t2 = t[2], t3 = t[3];
            const int64_t pix = pix0 + px;
            const int b = (int)(pix / hw);
            const int rem = (int)(pix - (int64_t)b * hw);
            const int gy = rem / p.W, gx = rem - gy * p.W;
            const float aw = s_aw[a], ah = s_ah[a];
            f32x4 o;
            float sc;
            if (p.mode == MYDET_DECODE_YOLO) {
                o[0] = (mydet_sigmoid(t0) + (float)gx) * p.stride;
                o[1] = (mydet_sigmoid(t1) + (float)gy) * p.stride;
                o[2] = expf(t2) * aw;
                o[3] = expf(t3) * ah;
                sc = mydet_sigmoid(rowp[a * p.cls_astride + p.conf_c0]) * cmax;
            } else if (p.mode == MYDET_DECODE_RETINA) {
                const float acx = p.stride * 0.5f + (float)gx * p.stride;
                const float acy = p.stride * 0.5f + (float)gy * p.stride;
                o[0] = acx + t0 * aw;
                o[1] = acy + t1 * ah;
                o[2] = expf(t2) * aw;
                o[3] = expf(t3) * ah;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = fminf(fmaxf(o[j], 1.0f), fmaxhw);
                sc = cmax;
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
                sc = sqrtf(mydet_sigmoid(rowp[a * p.cls_astride + p.conf_c0]) * cmax);
            }
            const int64_t n = (int64_t)b * p.N + p.n_off + ((int64_t)a * p.H + gy) * p.W + gx;
            *reinterpret_cast<f32x4 *>(p.bbox + n * 4) = o;
            p.cidx[n] = (int64_t)bi;
            p.score[n] = sc;
        }
        __syncthreads();
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
    p.row = (p.cls_span + (p.same ? 0 : p.box_span)) | 1;               // odd row length: conflict-free column walks
    p.PIX = 32;                                                        // 32-pixel tiles: 33 KB (YOLO), 4 workgroups per CU overlap load and compute
    if (const char *e = getenv("MYDET_DECODE_PIX")) p.PIX = atoi(e);      // tuning knob
    while (p.PIX > 4 && (size_t)p.PIX * p.row * sizeof(float) > 80 * 1024) p.PIX >>= 1;
    const size_t lds = (size_t)p.PIX * p.row * sizeof(float);
    if (lds > 80 * 1024) return MYDET_E_UNSUPP;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&decode_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  80 * 1024);
        attr_set = true;
    }
    int64_t blocks = (p.npix + p.PIX - 1) / p.PIX;
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(decode_kernel, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, p);
    return mydet_launch_status();
}
