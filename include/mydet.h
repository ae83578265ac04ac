/* mydet.h -- C ABI of libmydet_hip.so: the MI355X (gfx950) kernels behind the
 * single-stage detection inference hot path of duanzhiihao/myDetection.
 *
 * The reference has no FFI seam (it is pure Python over ATen/torchvision); its
 * seams are the Python plug-in factories models/registry.py:4-146 and
 * api/detection.py:19-205.  Each entry point below replaces the third-party native
 * op (ATen / torchvision) that a reference Python call site lands in; the call
 * site is cited per function.  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *  - Plain pointers and sizes only.  All pointers are DEVICE pointers (HBM).
 *  - `stream` is a hipStream_t passed as void*; 0 = the null stream.
 *  - Every function only enqueues work on `stream`; it never allocates persistent
 *    memory, never synchronises, and is graph-capturable.  The caller owns all
 *    buffers, including scratch.
 *  - Return value: 0 on success, otherwise a hipError_t, or a negative MYDET_E_*
 *    code for argument errors detected on the host before any launch.
 *  - Activations are float32, channels-last ("NHWC"): element (b,y,x,c) of a
 *    tensor with pixel stride `ld` (in floats, ld >= C) lives at
 *    ((b*H + y)*W + x)*ld + c.  A pixel stride larger than C lets a kernel read
 *    or write a channel slice of a wider buffer (free concat / padding).
 */
#ifndef MYDET_H
#define MYDET_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MYDET_ABI_VERSION 2      /* 2 (round 6): mydet_se_tail gained hpart_bytes; mydet_conv3x3_p3_f32 added */

#define MYDET_E_BADARG   (-1)   /* shape/stride/alignment precondition violated */
#define MYDET_E_UNSUPP   (-2)   /* valid request this build has no kernel for   */

/* activation codes for the conv epilogue */
#define MYDET_ACT_NONE   0
#define MYDET_ACT_LEAKY  1      /* LeakyReLU(0.1): models/modules.py:92        */
#define MYDET_ACT_SWISH  2      /* x*sigmoid(x):  models/modules.py:41-43      */

int mydet_abi_version(void);

/* Dense convolution as implicit GEMM on FP32 MFMA (v_mfma_f32_32x32x2_f32), with the
 * epilogue  y = act(acc*scale[n] + shift[n]) + residual  fused.
 * Replaces the conv2d -> batch_norm -> leaky_relu (-> add) ATen chain of
 *   ConvBnLeaky.forward  models/modules.py:94-95   (scale/shift = folded BN, eps 1e-5)
 *   DarkBlock.forward    models/modules.py:69-73   (residual != NULL)
 *   YOLOHead 1x1 convs   models/rpns.py:24-25      (scale NULL => 1, shift = bias)
 * x  : [B,H,W,Cin] pixel stride ldx.   Cin % 4 == 0, ldx % 4 == 0, 16-byte aligned.
 * w  : [Cout][KH][KW][Cin] contiguous (OHWI repack of the reference's OIHW weight).
 * y  : [B,Ho,Wo,Cout] pixel stride ldy.  residual: same shape, pixel stride ldr, or NULL.
 * Ho = (H + pad_t + pad_b - KH)/stride + 1 (likewise Wo); only pad_t/pad_l are
 * needed by the kernel, Ho/Wo are passed explicitly (asymmetric "SAME" pads allowed).
 * a_gate: NULL, or [B][Cin] per-image channel multipliers applied to x while it is staged
 *   (1x1 stride-1 convs without activation only): the squeeze-excite scale
 *   `torch.sigmoid(x_squeezed) * x` feeding `_project_conv`, external/efficientnet/model.py:83-85,
 *   without a pass over the expanded tensor.
 * workspace: NULL, or scratch (16-byte aligned) for the split-K tail: when the grid is a few full rounds of the
 *   chip plus a small remainder, the remainder tiles are cut along K, partial tiles go here and a fixup launch
 *   sums them in a fixed order (deterministic); 64 MiB covers every layer of the three models.
 * Also covers: MBConv expand/project convs (external/efficientnet/model.py:75,85), BiFPN
 * input projections (models/fpns.py:446-448), SeparableConv2d.pointwise (models/modules.py:20),
 * C6/C7 convs (models/backbones.py:183-200), dense cls_last conv (models/rpns.py:155-158).
 */
int mydet_conv2d_igemm_f32(const float *x, int64_t ldx, const float *w,
                           const float *scale, const float *shift,
                           const float *residual, int64_t ldr, const float *a_gate,
                           void *workspace, int64_t workspace_bytes,
                           float *y, int64_t ldy,
                           int B, int H, int W, int Cin, int Cout,
                           int KH, int KW, int stride, int pad_t, int pad_l,
                           int Ho, int Wo, int act, void *stream);

/* The same fused conv + epilogue for KH = KW = 3, stride 1, pad 1 (Ho = H, Wo = W) by Winograd F(2x2,3x3):
 * 16 transform-domain GEMMs on FP32 MFMA with the input/output transforms fused into staging and epilogue
 * (2.25x fewer multiplies; float32 throughout, only the association order of the sums differs from the direct
 * form).  Call sites replaced: the 3x3 ConvBnLeaky of Darknet53 / DarkBlock / YOLOBranch
 * (models/modules.py:69-73,94-95; models/backbones.py:14-41; models/fpns.py:38-47) and the dense 3x3 convs of
 * models/backbones.py:183-200, models/rpns.py:155-158.
 * u: weights in the transform domain, produced once per layer by mydet_wino_weights_f32 from the OHWI weight
 *    (mydet_wino_weights_floats(Cout, Cin) floats; 0 if the shape is unsupported).
 * workspace: NULL, or 16-byte aligned scratch (32 MiB suffices): grids of two or more resident rounds then run a
 *    stream-K schedule -- one round of persistent workgroups with equal shares of the K iterations; items whose K
 *    was cut leave partial outputs here and a fixup launch sums them in K order (deterministic).
 * Needs Cin % 8 == 0, Cout % 4 == 0, ldy % 4 == 0 (ldr % 4 == 0), 16-byte aligned pointers; otherwise
 * MYDET_E_UNSUPP and the caller uses mydet_conv2d_igemm_f32.
 */
int64_t mydet_wino_weights_floats(int Cout, int Cin);
int mydet_wino_weights_f32(const float *w_ohwi, int Cout, int Cin, float *u, void *stream);
int mydet_conv2d_wino_f32(const float *x, int64_t ldx, const float *u, const float *scale, const float *shift,
                          const float *residual, int64_t ldr, void *workspace, int64_t workspace_bytes, float *y,
                          int64_t ldy, int B, int H, int W, int Cin, int Cout, int act, void *stream);

/* First-layer convolution (Cin == 3, 3x3) reading the image with arbitrary strides
 * (NCHW as handed over by api/detection.py:160-166, or channels-last) and writing NHWC.
 * Replaces netlist[0] of Darknet53 (models/backbones.py:14) and the EfficientNet stem.
 * Cout must be 32.  Strides sxb/sxc/sxh/sxw in floats.
 */
int mydet_conv2d_stem_f32(const float *x, int64_t sxb, int64_t sxc, int64_t sxh, int64_t sxw,
                          const float *w /* [Cout][3][3][3] OHWI */,
                          const float *scale, const float *shift,
                          float *y, int64_t ldy,
                          int B, int H, int W, int Cout, int stride, int pad_t, int pad_l,
                          int Ho, int Wo, int act, void *stream);

/* Depthwise K x K convolution (K = 3 or 5, stride 1 or 2), y = act(conv*scale + shift) (scale/shift NULL =>
 * plain conv).  Replaces `_depthwise_conv` + `_bn1` + swish (external/efficientnet/model.py:77) and
 * SeparableConv2d.depthwise (models/modules.py:12-13,19).  w: [K][K][C] (repack of [C,1,K,K]).
 * se_partial != NULL: the launch also writes per-image channel sums of y split over S pixel slices,
 * se_partial[B][S+1][C] (slices 0..S-1; slice S is scratch for mydet_se_gate_f32) -- the squeeze of the following squeeze-excite (adaptive_avg_pool2d,
 * external/efficientnet/model.py:81) without another pass over y; deterministic (no atomics).
 * S must be mydet_dwconv_slices(Ho, Wo, C, K, stride): the number of slices the kernel chosen for the layer writes
 * (one per 8 x 16 output tile for the LDS-tiled stride-1 kernel, which takes the layers of 32 channels and more).
 */
/* Squeeze-excite tail inside the launch that produces the depthwise output (optional last argument of mydet_dwconv_f32,
 * mydet_mbconv_expand_dw_f32, mydet_stem_dw_f32; NULL = none).  The launch then also writes
 *     gate[b][c] = sigmoid(W2 . swish(W1 . mean_pixels(y[b]) + b1) + b2)[c]          (external/efficientnet/model.py:80-83)
 * -- what mydet_se_gate_f32 computes from se_partial in a launch of its own -- without that launch: every workgroup adds its
 * channels' share of W1 . sums while it holds them and publishes it (fire and forget); the last workgroup of an image waits
 * for the shares, sums them in a fixed order and runs the expand conv (csrc/se_tail.h).  Deterministic.
 *   w1 [Cse][C], b1 [Cse], w2t [Cse][C] (the expand conv TRANSPOSED), b2 [C], gate [B][C]  (all 16-byte aligned);
 *   hpart: the share buffer, 8-byte aligned, MYDET_SE_EPOCH_WORDS + 2 * B * groups * Cse 32-bit words (groups =
 *   mydet_dwconv_se_groups / mydet_mbconv_tiles of the layer): a header of MYDET_SE_EPOCH_WORDS words -- word 0 the launch
 *   counter, 1 when the buffer is made; word 1 the count of finished images, 0; word 2 the number of finishing workgroups that
 *   ever gave up waiting for a share and wrote a NaN gate (0 in a healthy run; never reset by the launches: a caller may poll it);
 *   the rest unused -- then (value, epoch) pairs, zero when the buffer is made.  hpart_bytes = the buffer's size: an entry point
 *   given a smaller one than its layer needs returns MYDET_E_BADARG and launches nothing.  After that only the launches touch it; ONE buffer serves every layer and batch size of a
 *   stream, but never two streams at a time.  Cse <= 96; MYDET_E_UNSUPP beyond.  se_partial may be NULL when the tail is given. */
#define MYDET_SE_EPOCH_WORDS 1024
typedef struct {
    const float *w1, *b1, *w2t, *b2;
    float *gate, *hpart;
    int Cse;
    int64_t hpart_bytes;      /* size of hpart: >= 4 * (MYDET_SE_EPOCH_WORDS + 2 * B * groups * Cse), checked by every entry point */
} mydet_se_tail;
int mydet_dwconv_slices(int Ho, int Wo, int C, int K, int stride);
/* workgroups per image of the kernel mydet_dwconv_f32 picks for the layer when it also emits the squeeze (sizes se->hpart) */
int mydet_dwconv_se_groups(int Ho, int Wo, int C, int K, int stride);
int mydet_dwconv_f32(const float *x, int64_t ldx, const float *w, const float *scale, const float *shift,
                     float *y, int64_t ldy, int B, int H, int W, int C, int K, int stride, int pad_t, int pad_l,
                     int Ho, int Wo, int act, float *se_partial, int S, const mydet_se_tail *se, void *stream);

/* Per-image channel sums of x split over S pixel slices: partial[B][S+1][C], slices 0..S-1 (standalone squeeze). */
int mydet_channel_sums_f32(const float *x, int64_t ldx, int B, int H, int W, int C, float *partial, int S,
                           void *stream);

/* Squeeze-excite gate from the partial sums partial[B][S+1][C]: mean = sum_{s<S} partial[b][s][:] / HW (stored
 * in slice S);
 * gate[b][c] = sigmoid(W2 . swish(W1 . mean + b1) + b2).  Replaces adaptive_avg_pool2d + _se_reduce + swish +
 * _se_expand + sigmoid (external/efficientnet/model.py:80-83).  w1 [Cse][C]; w2t [Cse][C] = _se_expand weight
 * transposed.
 */
int mydet_se_gate_f32(float *partial, int S, int B, int HW, int C, const float *w1, const float *b1, int Cse,
                      const float *w2t, const float *b2, float *gate, void *stream);

/* 3x3 stride-2 pad-1 max pool (-inf padding): nn.MaxPool2d(3, 2, 1) models/backbones.py:186,188,
 * tnf.max_pool2d models/fpns.py:405-416. */
int mydet_maxpool3s2_f32(const float *x, int64_t ldx, float *y, int64_t ldy, int B, int H, int W, int C,
                         int Ho, int Wo, void *stream);

/* BiFPN node input: y = swish(sum_i w_i * in_i), w = relu(weights) / (sum(relu(weights)) + 1e-4)
 * (LinearFusion.forward models/fpns.py:433-438, the part before spconv_bn).  n = 2 or 3 inputs of C
 * channels; mode_i: 0 = [B,H,W] map, 1 = [B,H/2,W/2] map read through nearest 2x upsampling
 * (upsample2x, models/fpns.py:442-444), 2 = [B,2H,2W] map read through max_pool2d(3,2,1). */
int mydet_bifpn_fuse_f32(int n, const float *in0, int64_t ld0, int mode0, const float *in1, int64_t ld1, int mode1,
                         const float *in2, int64_t ld2, int mode2, const float *weights, float *y, int64_t ldy,
                         int B, int H, int W, int C, void *stream);

/* y[b,yo,xo, 0:C1] = a[b, nearest(yo), nearest(xo), :]  ;  y[..., C1:C1+C2] = b[b,yo,xo,:]
 * Replaces F.interpolate(mode='nearest') + torch.cat((pre, x), 1) of
 * YOLOBranch.forward models/fpns.py:62-65.  C1, C2, strides multiples of 4.
 * If C2 == 0 / b == NULL it is a plain nearest resize.
 */
int mydet_upsample_concat_f32(const float *a, int64_t lda, int Ha, int Wa, int C1,
                              const float *b, int64_t ldb, int C2,
                              float *y, int64_t ldy, int B, int Ho, int Wo, void *stream);

/* The same concatenation consumed on the fly by the 1x1 ConvBnLeaky that follows it in YOLOBranch.forward
 * (models/fpns.py:62-66: `x = cat((upsample(pre), x), 1); x = self.cbl_0(x)`):
 *   y = LeakyReLU_0.1((conv1x1(cat((up2x_nearest(x_lo), x_hi), 1)) * scale + shift)
 * x_lo [B,H/2,W/2,ld_lo] (C_lo channels), x_hi [B,H,W,ld_hi] (C_hi channels), w [Cout][C_lo + C_hi] (OHWI, the
 * concatenation's channel order), y [B,H,W,ldy].  The concatenated tensor is never written; the sums run in the same k
 * order and tile shape as mydet_conv2d_igemm_f32 on the materialised tensor: bit-identical results.
 * workspace: as for mydet_conv2d_igemm_f32.  Needs H, W even, C_lo % 32 == 0, C_hi % 32 == 0, act == MYDET_ACT_LEAKY,
 * AND a shape that mydet_conv2d_igemm_f32 itself would run on its 64 x 64 x 32 tile (the only tile this entry point is
 * instantiated for: e.g. 64 < Cout, C_lo + C_hi <= 1024 or fewer than 1024 tiles of 128 x 128 -- YOLOv3's 768->256 and
 * 384->128 layers); otherwise MYDET_E_UNSUPP and the caller uses mydet_upsample_concat_f32 + mydet_conv2d_igemm_f32,
 * so the bit-identity above holds for every shape the call accepts. */
int mydet_conv1x1_upcat_f32(const float *x_lo, int64_t ld_lo, int C_lo, const float *x_hi, int64_t ld_hi, int C_hi,
                            const float *w, const float *scale, const float *shift, void *workspace,
                            int64_t workspace_bytes, float *y, int64_t ldy, int B, int H, int W, int Cout, int act,
                            void *stream);

/* The implicit GEMM of mydet_conv2d_igemm_f32 on the bfloat16 matrix instructions with float32-exact operands: every
 * float32 operand is cut into three bfloat16 pieces (a = a0 + a1 + a2 by round-to-nearest remainders, exact to 2^-27 |a|), a
 * product is the sum of the six piece products of weight >= 2^-18 (each exact in float32), accumulated in float32 by
 * v_mfma_f32_32x32x16_bf16: per product an error of 2^-26 |a b|, below the rounding of a float32 multiply-add -- the same
 * results as the float32 kernel to float32 round-off (tests hold both to 2e-5 * max|y| against float64) at 2.67 x its matrix
 * rate.  Same arguments as mydet_conv2d_igemm_f32 except: w_planes = the OHWI weight [Cout][K = KH*KW*Cin] as three bfloat16
 * planes in the kernels' slab-major, DMA-swizzled order (csrc/conv_igemm.hip: split_bf16_kernel; Cout padded to 256 rows),
 * made ONCE per layer by mydet_split_bf16_f32 into mydet_split_bf16_elems(Cout, K) uint16; a_gate (optional, 1x1 layers without an
 * activation: the squeeze-excite project convs) multiplies the activations per image and channel before they are split.  Cin % 16 == 0 (1x1 layers: Cin % 4 == 0, the
 * last 16-channel slab of the planes zero-filled by mydet_split_bf16_f32);
 * MYDET_E_UNSUPP otherwise (the caller then uses mydet_conv2d_igemm_f32).  Replaces the same reference lines.
 * Finite tensors only: an infinite operand, or a finite one of magnitude > 3.3962e38 (it rounds to a bfloat16 inf), yields NaN where
 * the float32 kernel and the reference yield inf (the split forms inf - inf); NaN propagates as NaN. */
int64_t mydet_split_bf16_elems(int Cout, int K);      /* uint16 elements of the operand below (0: K % 4 != 0) */
int mydet_split_bf16_f32(const float *w, int Cout, int K, uint16_t *planes, void *stream);
int mydet_conv2d_igemm_b3_f32(const float *x, int64_t ldx, const uint16_t *w_planes, const float *scale, const float *shift,
                              const float *residual, int64_t ldr, const float *a_gate, void *workspace, int64_t workspace_bytes, float *y,
                              int64_t ldy, int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad_t,
                              int pad_l, int Ho, int Wo, int act, void *stream);
/* 3x3 convolution, pad 1, stride 1 or 2, on the bfloat16 matrix instructions with the float32-exact split operands of
 * mydet_conv2d_igemm_b3_f32 -- same w_planes, same six piece products, float32 accumulation -- but with the workgroup's INPUT
 * PATCH resident in LDS (csrc/conv_p3.hip): a workgroup owns 8 x 16 output pixels (stride 2: 16 x 8 / 32 x 4 for a remainder of 8 / 4
 * columns) x (64 | 128) output channels, loads and
 * splits the patch of a 16-channel slab once and the nine taps read their matrix operands out of it at tap offsets, instead of
 * gathering and splitting every input element once per tap (2.25 x at stride 2, 9 x at stride 1).  K order (slab, tap) instead of
 * (tap, slab): results equal the other kernels' to float32 round-off (tests: 2e-5 * max|y| against float64).
 *   y = act((conv3x3(x) * scale + shift)) + residual,  Ho = (H - 1) / stride + 1, Wo likewise;  act: MYDET_ACT_NONE | _LEAKY.
 * Cin % 16 == 0, ldx % 4 == 0; MYDET_E_UNSUPP otherwise (the caller then uses mydet_conv2d_igemm_b3_f32 / _igemm_f32).
 * Replaces the ATen chain of models/modules.py:76-95 for the stride-2 ConvBnLeaky layers of models/backbones.py:14-30 and the
 * 32 -> 64 layer of the first DarkBlock (models/modules.py:56-73). */
int mydet_conv3x3_p3_f32(const float *x, int64_t ldx, const uint16_t *w_planes, const float *scale, const float *shift,
                         const float *residual, int64_t ldr, float *y, int64_t ldy, int B, int H, int W, int Cin, int Cout,
                         int stride, int act, void *stream);
/* Test hook: the split-bf16 launcher reads MYDET_B3_WIDE / MYDET_B3_WAVES once per process; this reads them again.
 * Returns the form bits (1 = wide 128 x 256 tiles from 192 output channels, 2 = 8-wave workgroups). */
int mydet_conv_b3_reload_tuning(void);

/* Test / tuning hook: workgroups per CU the runtime reports (hipOccupancyMaxActiveBlocksPerMultiprocessor) for the
 * base instance of conv_igemm tile configuration `cfg` (0, 1, 2, 3, 6, 8, 9); *assumed = the count the launch rule
 * computes its rounds with.  Returns the count or a negative MYDET_E_*.  No reference counterpart. */
int mydet_conv_igemm_occupancy(int cfg, int *assumed);

/* Focus.forward of the Ultralytics backbone (external/ultralytics/common.py:79-86): 2x2 space-to-depth,
 *   y[b, yo, xo, g*C + c] = x[b, c, 2*yo + dy, 2*xo + dx],  g = 0:(dy 0, dx 0) 1:(dy 1, dx 0) 2:(dy 0, dx 1) 3:(dy 1, dx 1)
 * -- the channel order of torch.cat([x[..., ::2, ::2], x[..., 1::2, ::2], x[..., ::2, 1::2], x[..., 1::2, 1::2]], 1).
 * x: logical [B,C,H,W] read through its element strides (sb, sc, sh, sw), H and W even; y: channels-last
 * [B,H/2,W/2,ldy], ldy >= 4C, ldy % 4 == 0.  The 3x3 conv that follows is mydet_conv2d_igemm_f32 on 4C channels. */
int mydet_space_to_depth_f32(const float *x, int64_t sb, int64_t sc, int64_t sh, int64_t sw, float *y, int64_t ldy,
                             int B, int C, int H, int W, void *stream);

/* SPP.forward of the Ultralytics backbone (external/ultralytics/common.py:59-70), the part between its two convs:
 *   y[..., 0:C] = x,  y[..., C:2C] = maxpool_k0(x),  y[..., 2C:3C] = maxpool_k1(x),  y[..., 3C:4C] = maxpool_k2(x)
 * with nn.MaxPool2d(kernel_size=k, stride=1, padding=k//2) (-inf padding); k0 <= k1 <= k2 odd (5, 9, 13).
 * x [B,H,W,ldx], y [B,H,W,ldy], C % 4 == 0, ldy >= 4C. */
int mydet_spp_concat_f32(const float *x, int64_t ldx, float *y, int64_t ldy, int B, int H, int W, int C, int k0, int k1,
                         int k2, void *stream);

/* Box decode of one pyramid level: raw head logits -> (bbox cxcywh, class_idx, score)
 * for every candidate, written into the level's slice [n_off, n_off + A*H*W) of the
 * per-image candidate arrays (models/general.py:74-76 concatenates levels along dim 1).
 *   mode MYDET_DECODE_YOLO   YOLOLayer.forward        models/detlayers/yolov3.py:41-69
 *   mode MYDET_DECODE_RETINA RetinaLayer.forward      models/detlayers/retinanet.py:63-82
 *   mode MYDET_DECODE_FCOS   FCOS_ATSS_Layer.forward  models/detlayers/fcos2.py:222-251
 * box : [B,H,W,*] pixel stride ldbox; anchor a's 4 box logits at a*box_astride + box_c0
 * cls : [B,H,W,*] pixel stride ldcls; anchor a's C class logits at a*cls_astride + cls_c0,
 *       its objectness/centerness logit (YOLO, FCOS) at a*cls_astride + conf_c0.
 *       (YOLO head: box == cls, astride 5+C, box_c0 0, conf_c0 4, cls_c0 5.)
 * anchors_wh: HOST pointer (the one exception to "device pointers only") to A pairs (w,h)
 *       in pixels (A <= 16); the pairs travel as kernel arguments so the launch stays
 *       graph-capturable (YOLO, RETINA; ignored for FCOS, where A == 1).
 * Candidate order inside a level is (a, y, x), x fastest.  N = candidates per image.
 * Outputs: bbox [B,N,4] f32, class_idx [B,N] i64, score [B,N] f32.
 * ldbox, ldcls multiples of 4; 16-byte aligned bases.
 */
#define MYDET_DECODE_YOLO   0
#define MYDET_DECODE_RETINA 1
#define MYDET_DECODE_FCOS   2
typedef struct mydet_decode_level {
    const float *box; int64_t ldbox;     /* device */
    const float *cls; int64_t ldcls;     /* device */
    const float *anchors_wh;             /* HOST, A pairs, or NULL (FCOS) */
    int H, W;
    float stride;
    int64_t n_off;                       /* first candidate of this level inside [0, N) */
} mydet_decode_level;
/* All pyramid levels in one launch (the per-level loop of models/general.py:69-72 + the torch.cat of :74-76);
 * `levels` is a HOST array of nlevels (<= 5) descriptors; the other arguments as for mydet_decode_f32. */
int mydet_decode_levels_f32(int mode, int nlevels, const mydet_decode_level *levels, int box_astride, int box_c0,
                            int cls_astride, int cls_c0, int conf_c0, int A, int C, int B, int img_h, int img_w,
                            float *bbox, int64_t *class_idx, float *score, int64_t N, void *stream);
int mydet_decode_f32(int mode,
                     const float *box, int64_t ldbox, int box_astride, int box_c0,
                     const float *cls, int64_t ldcls, int cls_astride, int cls_c0, int conf_c0,
                     const float *anchors_wh, int A, int C,
                     int B, int H, int W, float stride, int img_h, int img_w,
                     float *bbox, int64_t *class_idx, float *score, int64_t N, int64_t n_off,
                     void *stream);

/* Batched confidence filter -> top-k -> class-aware greedy NMS, one image per workgroup.
 * Replaces ImageObjects.post_process / non_max_suppression (utils/structures.py:92-173)
 * and torchvision.ops.nms; the candidates never leave HBM.
 *   keep score >= conf_thres (float32 compare); if more than `topk` (<=512) pass, keep the
 *   topk highest (ties: lowest candidate index); per class ascending: greedy NMS on
 *   x1y1x2y2 = (cx-w/2, cy-h/2, cx+w/2, cy+h/2), suppress when (double)IoU > nms_thres;
 *   survivors ordered class ascending, score descending (ties: lowest index).
 * In : bbox [B,N,4], class_idx [B,N] i64, score [B,N].   N < 2^20, class ids in [0, 2^12): an image in which a
 *      candidate that passes the filter / top-k carries a class id outside that range gets
 *      count = MYDET_COUNT_BAD_CLASS (-1) and all-zero rows instead of silently aliased classes.
 * Out: count [B] i32; out_bbox [B,topk,4]; out_class [B,topk] i64; out_score [B,topk];
 *      out_index [B,topk] i32 = candidate index in [0,N) of each survivor (rows >= count
 *      are zero-filled).
 * scratch: B*N*8 bytes.
 */
int mydet_postprocess_f32(const float *bbox, const int64_t *class_idx, const float *score,
                          int B, int64_t N, float conf_thres, double nms_thres, int topk,
                          int32_t *count, float *out_bbox, int64_t *out_class, float *out_score,
                          int32_t *out_index, void *scratch, void *stream);

/* Same post-processing (topk = 512), written as ONE fixed-size record per image -- the wire format of the
 * multi-GPU exchange (one all-gather of these records; no reference counterpart), so nothing is
 * packed or unpacked between the kernel and the collective.  A record is MYDET_REC_WORDS int32 words
 * (16 400 B, rows 16-byte aligned):
 *   [COUNT] count i32, 3 zero words | [BBOX] 512 x (cx,cy,w,h) f32 | [SCORE] 512 f32 |
 *   [CLASS] 512 i64 | [INDEX] 512 i32          (entries >= count are zero)
 * records: B * MYDET_REC_WORDS words, 16-byte aligned.  scratch: B*N*8 bytes.
 */
#define MYDET_COUNT_BAD_CLASS (-1)
#define MYDET_REC_TOPK   512
#define MYDET_REC_COUNT  0
#define MYDET_REC_BBOX   4
#define MYDET_REC_SCORE  (MYDET_REC_BBOX + 4 * MYDET_REC_TOPK)
#define MYDET_REC_CLASS  (MYDET_REC_SCORE + MYDET_REC_TOPK)
#define MYDET_REC_INDEX  (MYDET_REC_CLASS + 2 * MYDET_REC_TOPK)
#define MYDET_REC_WORDS  (MYDET_REC_INDEX + MYDET_REC_TOPK)
int mydet_postprocess_records_f32(const float *bbox, const int64_t *class_idx, const float *score,
                                  int B, int64_t N, float conf_thres, double nms_thres,
                                  int32_t *records, void *scratch, void *stream);

/* Winograd F(4x4,3x3) form of the same 3x3 stride-1 pad-1 conv + BN + act (+ residual) as mydet_conv2d_wino_f32
 * (4x fewer multiplies than the direct form; used for the deep layers with chip-filling grids).  `u` = the
 * transform-domain weights made by mydet_wino4_weights_f32 from the OHWI weight (mydet_wino4_weights_floats(Cout, Cin)
 * floats; Cin % 4 == 0).  `ws` = device scratch of at least mydet_wino4_workspace_bytes(B, H, W, Cin, Cout) bytes
 * (36 floats per 4x4-output tile and input channel: the transform-domain input, written by a first launch and
 * streamed by the second, plus 64 MiB for the partial tiles of the K-cut tail: when the workgroup count is whole rounds
 * of the chip plus a small remainder, the remainder runs as K pieces that a fourth launch sums in K order -- deterministic);
 * stream-ordered, so one buffer serves every layer of a stream.
 * MYDET_E_UNSUPP (-2) for shapes it does not cover: the caller then uses mydet_conv2d_wino_f32 / _igemm_f32.
 * Replaces the same reference code as mydet_conv2d_igemm_f32 (models/modules.py:69-73,94-95). */
int64_t mydet_wino4_weights_floats(int Cout, int Cin);
int mydet_wino4_weights_f32(const float *w, int Cout, int Cin, float *u, void *stream);
int64_t mydet_wino4_workspace_bytes(int B, int H, int W, int Cin, int Cout);
/* Re-reads the MYDET_W4_TAIL* tuning variables (they are read once per process otherwise): tests and in-process sweeps. */
int mydet_wino4_reload_tuning(void);
int mydet_conv2d_wino4_f32(const float *x, int64_t ldx, const float *u, const float *scale, const float *shift,
                           const float *residual, int64_t ldr, float *ws, int64_t ws_bytes, float *y, int64_t ldy,
                           int B, int H, int W, int Cin, int Cout, int act, void *stream);
/* Test hook (host only, no GPU call): the K-cut tail plan mydet_conv2d_wino4_f32 uses on a chip that holds `slots` of its
 * workgroups (2 per CU).  Items are ids of 64-id blocks; out[0] = ids covered by the main launch, out[1] = groups, then per
 * group {first id, blocks, ids taken per block, cuts along K, scratch offset in KiB} (out: 17 ints).  Returns the number of
 * groups (0 = no tail) or a negative MYDET_E_*.  No reference counterpart. */
int mydet_wino4_tail_plan(int B, int H, int W, int Cin, int Cout, int slots, int32_t *out);

/* Bilinear resize of one 8-bit RGB image [H][W][3] -> [oh][ow][3] (rows src_row_bytes / dst_row_bytes apart, so the
 * result can land inside a padded batch buffer), bit-exact with PIL.Image.resize(size, BILINEAR), i.e. with the
 * reference's tvf.resize of a PIL image (utils/image_ops.py:22-35, :55-137; api/detection.py:177-205): Pillow's
 * two-pass 8-bit fixed-point filter.  bounds_* int32 [o][2] = (first tap, tap count), k* int32 [o][ks] = 22-bit
 * integer weights, both DEVICE arrays built by Pillow's rule (mydetection_amd/utils/image_ops.py:resample_tables);
 * NULL tables skip that pass (the size must then be unchanged). */
int mydet_resize_bilinear_u8(const unsigned char *src, int H, int W, int64_t src_row_bytes, unsigned char *dst, int oh,
                             int ow, int64_t dst_row_bytes, const int32_t *bounds_x, const int32_t *kx, int ksx,
                             const int32_t *bounds_y, const int32_t *ky, int ksy, void *stream);

/* Batched forms over per-image groups of K detection slots (e.g. the records of mydet_postprocess_records_f32):
 * bbox of image b at bbox + b*bbox_stride (floats), `count[b*count_stride]` slots valid.
 *   to_original: utils/structures.py:175-189 with one pad_info row (ori w, ori h, tl x, tl y, imw, imh) per image,
 *                pad_info DEVICE float [B][6] -- the per-image call of api/detection.py:173-174 for a whole batch.
 *   to_json:     the arithmetic of ImageObjects.to_json (utils/structures.py:243-256), which runs in Python floats:
 *                out[b][k] = { (double)cx - (double)w/2, (double)cy - (double)h/2, (double)w, (double)h, (double)score },
 *                out_cat[b][k] = cat_table[class] (class itself when cat_table is NULL; -1 outside the table);
 *                rows >= count are zero; count may be NULL (all K rows valid, e.g. B = 1 for one ImageObjects). */
int mydet_bboxes_to_original_batched_f32(float *bbox, int64_t bbox_stride, const int32_t *count, int64_t count_stride,
                                         int B, int K, const float *pad_info, void *stream);
int mydet_detections_to_json_f64(const float *bbox, int64_t bbox_stride, const float *score, int64_t score_stride,
                                 const int64_t *cls, int64_t cls_stride, const int32_t *count, int64_t count_stride,
                                 int B, int K, const int64_t *cat_table, int n_cat, double *out, int64_t *out_cat,
                                 void *stream);

/* Fused front half of an MBConv block (external/efficientnet/model.py:71-79): expand 1x1 + BN0 + swish -> depthwise
 * k x k stride s ("static SAME" pad of the EXPANDED map, utils.py:122-145) + BN1 + swish, plus the SE squeeze sums.
 * Replaces mydet_conv2d_igemm_f32 (expand) + mydet_dwconv_f32 for the shallow blocks; the 6x-wide expanded tensor
 * never reaches HBM.  x logical [B,Cin,H,W] (pixel stride ldx); w_expand [Cexp][Cin] and w_dw [K][K][Cexp] with the
 * per-channel BatchNorm scale already multiplied in; shift0 / shift1 = the folded BatchNorm shifts (they initialise
 * the accumulators); y logical [B,Cexp,Ho,Wo].  se_partial (optional): [B][S+1][Cexp] per-tile channel
 * sums of y with S == mydet_mbconv_tiles(Ho, Wo, stride) (slice S is scratch for mydet_se_gate_f32).
 * Instantiated for (K, stride, Cin) in {(3,2,16), (3,1,24), (5,2,24), (5,1,40), (3,2,40)}: MYDET_E_UNSUPP otherwise. */
int mydet_mbconv_tiles(int Ho, int Wo, int stride);
int mydet_mbconv_expand_dw_f32(const float *x, int64_t ldx, const float *w_expand, const float *shift0,
                               const float *w_dw, const float *shift1,
                               float *y, int64_t ldy, int B, int H, int W, int Cin, int Cexp, int K, int stride,
                               int pad_t, int pad_l, int Ho, int Wo, float *se_partial, int S, const mydet_se_tail *se,
                               void *stream);

/* EfficientNet stem fused with the depthwise conv of the first MBConv block (which has expand_ratio 1):
 *     y = swish(BN1(depthwise3x3_s1_pad1( swish(BN0(conv3x3_s2(image))) )))      + per-tile channel sums of y
 * Replaces _conv_stem -> _bn0 -> swish (external/efficientnet/model.py:133-140) and _depthwise_conv -> _bn1 -> swish +
 * the adaptive_avg_pool2d of block 0 (:76-80); the 32-channel stem output never reaches memory.
 * x: the image, logical [B,3,H,W], strides sxb/sxc/sxh/sxw in floats (as mydet_conv2d_stem_f32).  w_stem: OHWI [32][3][3][3]
 * with BN0's scale folded in, shift0 [32]; w_dw [3][3][32] with BN1's scale folded in, shift1 [32].  pad_t / pad_l: the
 * stem's "SAME" padding (top / left; bottom / right follow from Hs, Ws).  y [B,Hs,Ws,ldy].  se_partial (optional):
 * [B][S+1][32] with S == mydet_mbconv_tiles(Hs, Ws, 1).  C must be 32 (MYDET_E_UNSUPP otherwise). */
int mydet_stem_dw_f32(const float *x, int64_t sxb, int64_t sxc, int64_t sxh, int64_t sxw, const float *w_stem,
                      const float *shift0, const float *w_dw, const float *shift1, float *y, int64_t ldy, int B, int H, int W,
                      int C, int pad_t, int pad_l, int Hs, int Ws, float *se_partial, int S, const mydet_se_tail *se,
                      void *stream);

/* Fused separable-conv node of the 88-channel BiFPN / EfDetHead pyramid, several nodes per launch:
 *     y = act( pointwise1x1( depthwise3x3_pad1( pre(in...) ) ) * scale + shift )
 *   n_in == 1: pre = identity                 spconv3x3_bn_swish / last sepconv of a head tower, models/rpns.py:121-205
 *   n_in >= 2: pre = swish(sum_i w_i * in_i), w = relu(fuse_weights) / (sum + 1e-4)     LinearFusion, models/fpns.py:421-439;
 *              mode[i]: 0 same size, 1 half-size map read through nearest 2x, 2 double-size map read through
 *              max_pool2d(3,2,1) (the top-down / bottom-up paths of BiFPN5.forward, models/fpns.py:398-418)
 * Replaces mydet_bifpn_fuse_f32 + mydet_dwconv_f32 + mydet_conv2d_igemm_f32 for these nodes (SeparableConv2d,
 * models/modules.py:5-21, with the following BatchNorm folded into scale/shift).
 * in[i]: logical [B,C,h,w] channels-last, pixel stride ld[i].  w_dw [3][3][C].  w_pw_packed: the pointwise weight
 * W[Cout][C] in MFMA operand order with four k-steps of a lane side by side, ceil(Cout/16) x ceil(C/16) x 64 x 4 floats:
 *     packed[nb][kq][lane][e] = W[16*nb + (lane & 15)][4*(4*kq + e) + (lane >> 4)]   (rows >= Cout and k >= C are zero):
 *     a workgroup copies a block to LDS with 16-byte loads and every wave reads its operands from there.
 * scale may be NULL (no BatchNorm: y = conv + shift).  Cout % 4 == 0.  act: MYDET_ACT_NONE | MYDET_ACT_SWISH.
 * `nodes` is a HOST array of n (<= MYDET_SEPCONV_MAX_NODES) descriptors; all nodes share B and C (C == 88). */
#define MYDET_SEPCONV_MAX_NODES 10
typedef struct {
    const float *in[3];
    int64_t ld[3];
    int mode[3];
    int n_in;
    const float *fuse_weights;
    const float *w_dw;
    const float *w_pw_packed;
    const float *scale;
    const float *shift;
    float *y;
    int64_t ldy;
    int H, W, Cout, act;
} mydet_sepconv_node;
int mydet_sepconv_nodes_f32(int n, const mydet_sepconv_node *nodes, int B, int C, void *stream);

/* The LAST layers of the EfDetHead towers with RetinaLayer's decode in their epilogue: replaces the last
 * SeparableConv2d of class_nets / bbox_nets (models/rpns.py:121-197) + RetinaLayer.forward
 * (models/detlayers/retinanet.py:63-82) + the level concatenation of models/general.py:74-76, i.e.
 * mydet_sepconv_nodes_f32 + mydet_decode_levels_f32(MYDET_DECODE_RETINA) without the A*n_cls class logits per pixel ever
 * reaching memory.  Same results as that pair (same arithmetic, candidate order (a, y, x), first maximum on ties).
 *   kind 0 (class tower): node.Cout = A * cpad, cpad = 16 * ceil(n_cls / 16): w_pw_packed / shift hold anchor a's n_cls
 *           rows at [a * cpad, a * cpad + n_cls), zero rows after them; writes score (sigmoid of the anchor's largest
 *           logit) and class_idx.  n_cls in 65..96.
 *   kind 1 (box tower):   node.Cout = 4 * A (tx, ty, tw, th per anchor); anchors_wh = HOST pointer to A (w, h) pairs in
 *           pixels; writes bbox (cx, cy, w, h clamped to [1, max(img_h, img_w)]).
 * node.y / node.ldy are ignored, node.n_in == 1, node.act == MYDET_ACT_NONE.  n_off = first candidate of the node's
 * level inside [0, N).  bbox [B,N,4] f32, class_idx [B,N] i64, score [B,N] f32.  C == 88, A <= 12. */
typedef struct {
    mydet_sepconv_node node;
    int kind;
    float stride;
    const float *anchors_wh;
    int64_t n_off;
} mydet_sepconv_decode_node;
int mydet_sepconv_decode_retina_f32(int n, const mydet_sepconv_decode_node *nodes, int B, int C, int A, int n_cls,
                                    int img_h, int img_w, float *bbox, int64_t *class_idx, float *score, int64_t N,
                                    void *stream);

/* Pairwise IoU [Na,Nb]; utils/bbox_ops.py:6-49 (xyxy != 0: corner format, else cxcywh). */
int mydet_bboxes_iou_f32(const float *a, int Na, const float *b, int Nb, int xyxy,
                         float *iou, void *stream);

/* Centre format -> corner format, utils/bbox_ops.py:309-316: n rows of `width` >= 4 floats; columns 0..3 of a row
 * become (cx - w/2, cy - h/2, cx + w/2, cy + h/2) with the reference's float32 operation order (bit-exact), columns
 * 4.. are copied.  Out of place (in == out is allowed: a thread reads its row before it writes it). */
int mydet_cxcywh_to_x1y1x2y2_f32(const float *cxcywh, float *x1y1x2y2, int64_t n, int width, void *stream);

/* In-place undo of resize/pad on cxcywh boxes; utils/structures.py:175-189. */
int mydet_bboxes_to_original_f32(float *bbox, int64_t n, float ori_w, float ori_h,
                                 float tl_x, float tl_y, float imw, float imh, void *stream);

/* Device-side image preparation (first stage before the path): uint8 [B,H,W,3] images ->
 * float32 [B,3,Hp,Wp]: zero-pad right/bottom (utils/image_ops.py:38-52), /255 (tvf.to_tensor,
 * api/detection.py:160), and, when norm != 0, (x - mean)/std per channel (utils/image_ops.py:177-180).
 * mean3/std3 are HOST pointers to 3 floats. */
int mydet_preprocess_u8_f32(const unsigned char *img, int B, int H, int W, float *out, int Hp, int Wp, int norm,
                            const float *mean3, const float *std3, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MYDET_H */
