/* Oracle (test infrastructure only; never linked into the product library).
 *
 * Plain-C restatement of the CPU kernel behind torchvision.ops.nms, which the
 * reference calls at utils/structures.py:133,162 (per-class loop :158-167).
 * torchvision is a third-party dependency that is absent from /root/reference
 * and not installed here (version unpinned: the reference has no requirements
 * file), so this follows the published algorithm of
 * torchvision/csrc/ops/cpu/nms_kernel.cpp `nms_kernel_impl<float>`:
 *   areas = (x2-x1)*(y2-y1); order = stable sort of scores, descending;
 *   for i in order: if suppressed skip; keep i; for later j in order:
 *     w = max(0, min(x2)-max(x1)); h likewise; inter = w*h;
 *     ovr = inter / (area_i + area_j - inter);   (float32)
 *     if (ovr > iou_threshold) suppress j         (threshold is a double)
 * PARITY UNPINNED at this boundary (see oracle/__init__.py).
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off: no FMA contraction, as the
 * generic x86-64 torchvision build has none).
 */
#include <stdint.h>
#include <stdlib.h>

static void stable_order_desc(const float *scores, int n, int32_t *order, int32_t *tmp) {
    /* bottom-up merge sort: stable, descending */
    for (int i = 0; i < n; ++i) order[i] = i;
    for (int width = 1; width < n; width *= 2) {
        for (int lo = 0; lo < n; lo += 2 * width) {
            int mid = lo + width < n ? lo + width : n;
            int hi = lo + 2 * width < n ? lo + 2 * width : n;
            int a = lo, b = mid, k = lo;
            while (a < mid && b < hi) {
                /* take from the right run only when strictly greater: keeps ties in input order */
                if (scores[order[b]] > scores[order[a]]) tmp[k++] = order[b++];
                else tmp[k++] = order[a++];
            }
            while (a < mid) tmp[k++] = order[a++];
            while (b < hi) tmp[k++] = order[b++];
        }
        for (int i = 0; i < n; ++i) order[i] = tmp[i];
    }
}

/* boxes: n rows of (x1,y1,x2,y2); keep: out, capacity n; returns number kept */
int nms_ref_f32(const float *boxes, const float *scores, int n, double iou_threshold,
                int64_t *keep) {
    if (n <= 0) return 0;
    int32_t *order = (int32_t *)malloc(sizeof(int32_t) * (size_t)n * 2);
    uint8_t *suppressed = (uint8_t *)calloc((size_t)n, 1);
    float *areas = (float *)malloc(sizeof(float) * (size_t)n);
    stable_order_desc(scores, n, order, order + n);
    for (int i = 0; i < n; ++i)
        areas[i] = (boxes[4 * i + 2] - boxes[4 * i + 0]) * (boxes[4 * i + 3] - boxes[4 * i + 1]);
    int num = 0;
    for (int _i = 0; _i < n; ++_i) {
        int i = order[_i];
        if (suppressed[i]) continue;
        keep[num++] = i;
        float ix1 = boxes[4 * i], iy1 = boxes[4 * i + 1], ix2 = boxes[4 * i + 2], iy2 = boxes[4 * i + 3];
        float iarea = areas[i];
        for (int _j = _i + 1; _j < n; ++_j) {
            int j = order[_j];
            if (suppressed[j]) continue;
            float xx1 = ix1 > boxes[4 * j] ? ix1 : boxes[4 * j];
            float yy1 = iy1 > boxes[4 * j + 1] ? iy1 : boxes[4 * j + 1];
            float xx2 = ix2 < boxes[4 * j + 2] ? ix2 : boxes[4 * j + 2];
            float yy2 = iy2 < boxes[4 * j + 3] ? iy2 : boxes[4 * j + 3];
            float w = xx2 - xx1; if (!(w > 0.0f)) w = 0.0f;
            float h = yy2 - yy1; if (!(h > 0.0f)) h = 0.0f;
            float inter = w * h;
            float ovr = inter / (iarea + areas[j] - inter);
            if ((double)ovr > iou_threshold) suppressed[j] = 1;
        }
    }
    free(order); free(suppressed); free(areas);
    return num;
}
