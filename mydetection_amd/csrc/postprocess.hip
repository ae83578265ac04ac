// Batched post-processing: confidence filter -> top-k -> class-aware greedy NMS.
//
// One 1024-thread workgroup per image; the N candidates (25 200 for YOLOv3 @640^2)
// never leave HBM and only <= 512 survivors per image are written.  The work is
// latency-bound (<= 512 boxes, <= 131k IoUs per image), not bandwidth-bound, so the
// design goal is few dependent steps:
//   1. filter: one coalesced sweep of score[N]; passing candidates become unique 64-bit
//      keys (sortable score bits << 32 | ~index) compacted into an HBM scratch strip.
//   2. top-k (only if more than k pass): 8-pass byte-wise radix SELECT of the k-th
//      largest key -- no sort of the N candidates.  Key order = score desc, index asc,
//      which fixes the tie order torch.topk leaves unspecified.
//   3. bitonic sort of the <= 512 selected in LDS by (class asc, score desc, index asc):
//      this is the reference's output order (per-class loop over sorted unique labels,
//      utils/structures.py:158-167, each class in stable score order), and it makes
//      every class a contiguous segment.
//   4. suppression matrix: bit j of row i set iff j > i, same class, (double)IoU > thr --
//      IoU in the exact float32 operation order of torchvision's CPU nms kernel (this file
//      is built with -ffp-contract=off).
//   5. greedy pass by one wave over the 512-bit rows (next rows prefetched off the
//      dependent chain), then rank-by-popcount compaction of survivors.
// Replaces ImageObjects.post_process / non_max_suppression (utils/structures.py:92-173).
#include "common.h"

namespace {

constexpr int KMAX = 512;
constexpr int NT = 1024;
constexpr int IDX_BITS = 17;

struct PPArgs {
    const float *bbox;
    const int64_t *cidx;
    const float *score;
    int64_t N;
    float conf;
    double nms;
    int topk;
    int32_t *count;
    float *obox;
    int64_t *ocls;
    float *oscore;
    int32_t *oidx;
    unsigned long long *scratch;
};

__device__ __forceinline__ unsigned sortable(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(NT) void postprocess_kernel(const PPArgs p) {
    __shared__ unsigned long long s_key[KMAX];
    __shared__ unsigned long long s_mask[KMAX * 8];
    __shared__ float s_x1[KMAX], s_y1[KMAX], s_x2[KMAX], s_y2[KMAX], s_area[KMAX];
    __shared__ int s_cls[KMAX];
    __shared__ unsigned s_hist[256];
    __shared__ unsigned long long s_removed[8];
    __shared__ unsigned long long s_prefix;
    __shared__ int s_n, s_nsel, s_need;

    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    const float *sc = p.score + (int64_t)b * p.N;
    const int64_t *ci = p.cidx + (int64_t)b * p.N;
    const float *bb = p.bbox + (int64_t)b * p.N * 4;
    unsigned long long *keys = p.scratch + (int64_t)b * p.N;
    const int N = (int)p.N;

    if (tid == 0) { s_n = 0; s_nsel = 0; }
    __syncthreads();
    // 1. filter (>= in float32, as `self.scores >= conf_thres`)
    for (int i = tid; i < N; i += NT) {
        const float s = sc[i];
        if (s >= p.conf) {
            const int pos = atomicAdd(&s_n, 1);
            keys[pos] = ((unsigned long long)sortable(s) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
        }
    }
    __syncthreads();
    const int n = s_n;
    // scratch written above is re-read by other threads of this workgroup only
    __threadfence_block();

    // 2. k-th largest key by radix select
    unsigned long long kth = 0;
    if (n > p.topk) {
        if (tid == 0) { s_prefix = 0; s_need = p.topk; }
        for (int pass = 0; pass < 8; ++pass) {
            const int shift = 56 - 8 * pass;
            if (tid < 256) s_hist[tid] = 0;
            __syncthreads();
            const unsigned long long prefix = s_prefix;
            const unsigned long long himask = pass == 0 ? 0ull : (~0ull << (shift + 8));
            for (int i = tid; i < n; i += NT) {
                const unsigned long long k = keys[i];
                if ((k & himask) == prefix) atomicAdd(&s_hist[(unsigned)(k >> shift) & 255u], 1u);
            }
            __syncthreads();
            if (tid == 0) {
                int need = s_need;
                int bin = 255;
                for (; bin > 0; --bin) {
                    const int h = (int)s_hist[bin];
                    if (h >= need) break;
                    need -= h;
                }
                s_need = need;
                s_prefix = prefix | ((unsigned long long)bin << shift);
            }
            __syncthreads();
        }
        kth = s_prefix;
    }

    // 3. gather the selected, build (class, ~score, index) keys, sort
    for (int i = tid; i < KMAX; i += NT) s_key[i] = ~0ull;
    __syncthreads();
    for (int i = tid; i < n; i += NT) {
        const unsigned long long k = keys[i];
        if (k >= kth) {
            const int pos = atomicAdd(&s_nsel, 1);
            const unsigned idx = 0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull);
            const unsigned su = (unsigned)(k >> 32);
            const unsigned long long c = (unsigned long long)ci[idx] & 0x7FFFull;
            s_key[pos] = (c << 49) | ((unsigned long long)(~su) << IDX_BITS) | (unsigned long long)idx;
        }
    }
    __syncthreads();
    const int nsel = s_nsel;
    for (int k = 2; k <= KMAX; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (tid < KMAX) {
                const int ixj = tid ^ j;
                if (ixj > tid) {
                    const unsigned long long a = s_key[tid], c = s_key[ixj];
                    const bool up = (tid & k) == 0;
                    if ((a > c) == up) { s_key[tid] = c; s_key[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }

    // 4. corners + areas, then the suppression matrix
    if (tid < nsel) {
        const unsigned long long k = s_key[tid];
        const unsigned idx = (unsigned)(k & ((1ull << IDX_BITS) - 1));
        const f32x4 v = *reinterpret_cast<const f32x4 *>(bb + (int64_t)idx * 4);
        const float hw = v[2] / 2.0f, hh = v[3] / 2.0f;
        const float x1 = v[0] - hw, y1 = v[1] - hh, x2 = v[0] + hw, y2 = v[1] + hh;
        s_x1[tid] = x1; s_y1[tid] = y1; s_x2[tid] = x2; s_y2[tid] = y2;
        s_area[tid] = (x2 - x1) * (y2 - y1);
        s_cls[tid] = (int)(k >> 49);
    }
    __syncthreads();
    const int nwords = (nsel + 63) >> 6;
    for (int item = tid; item < nsel * 8; item += NT) {
        const int i = item >> 3, w = item & 7;
        unsigned long long bits = 0;
        if (w < nwords && w * 64 + 63 > i) {
            const float ix1 = s_x1[i], iy1 = s_y1[i], ix2 = s_x2[i], iy2 = s_y2[i], ia = s_area[i];
            const int ic = s_cls[i];
            for (int jj = 0; jj < 64; ++jj) {
                const int j = w * 64 + jj;
                if (j > i && j < nsel && s_cls[j] == ic) {
                    const float xx1 = fmaxf(ix1, s_x1[j]), yy1 = fmaxf(iy1, s_y1[j]);
                    const float xx2 = fminf(ix2, s_x2[j]), yy2 = fminf(iy2, s_y2[j]);
                    const float ww = fmaxf(0.0f, xx2 - xx1), hh = fmaxf(0.0f, yy2 - yy1);
                    const float inter = ww * hh;
                    const float ovr = inter / (ia + s_area[j] - inter);
                    if ((double)ovr > p.nms) bits |= 1ull << jj;
                }
            }
        }
        s_mask[item] = bits;
    }
    __syncthreads();

    // 5. greedy scan (wave 0; lane w < 8 owns word w of the removed set)
    if (tid < 64) {
        unsigned long long removed = 0;
        const int lw = tid & 7;
        unsigned long long row = nsel > 0 ? s_mask[lw] : 0ull;
        for (int i = 0; i < nsel; ++i) {
            const unsigned long long nxt = (i + 1 < nsel) ? s_mask[(i + 1) * 8 + lw] : 0ull;   // off the chain
            const unsigned long long rw = __shfl(removed, i >> 6);
            if (!((rw >> (i & 63)) & 1ull)) removed |= row;
            row = nxt;
        }
        if (tid < 8) s_removed[tid] = removed;
    }
    __syncthreads();

    // 6. survivors, ranked by popcount, in sorted (class asc, score desc) order
    int total = 0;
    {
        int before = 0;
        const int myw = tid >> 6;
        for (int w = 0; w < 8; ++w) {
            const int lim = nsel - w * 64;
            unsigned long long valid = lim >= 64 ? ~0ull : (lim <= 0 ? 0ull : ((1ull << lim) - 1ull));
            const unsigned long long kept = ~s_removed[w] & valid;
            const int pc = __popcll(kept);
            if (w < myw) before += pc;
            total += pc;
        }
        if (tid < nsel) {
            const unsigned long long kept = ~s_removed[myw];
            if ((kept >> (tid & 63)) & 1ull) {
                const int pos = before + __popcll(kept & ((1ull << (tid & 63)) - 1ull));
                const unsigned long long k = s_key[tid];
                const unsigned idx = (unsigned)(k & ((1ull << IDX_BITS) - 1));
                const int64_t o = (int64_t)b * p.topk + pos;
                *reinterpret_cast<f32x4 *>(p.obox + o * 4) = *reinterpret_cast<const f32x4 *>(bb + (int64_t)idx * 4);
                p.ocls[o] = ci[idx];
                p.oscore[o] = sc[idx];
                p.oidx[o] = (int32_t)idx;
            }
        }
    }
    for (int r = total + tid; r < p.topk; r += NT) {
        const int64_t o = (int64_t)b * p.topk + r;
        *reinterpret_cast<f32x4 *>(p.obox + o * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
        p.ocls[o] = 0;
        p.oscore[o] = 0.f;
        p.oidx[o] = 0;
    }
    if (tid == 0) p.count[b] = total;
}

}  // namespace

extern "C" int mydet_postprocess_f32(const float *bbox, const int64_t *class_idx, const float *score, int B,
                                     int64_t N, float conf_thres, double nms_thres, int topk, int32_t *count,
                                     float *out_bbox, int64_t *out_class, float *out_score, int32_t *out_index,
                                     void *scratch, void *stream) {
    if (B <= 0 || N < 0 || topk <= 0 || topk > KMAX) return MYDET_E_BADARG;
    if (N >= (1ll << IDX_BITS)) return MYDET_E_UNSUPP;
    if (!count || !out_bbox || !out_class || !out_score || !out_index) return MYDET_E_BADARG;
    if (N > 0 && (!bbox || !class_idx || !score || !scratch)) return MYDET_E_BADARG;
    if (((uintptr_t)bbox & 15) || ((uintptr_t)out_bbox & 15) || ((uintptr_t)scratch & 7)) return MYDET_E_BADARG;
    PPArgs p;
    p.bbox = bbox; p.cidx = class_idx; p.score = score; p.N = N; p.conf = conf_thres; p.nms = nms_thres;
    p.topk = topk; p.count = count; p.obox = out_bbox; p.ocls = out_class; p.oscore = out_score;
    p.oidx = out_index; p.scratch = (unsigned long long *)scratch;
    hipLaunchKernelGGL(postprocess_kernel, dim3(B), dim3(NT), 0, (hipStream_t)stream, p);
    return mydet_launch_status();
}
