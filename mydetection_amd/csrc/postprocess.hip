// Batched post-processing: confidence filter -> top-k -> class-aware greedy NMS.
//
// One 1024-thread workgroup per image; the N candidates (25 200 for YOLOv3 @640^2)
// never leave HBM and only <= 512 survivors per image are written.  The work is
// latency-bound (<= 512 boxes, <= 131k IoUs per image), not bandwidth-bound, so the
// design goal is few dependent steps:
//   1. filter: one coalesced sweep of score[N]; passing candidates become unique 64-bit
//      keys (sortable score bits << 32 | ~index) compacted into an HBM scratch strip.
//   2. top-k (only if more than k pass): the k-th largest key is found by 4096-bin histogram rounds over its bits
//      from the top, then bit by bit on the few keys left -- no sort of the N candidates.  Key order = score desc,
//      index asc, which fixes the tie order torch.topk leaves unspecified.
//   3. bitonic sort of the <= 512 selected in LDS by (class asc, score desc, index asc):
//      this is the reference's output order (per-class loop over sorted unique labels,
//      utils/structures.py:158-167, each class in stable score order), and it makes
//      every class a contiguous segment.
//   4. suppression matrix: bit j of row i set iff j > i, same class, (double)IoU > thr --
//      IoU in the exact float32 operation order of torchvision's CPU nms kernel (this file
//      is built with -ffp-contract=off).
//   5. greedy selection as a fixed point: removed <- OR of the rows of the boxes not removed, all 16 waves per
//      round, until nothing changes (the greedy recursion has one solution, so the fixed point is it; real
//      detections settle in a few rounds); inputs that do not settle in 12 rounds take the sequential form
//      (one wave, 64 boxes at a time, dependent chain on the scalar unit).  Rank-by-popcount compaction follows.
// Replaces ImageObjects.post_process / non_max_suppression (utils/structures.py:92-173).
#include "common.h"

namespace {

constexpr int KMAX = 512;
constexpr int NT = 1024;
constexpr int IDX_BITS = 20;      // sort key = class (12 bits) | ~score (32) | candidate index (20)
constexpr int CLS_SHIFT = 32 + IDX_BITS;

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
    // per-image strides of the five outputs, in elements of each (dense arrays, or fields of one wire record)
    int64_t count_st, obox_st, ocls_st, oscore_st, oidx_st;
    int count_pad;                    // zero words written behind the count (record header padding)
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
    __shared__ int s_cls[KMAX], s_segend[KMAX];
    __shared__ int s_cnt[3];
    __shared__ unsigned long long s_removed[8];
    __shared__ int s_n, s_nsel, s_bad;

    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    const float *sc = p.score + (int64_t)b * p.N;
    const int64_t *ci = p.cidx + (int64_t)b * p.N;
    const float *bb = p.bbox + (int64_t)b * p.N * 4;
    unsigned long long *keys = p.scratch + (int64_t)b * p.N;
    const int N = (int)p.N;

    if (tid == 0) { s_n = 0; s_nsel = 0; s_bad = 0; }
    __syncthreads();
    // 1. filter (>= in float32, as `self.scores >= conf_thres`)
    //    FU independent loads in flight per thread (a sweep of 76 725 scores -- EfficientDet-D1 -- is five dependent
    //    round trips to memory other XCDs have just written instead of ten); a wave reserves its slots with ONE LDS
    //    atomic (ballot + popcount ranks), so the strip order varies between runs but the key set does not -- and only
    //    the set is used.
    constexpr int FU = 16;
    for (int base = 0; base < N; base += FU * NT) {
        float s8[FU];
#pragma unroll
        for (int u = 0; u < FU; ++u) {
            const int i = base + u * NT + tid;
            s8[u] = __builtin_nontemporal_load(sc + (i < N ? i : N - 1));      // read once; clamped, masked below
        }
#pragma unroll
        for (int u = 0; u < FU; ++u) {
            const int i = base + u * NT + tid;
            const bool pass = i < N && s8[u] >= p.conf;
            const unsigned long long m = __ballot(pass);
            if (m) {                                            // wave-uniform
                int start = 0;
                if ((tid & 63) == 0) start = atomicAdd(&s_n, __popcll(m));
                start = __shfl(start, 0);
                if (pass)
                    keys[start + __popcll(m & ((1ull << (tid & 63)) - 1ull))] =
                        ((unsigned long long)sortable(s8[u]) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
            }
        }
    }
    __syncthreads();
    const int n = s_n;
    // scratch written above is re-read by other threads of this workgroup only
    __threadfence_block();

    // 2. k-th largest key (only if more than k pass).  4096-bin histogram rounds fix its bits twelve at a time from the
    //    top (one LDS atomic per key that still matches, a block-wide suffix scan finds the bin holding the wanted
    //    rank) until at most 1024 keys share the prefix -- two rounds for distinct scores, up to five when thousands
    //    of candidates tie on one score and only the index bits tell them apart.  Those keys are compacted into LDS
    //    and the remaining bits are built bit by bit on the list (ballot + popcount counting, stops as soon as a
    //    candidate cuts off exactly the wanted number).  The first 16 keys of a thread stay in registers; the others
    //    are re-read with eight loads in flight.
    unsigned long long kth = 0;
    if (n > p.topk) {
        constexpr int RK = 16, LIST = 1024;
        __shared__ int s_wtot[NT / 64], s_sel[3], s_m;
        unsigned *hist = reinterpret_cast<unsigned *>(s_mask);                 // 4096 bins (s_mask is free until step 4)
        unsigned long long *list = s_mask + 2048;                              // LIST keys behind the histogram
        unsigned long long rk[RK];
#pragma unroll
        for (int j = 0; j < RK; ++j) rk[j] = tid + j * NT < n ? keys[tid + j * NT] : 0ull;
        const int wave = tid >> 6, lane = tid & 63;
        // keys beyond the register-resident ones: eight independent loads in flight, then `f` on each
        auto for_tail = [&](auto &&f) {
            for (int base = tid + RK * NT; base < n; base += 8 * NT) {
                unsigned long long kk[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) kk[u] = base + u * NT < n ? keys[base + u * NT] : 0ull;
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (base + u * NT < n) f(kk[u]);
            }
        };
        unsigned long long prefix = 0;
        int need = p.topk;                                  // rank of the wanted key among the keys matching `prefix`
        int low = 64;                                       // bits of `prefix` not fixed yet
        for (int level = 0; level < 5; ++level) {
            const int shift = 52 - 12 * level;
            for (int i = tid; i < 4096; i += NT) hist[i] = 0u;
            __syncthreads();
            auto tally = [&](unsigned long long k) {
                if (level == 0 || (k >> low) == (prefix >> low)) atomicAdd(&hist[(unsigned)(k >> shift) & 4095u], 1u);
            };
#pragma unroll
            for (int j = 0; j < RK; ++j)
                if (tid + j * NT < n) tally(rk[j]);
            for_tail(tally);
            __syncthreads();
            // thread t owns bins 4t..4t+3; suffix sums over the threads above it locate the bin of rank `need`
            const unsigned c0 = hist[4 * tid], c1 = hist[4 * tid + 1], c2 = hist[4 * tid + 2], c3 = hist[4 * tid + 3];
            const int own = (int)(c0 + c1 + c2 + c3);
            int suf = own;                                   // inclusive suffix sum inside the wave
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int v = __shfl_down(suf, off);
                if (lane + off < 64) suf += v;
            }
            if (lane == 0) s_wtot[wave] = suf;
            __syncthreads();
            int above = suf - own;                           // keys in bins above this thread's four
            for (int w = wave + 1; w < NT / 64; ++w) above += s_wtot[w];
            if (above < need && above + own >= need) {       // exactly one thread
                const unsigned c[4] = {c0, c1, c2, c3};
                int a = above, bsel = 0;
#pragma unroll
                for (int q = 3; q >= 0; --q) {
                    if (a < need && a + (int)c[q] >= need) { bsel = q; break; }
                    a += (int)c[q];
                }
                s_sel[0] = 4 * tid + bsel;
                s_sel[1] = a;
                s_sel[2] = (int)c[bsel];
            }
            __syncthreads();
            prefix |= (unsigned long long)(unsigned)s_sel[0] << shift;
            need -= s_sel[1];
            low = shift;
            const int in_bin = s_sel[2];
            __syncthreads();
            if (in_bin <= LIST) break;                       // uniform; after five rounds low = 4: at most 16 keys match
        }
        // keys sharing the fixed prefix -> LDS list (at most LIST of them)
        if (tid == 0) s_m = 0;
        if (tid < 3) s_cnt[tid] = 0;
        __syncthreads();
        auto collect = [&](unsigned long long k) {
            if ((k >> low) == (prefix >> low)) {
                const int pos = atomicAdd(&s_m, 1);
                list[pos < LIST ? pos : LIST - 1] = k;
            }
        };
#pragma unroll
        for (int j = 0; j < RK; ++j)
            if (tid + j * NT < n) collect(rk[j]);
        for_tail(collect);
        __syncthreads();
        const int m = s_m;
        const unsigned long long lk = tid < m ? list[tid] : 0ull;
        for (int bit = low - 1, it = 0; bit >= 0; --bit, ++it) {
            const unsigned long long cand = prefix | (1ull << bit);
            int c = tid < m && lk >= cand ? 1 : 0;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off);
            if (lane == 0) atomicAdd(&s_cnt[it % 3], c);
            if (tid == 0) s_cnt[(it + 1) % 3] = 0;
            __syncthreads();
            const int cnt = s_cnt[it % 3];
            if (cnt >= need) prefix = cand;
            if (cnt == need) break;
        }
        kth = prefix;
        __syncthreads();                                    // s_mask is reused by step 4
    }

    // 3. gather the selected, build (class, ~score, index) keys, sort
    for (int i = tid; i < KMAX; i += NT) s_key[i] = ~0ull;
    __syncthreads();
    for (int base = tid; base < n; base += 8 * NT) {         // eight independent key loads in flight
        unsigned long long kk[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) kk[u] = base + u * NT < n ? keys[base + u * NT] : 0ull;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned long long k = kk[u];
            if (base + u * NT < n && k >= kth) {
                const int pos = atomicAdd(&s_nsel, 1);
                const unsigned idx = 0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull);
                const unsigned su = (unsigned)(k >> 32);
                const unsigned long long c = (unsigned long long)ci[idx];       // negative ids wrap to huge values
                if (c > 0xFFFull) s_bad = 1;                                    // the sort key has 12 class bits
                s_key[pos] = ((c & 0xFFFull) << CLS_SHIFT) | ((unsigned long long)(~su) << IDX_BITS) | (unsigned long long)idx;
            }
        }
    }
    __syncthreads();
    const int nsel = s_nsel;
    const bool bad = s_bad != 0;       // a selected candidate's class id is outside [0, 4096): fail the image loudly
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
        s_cls[tid] = (int)(k >> CLS_SHIFT);
    }
    __syncthreads();
    // rows are zero except inside the row's own class segment (classes are contiguous after the sort).  
    for (int w = tid; w < nsel * 8; w += NT) s_mask[w] = 0ull;
    if (tid < nsel) {                                   // end of my class segment: upper bound of my class
        const int ic = s_cls[tid];
        int lo = tid + 1, hi = nsel;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (s_cls[mid] == ic) lo = mid + 1; else hi = mid;
        }
        s_segend[tid] = lo;
    }
    __syncthreads();
    // a wave owns rows i = wave, wave+16, ...; for each it walks only the 64-box words that overlap
    // [i+1, end of i's class segment): lane = one j, the word is the wave's ballot (no atomics, conflict-free LDS)
    {
        const int wave = tid >> 6, lane = tid & 63;
        for (int i = wave; i < nsel; i += NT / 64) {
            const int e = s_segend[i];
            const float ix1 = s_x1[i], iy1 = s_y1[i], ix2 = s_x2[i], iy2 = s_y2[i], ia = s_area[i];
            for (int w = (i + 1) >> 6; w * 64 < e; ++w) {
                const int j = w * 64 + lane;
                bool sup = false;
                if (j > i && j < e) {
                    const float xx1 = fmaxf(ix1, s_x1[j]), yy1 = fmaxf(iy1, s_y1[j]);
                    const float xx2 = fminf(ix2, s_x2[j]), yy2 = fminf(iy2, s_y2[j]);
                    const float ww = fmaxf(0.0f, xx2 - xx1), hh = fmaxf(0.0f, yy2 - yy1);
                    const float inter = ww * hh;
                    const float ovr = inter / (ia + s_area[j] - inter);
                    sup = (double)ovr > p.nms;
                }
                const unsigned long long bits = __ballot(sup);
                if (lane == 0) s_mask[i * 8 + w] = bits;
            }
        }
    }
    __syncthreads();

    // 5. greedy selection.  kept[i] = no kept j < i suppresses i, and that recursion has exactly one solution, so
    //    any fixed point of  removed <- OR of the rows of the boxes not in removed  IS the greedy result.  Starting
    //    from removed = 0, round t fixes every box whose chain of (suppressor of suppressor of ...) is shorter than
    //    t -- a handful of rounds on real detections -- and each round is one parallel OR-reduction of the 512 x 8
    //    mask words by all 16 waves.  Inputs that have not settled after MAX_ROUNDS fall back to the sequential scan.
    constexpr int MAX_ROUNDS = 12;
    __shared__ unsigned long long s_part[NT / 64][8];
    __shared__ int s_changed;
    bool settled = false;
    {
        const int wave = tid >> 6, lane = tid & 63;
        // rows r = tid and tid + 512 (two half-rows per thread would not balance; a thread owns 4 words of a row)
        const int row = tid >> 1, w0 = (tid & 1) * 4;
        unsigned long long mine[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) mine[w] = row < nsel ? s_mask[row * 8 + w0 + w] : 0ull;
        if (tid < 8) s_removed[tid] = 0ull;
        __syncthreads();
        for (int round = 0; round < MAX_ROUNDS; ++round) {
            const bool kept = row < nsel && !((s_removed[row >> 6] >> (row & 63)) & 1ull);
            unsigned long long acc[4];
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                unsigned lo = kept ? (unsigned)mine[w] : 0u, hi = kept ? (unsigned)(mine[w] >> 32) : 0u;
                // lanes with the same parity hold the same four words: reduce over stride-2 partners
#pragma unroll
                for (int off = 32; off >= 2; off >>= 1) {
                    lo |= __shfl_xor(lo, off);
                    hi |= __shfl_xor(hi, off);
                }
                acc[w] = ((unsigned long long)hi << 32) | lo;
            }
            if (lane < 2) {
#pragma unroll
                for (int w = 0; w < 4; ++w) s_part[wave][lane * 4 + w] = acc[w];
            }
            if (tid == 0) s_changed = 0;
            __syncthreads();
            if (tid < 8) {
                unsigned long long r = 0ull;
#pragma unroll
                for (int v = 0; v < NT / 64; ++v) r |= s_part[v][tid];
                if (r != s_removed[tid]) { s_removed[tid] = r; s_changed = 1; }
            }
            __syncthreads();
            if (!s_changed) { settled = true; break; }         // uniform: read after the barrier by every thread
            __syncthreads();                                    // s_changed is reset at the top of the next round
        }
    }
    // sequential form (wave 0; lane w < 8 owns word w of the removed set): one wave, 64 boxes at a time.  Lane l
    // holds row 64c+l.  Inside a chunk the dependent chain runs on the scalar unit only (bit test on a 64-bit scalar,
    // v_readlane of the diagonal word, scalar OR); the chunk's survivors then OR their rows into the later words.
    if (!settled && tid < 64) {
        const int lane = tid;
        unsigned long long removed[8];
#pragma unroll
        for (int w = 0; w < 8; ++w) removed[w] = 0ull;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if (c * 64 < nsel) {                                   // uniform
                const int r = c * 64 + lane;
                const bool valid = r < nsel;
                unsigned long long mw[8];
#pragma unroll
                for (int w = c; w < 8; ++w) mw[w] = valid ? s_mask[r * 8 + w] : 0ull;
                const unsigned dlo = (unsigned)mw[c], dhi = (unsigned)(mw[c] >> 32);
                unsigned long long rc = removed[c];
                const int nb = nsel - c * 64 < 64 ? nsel - c * 64 : 64;
                for (int bq = 0; bq < nb; ++bq) {
                    const unsigned long long d = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(dhi, bq) << 32) |
                                                 (unsigned long long)(unsigned)__builtin_amdgcn_readlane(dlo, bq);
                    if (!((rc >> bq) & 1ull)) rc |= d;
                }
                removed[c] = rc;
                const bool kept = valid && !((rc >> lane) & 1ull);
#pragma unroll
                for (int w = c + 1; w < 8; ++w) {
                    unsigned lo = kept ? (unsigned)mw[w] : 0u, hi = kept ? (unsigned)(mw[w] >> 32) : 0u;
#pragma unroll
                    for (int off = 32; off >= 1; off >>= 1) {
                        lo |= __shfl_xor(lo, off);
                        hi |= __shfl_xor(hi, off);
                    }
                    removed[w] |= ((unsigned long long)hi << 32) | lo;
                }
            }
        }
#pragma unroll
        for (int w = 0; w < 8; ++w)
            if (lane == w) s_removed[w] = removed[w];
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
        if (bad) total = 0;
        if (tid < nsel && !bad) {
            const unsigned long long kept = ~s_removed[myw];
            if ((kept >> (tid & 63)) & 1ull) {
                const int pos = before + __popcll(kept & ((1ull << (tid & 63)) - 1ull));
                const unsigned long long k = s_key[tid];
                const unsigned idx = (unsigned)(k & ((1ull << IDX_BITS) - 1));
                *reinterpret_cast<f32x4 *>(p.obox + b * p.obox_st + pos * 4) = *reinterpret_cast<const f32x4 *>(bb + (int64_t)idx * 4);
                p.ocls[b * p.ocls_st + pos] = ci[idx];
                p.oscore[b * p.oscore_st + pos] = sc[idx];
                p.oidx[b * p.oidx_st + pos] = (int32_t)idx;
            }
        }
    }
    for (int r = total + tid; r < p.topk; r += NT) {
        *reinterpret_cast<f32x4 *>(p.obox + b * p.obox_st + r * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
        p.ocls[b * p.ocls_st + r] = 0;
        p.oscore[b * p.oscore_st + r] = 0.f;
        p.oidx[b * p.oidx_st + r] = 0;
    }
    if (tid <= p.count_pad) p.count[b * p.count_st + tid] = tid == 0 ? (bad ? MYDET_COUNT_BAD_CLASS : total) : 0;
}

}  // namespace

static int launch_postprocess(PPArgs &p, const float *bbox, const int64_t *class_idx, const float *score, int B, int64_t N,
                              float conf_thres, double nms_thres, int topk, void *scratch, void *stream) {
    if (B <= 0 || N < 0 || topk <= 0 || topk > KMAX) return MYDET_E_BADARG;
    if (N >= (1ll << IDX_BITS)) return MYDET_E_UNSUPP;
    if (!p.count || !p.obox || !p.ocls || !p.oscore || !p.oidx) return MYDET_E_BADARG;
    if (N > 0 && (!bbox || !class_idx || !score || !scratch)) return MYDET_E_BADARG;
    if (((uintptr_t)bbox & 15) || ((uintptr_t)p.obox & 15) || ((uintptr_t)scratch & 7) || ((uintptr_t)p.ocls & 7)) return MYDET_E_BADARG;
    p.bbox = bbox; p.cidx = class_idx; p.score = score; p.N = N; p.conf = conf_thres; p.nms = nms_thres;
    p.topk = topk; p.scratch = (unsigned long long *)scratch;
    hipLaunchKernelGGL(postprocess_kernel, dim3(B), dim3(NT), 0, (hipStream_t)stream, p);
    return mydet_launch_status();
}

extern "C" int mydet_postprocess_f32(const float *bbox, const int64_t *class_idx, const float *score, int B,
                                     int64_t N, float conf_thres, double nms_thres, int topk, int32_t *count,
                                     float *out_bbox, int64_t *out_class, float *out_score, int32_t *out_index,
                                     void *scratch, void *stream) {
    PPArgs p;
    p.count = count; p.obox = out_bbox; p.ocls = out_class; p.oscore = out_score; p.oidx = out_index;
    p.count_pad = 0; p.count_st = 1; p.obox_st = (int64_t)topk * 4; p.ocls_st = topk; p.oscore_st = topk; p.oidx_st = topk;
    return launch_postprocess(p, bbox, class_idx, score, B, N, conf_thres, nms_thres, topk, scratch, stream);
}

extern "C" int mydet_postprocess_records_f32(const float *bbox, const int64_t *class_idx, const float *score, int B,
                                             int64_t N, float conf_thres, double nms_thres, int32_t *records,
                                             void *scratch, void *stream) {
    if ((uintptr_t)records & 15) return MYDET_E_BADARG;
    PPArgs p;
    if (!records) { p.count = nullptr; p.obox = nullptr; p.ocls = nullptr; p.oscore = nullptr; p.oidx = nullptr;
                    return launch_postprocess(p, bbox, class_idx, score, B, N, conf_thres, nms_thres, MYDET_REC_TOPK, scratch, stream); }
    p.count = records + MYDET_REC_COUNT;
    p.obox = reinterpret_cast<float *>(records + MYDET_REC_BBOX);
    p.oscore = reinterpret_cast<float *>(records + MYDET_REC_SCORE);
    p.ocls = reinterpret_cast<int64_t *>(records + MYDET_REC_CLASS);
    p.oidx = records + MYDET_REC_INDEX;
    p.count_pad = MYDET_REC_BBOX - 1; p.count_st = MYDET_REC_WORDS; p.obox_st = MYDET_REC_WORDS; p.oscore_st = MYDET_REC_WORDS;
    p.ocls_st = MYDET_REC_WORDS / 2; p.oidx_st = MYDET_REC_WORDS;
    return launch_postprocess(p, bbox, class_idx, score, B, N, conf_thres, nms_thres, MYDET_REC_TOPK, scratch, stream);
}
