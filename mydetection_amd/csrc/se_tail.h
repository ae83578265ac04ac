// Squeeze-excite tail INSIDE the depthwise launch (external/efficientnet/model.py:80-83):
//     gate = sigmoid(W2 . swish(W1 . mean(y) + b1) + b2),   y = the depthwise launch's own output
// As its own launch the tail sat on the critical path of every MBConv block (depthwise -> gate -> project conv): 23 launches
// of one workgroup per image, 7-22 us each, a chain of dependent round trips on operands other XCDs had just written.
// Here the work is spread over the launch that produces y:
//   * the reduce conv is linear in the channel sums, so every workgroup adds ITS channels' share
//         h_wg[o] = sum_{c in its channels} W1[o][c] * (sum over its pixels of y[c])
//     right after it formed those sums (se_fc1_accumulate; the weights do not depend on the data) and publishes the Cse
//     values with device-scope stores;
//   * the LAST workgroup of an image (by block id) waits until every share of its image is there, adds them in workgroup
//     order (fixed order: deterministic), applies 1 / HW, bias and swish, and runs the expand conv + sigmoid for the image's C
//     channels.  Nobody else waits for anything: a share is Cse fire-and-forget stores.
// Stores / loads that cross workgroups carry the sc1 (device) scope, so no L2 write-back or invalidate is needed -- the idiom
// the split-K experiment of round 4 proved bit-exact (profiles/r04_tried/igemm_inkernel_split_sum.diff.txt).  The share
// buffer holds an "empty" mark in every float between launches; the finishing workgroup puts it back.
// (A first form with an arrival counter -- every workgroup waits for its stores, then for its atomic -- doubled the depthwise
// kernels' time: two dependent device-scope round trips at the end of thousands of 8 us workgroups.)
#pragma once
#include "common.h"

typedef mydet_se_tail SeTail;      // include/mydet.h; .gate == nullptr: no in-launch tail

constexpr int MYDET_SE_MAX_CSE = 96;
constexpr int MYDET_SE_LDS_FLOATS = MYDET_SE_MAX_CSE + 256 + MYDET_SE_MAX_CSE;     // h_acc | phase sums | hidden layer

__device__ __forceinline__ void mydet_store_dev(float *p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float mydet_load_dev(const float *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// h_acc[o] += sum_{j < nc} W1[o][c0 + j] * tot[j] for every o < Cse; tot: LDS, channel sums of channels c0 .. c0 + nc - 1
// (c0 % 4 == 0).  256 threads: thread (o = tid / 4, part = tid % 4) takes 8 of every 32 channels, the four parts are added in
// lane order.  h_acc[o] is owned by one thread, so successive calls of a workgroup accumulate in program order.
// The caller puts a barrier between writing tot / the previous call and this one, and after the last call.
__device__ __forceinline__ void se_fc1_accumulate(const SeTail &t, int C, const float *tot, int c0, int nc, float *h_acc) {
    const int tid = threadIdx.x, part = tid & 3;
    for (int o0 = 0; o0 < t.Cse; o0 += 64) {
        const int o = o0 + (tid >> 2);
        const bool ov = o < t.Cse;
        const float *wrow = t.w1 + (int64_t)(ov ? o : 0) * C + c0;
        float acc = 0.f;
        for (int cc = 0; cc < nc; cc += 32) {
            const int j0 = cc + part * 8;
            f32x4 wa = {0.f, 0.f, 0.f, 0.f}, wb = wa;
            if (j0 < nc) wa = *reinterpret_cast<const f32x4 *>(wrow + j0);
            if (j0 + 4 < nc) wb = *reinterpret_cast<const f32x4 *>(wrow + j0 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc = fmaf(wa[e], j0 + e < nc ? tot[j0 + e] : 0.f, acc);
                acc = fmaf(wb[e], j0 + 4 + e < nc ? tot[j0 + 4 + e] : 0.f, acc);
            }
        }
        acc += __shfl_xor(acc, 1);
        acc += __shfl_xor(acc, 2);
        if (ov && part == 0) h_acc[o] += acc;
    }
}

// Called by ALL 256 threads of every workgroup of image b once its h_acc is complete (and a barrier has made it visible).
// wg = this workgroup's index among the nwg workgroups of its image (in dispatch order: the kernels number their blocks image
// by image), hpart = the image-major share buffer [B][nwg][Cse] whose every float holds MYDET_SE_EMPTY between launches.
//   * every workgroup but the last of its image: Cse device-scope stores, fire and forget -- no wait, no atomic, no fence;
//   * the last workgroup (wg == nwg - 1; block ids are dispatched in order, so every other workgroup of the image is running
//     or done when it starts): reads all shares at device scope until none is empty (a bounded poll: workgroups that are
//     still finishing), puts the empty mark back for the next launch, and finishes the gate.  Sums in workgroup order:
//     deterministic.  If the poll gives up (it never should) the gate is NaN, so the failure cannot pass for a result.
constexpr unsigned MYDET_SE_EMPTY = 0x7FC5E5E5u;      // a quiet NaN with a payload no arithmetic produces

__device__ __forceinline__ void se_tail_finish(const SeTail &t, float *lds, int C, int HW, int b, int wg, int nwg) {
    const int tid = threadIdx.x, Cse = t.Cse;
    float *hp = t.hpart + (int64_t)b * nwg * Cse;
    if (tid < Cse) mydet_store_dev(hp + (int64_t)wg * Cse + tid, lds[tid]);
    if (wg != nwg - 1) return;
    float *phase = lds + MYDET_SE_MAX_CSE, *hid = phase + 256;
    // hidden layer: the shares of all workgroups in workgroup order (P interleaved chains, combined in chain order)
    const int P = 256 / Cse;                                   // >= 2 (Cse <= 96)
    const int o = tid % Cse, ph = tid / Cse;
    const float empty = __builtin_bit_cast(float, MYDET_SE_EMPTY);
    // the expand conv's operands do not depend on the data: bias and the first eight rows of this thread's first channel quad
    // are requested before the wait for the shares (consumed in place of the same loads below)
    const int Q = C >> 2;
    f32x4 pw[8], pb2 = {0.f, 0.f, 0.f, 0.f};
    {
        const int q = tid < Q ? tid : 0;
        pb2 = *reinterpret_cast<const f32x4 *>(t.b2 + q * 4);
#pragma unroll
        for (int j = 0; j < 8; ++j) pw[j] = *reinterpret_cast<const f32x4 *>(t.w2t + (int64_t)(j < Cse ? j : 0) * C + q * 4);
    }
    float sum = 0.f;
    int spins = 0;
    bool ok = true;
    for (;;) {
        int missing = 0;
        sum = 0.f;
        if (ph < P) {
            for (int w0 = ph; w0 < nwg; w0 += 16 * P) {        // 16 independent device-scope loads in flight
                float v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int w = w0 + u * P;
                    v[u] = w < nwg ? mydet_load_dev(hp + (int64_t)w * Cse + o) : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    missing |= __builtin_bit_cast(unsigned, v[u]) == MYDET_SE_EMPTY;
                    sum += v[u];
                }
            }
        }
        if (!__syncthreads_or(missing)) break;
        if (++spins >= (1 << 14)) { ok = false; break; }        // (uniform: every thread sees the same vote and count)
        __builtin_amdgcn_s_sleep(2);
    }
    if (ph < P) {
        for (int w = ph; w < nwg; w += P) mydet_store_dev(hp + (int64_t)w * Cse + o, empty);      // ready for the next launch
        phase[ph * Cse + o] = sum;
    }
    __syncthreads();
    if (tid < Cse) {
        float s = phase[tid];
        for (int k = 1; k < P; ++k) s += phase[k * Cse + tid];
        const float v = s * (1.0f / (float)HW) + t.b1[tid];
        hid[tid] = ok ? v * mydet_sigmoid(v) : __builtin_nanf("");
    }
    __syncthreads();
    // expand conv + sigmoid: a thread per channel QUAD (16-byte rows of the transposed weight), k in order, eight rows in flight
    for (int q = tid; q < Q; q += 256) {
        f32x4 e = q == tid ? pb2 : *reinterpret_cast<const f32x4 *>(t.b2 + q * 4);
        int k = 0;
        if (q == tid && Cse >= 8) {                    // the prefetched first eight rows
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int c = 0; c < 4; ++c) e[c] = fmaf(pw[j][c], hid[j], e[c]);
            k = 8;
        }
        for (; k + 7 < Cse; k += 8) {
            f32x4 w[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) w[j] = *reinterpret_cast<const f32x4 *>(t.w2t + (int64_t)(k + j) * C + q * 4);
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int c = 0; c < 4; ++c) e[c] = fmaf(w[j][c], hid[k + j], e[c]);
        }
        for (; k < Cse; ++k) {
            const f32x4 w = *reinterpret_cast<const f32x4 *>(t.w2t + (int64_t)k * C + q * 4);
#pragma unroll
            for (int c = 0; c < 4; ++c) e[c] = fmaf(w[c], hid[k], e[c]);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) e[c] = mydet_sigmoid(e[c]);
        *reinterpret_cast<f32x4 *>(t.gate + (int64_t)b * C + q * 4) = e;
    }
}

// host side: arguments of an in-launch tail are complete and inside the kernels' limits
static inline int mydet_se_tail_check(const SeTail &t, int C) {
    if (!t.gate) return 0;
    if (!t.w1 || !t.b1 || !t.w2t || !t.b2 || !t.hpart) return MYDET_E_BADARG;
    if (((uintptr_t)t.w1 & 15) || ((uintptr_t)t.w2t & 15) || ((uintptr_t)t.b2 & 15) || ((uintptr_t)t.gate & 15) || (C & 3)) return MYDET_E_BADARG;
    if (t.Cse < 1 || t.Cse > MYDET_SE_MAX_CSE) return MYDET_E_UNSUPP;
    return 0;
}
