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
// the split-K experiment of round 4 proved bit-exact (profiles/r04_tried/igemm_inkernel_split_sum.diff.txt).
// How the finishing workgroup knows a share is THIS launch's: the buffer has ONE launch counter (its "epoch"); a workgroup reads it
// when it starts and publishes every share value as an 8-byte (value, epoch) pair in ONE store; the finishing workgroup of an image
// accepts pairs of its epoch only; the LAST finishing workgroup of the launch (an atomic count of the finished images) advances the
// counter -- after every share of the launch has been published, i.e. after every workgroup of the launch has read it.  Share words
// have exactly one writer per launch, and a stale pair anywhere in the buffer, whatever layer, batch size or image wrote it, carries
// an older epoch.  (A counter per image slot was the first form: slots that different batch sizes had advanced differently let a
// stale pair of another slot carry the current epoch of this one -- a one-in-hundreds flake of the full test suite.)
// (Two earlier forms.  An arrival counter -- every workgroup waits for its stores, then for its atomic -- doubled the
// depthwise kernels' time: two dependent device-scope round trips at the end of thousands of 8 us workgroups.  An "empty" mark
// that the finishing workgroup put back into every share word after reading it: correct in every single-stream run, and WRONG
// under two batch lanes -- the finishing workgroup's reset and the owner's earlier store of the same launch are two writers
// of one word from different XCDs, the owner's value could land last, and the next layer's finishing workgroup, started
// early because the other lane held the CUs, summed the PREVIOUS layer's share: profiles/HISTORY.md, round 5.)
#pragma once
#include "common.h"

typedef mydet_se_tail SeTail;      // include/mydet.h; .gate == nullptr: no in-launch tail

constexpr int MYDET_SE_MAX_CSE = 96;
constexpr int MYDET_SE_LDS_FLOATS = MYDET_SE_MAX_CSE + 256 + MYDET_SE_MAX_CSE;     // h_acc | phase sums | hidden layer

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

// The share buffer (include/mydet.h: mydet_se_tail.hpart), 8-byte aligned, 32-bit words:
//     [0]                         the launch counter (never 0; 1 when the buffer is made)
//     [1]                         images finished in the running launch (0 between launches)
//     [2]                         finishing workgroups that gave up their poll and wrote a NaN gate (never reset; 0 in a healthy run)
//     [3, MYDET_SE_EPOCH_WORDS)   unused
//     then [B][nwg][Cse] pairs    (value, epoch) of workgroup wg's share of hidden unit o; all zero when the buffer is made
constexpr int MYDET_SE_EPOCHS = MYDET_SE_EPOCH_WORDS;

// the launch's epoch, requested when the workgroup starts (the value is needed only when its share leaves)
__device__ __forceinline__ unsigned se_epoch(const SeTail &t, int b) {
    (void)b;
    return __hip_atomic_load(reinterpret_cast<const unsigned *>(t.hpart), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Called by ALL 256 threads of every workgroup of image b once its h_acc is complete (and a barrier has made it visible).
// wg = this workgroup's index among the nwg workgroups of its image, nimg = images of the launch, epoch = se_epoch(t, b) read by
// this workgroup.
//   * every workgroup but the last of its image: Cse device-scope 8-byte stores, fire and forget -- no wait, no atomic, no fence;
//   * the last workgroup (wg == nwg - 1; within an XCD block ids are dispatched in order, so most of the image's other
//     workgroups are running or done when it starts -- a matter of waiting time only): reads all pairs at device scope until
//     every one carries its epoch (a bounded poll), sums the values in workgroup order (deterministic), finishes the gate and
//     counts its image as finished; the last image of the launch advances the counter.  If the poll gives up (it never should) the gate is NaN, so the failure cannot pass for
//     a result.
__device__ __forceinline__ void se_tail_finish(const SeTail &t, float *lds, int C, int HW, int b, int wg, int nwg, int nimg, unsigned epoch) {
    const int tid = threadIdx.x, Cse = t.Cse;
    unsigned long long *hp = reinterpret_cast<unsigned long long *>(t.hpart + MYDET_SE_EPOCHS) + (int64_t)b * nwg * Cse;
    if (tid < Cse)
        __hip_atomic_store(hp + (int64_t)wg * Cse + tid, ((unsigned long long)epoch << 32) | __builtin_bit_cast(unsigned, lds[tid]),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (wg != nwg - 1) return;
    float *phase = lds + MYDET_SE_MAX_CSE, *hid = phase + 256;
    // hidden layer: the shares of all workgroups in workgroup order (P interleaved chains, combined in chain order)
    const int P = 256 / Cse;                                   // >= 2 (Cse <= 96)
    const int o = tid % Cse, ph = tid / Cse;
    const int Q = C >> 2;
    float sum = 0.f;
    int spins = 0;
    bool ok = true;
    for (;;) {
        int missing = 0;
        sum = 0.f;
        if (ph < P) {
            for (int w0 = ph; w0 < nwg; w0 += 8 * P) {         // 8 independent 8-byte device-scope loads in flight
                unsigned long long v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int w = w0 + u * P;
                    v[u] = __hip_atomic_load(hp + (int64_t)(w < nwg ? w : wg) * Cse + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const bool in = w0 + u * P < nwg;
                    missing |= in && (unsigned)(v[u] >> 32) != epoch;
                    sum += in ? __builtin_bit_cast(float, (unsigned)v[u]) : 0.f;
                }
            }
        }
        if (!__syncthreads_or(missing)) break;
        if (++spins >= (1 << 14)) {                              // (uniform: every thread sees the same vote and count)
            ok = false;                                          // the gate becomes NaN below AND the give-up is counted in the header,
            if (tid == 0) __hip_atomic_fetch_add(reinterpret_cast<unsigned *>(t.hpart) + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;                                               // where ops.se_tail_timeouts / a C caller can see it
        }
        __builtin_amdgcn_s_sleep(2);
    }
    if (tid == 0) {      // this image is finished; the last one of the launch hands the next launch a new epoch (never 0)
        unsigned *hdr = reinterpret_cast<unsigned *>(t.hpart);
        if (__hip_atomic_fetch_add(hdr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)nimg - 1u) {
            __hip_atomic_store(hdr + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(hdr, epoch + 1u ? epoch + 1u : 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (ph < P) phase[ph * Cse + o] = sum;
    __syncthreads();
    if (tid < Cse) {
        float s = phase[tid];
        for (int k = 1; k < P; ++k) s += phase[k * Cse + tid];
        const float v = s * (1.0f / (float)HW) + t.b1[tid];
        hid[tid] = ok ? v * mydet_sigmoid(v) : __builtin_nanf("");
    }
    __syncthreads();
    // expand conv + sigmoid: a thread per channel QUAD (16-byte rows of the transposed weight), k in order, eight rows in flight.
    // One v_fma_f32 per channel.  Round 5's build let the compiler pair these into v_pk_fma_f32, and THAT instruction returned
    // wrong low halves in lanes 48-63 whenever waves of another kernel issued dense bf16 MFMAs on the same SIMD (the other batch
    // lane's split-bf16 convs): reproduced stand-alone by tools/hw_pk_probe.hip, written up in profiles/r06_pk_fma_finding.md.
    // The whole library is built without packed-f32 VALU ops since (csrc/Makefile: NOPK; a CPU test disassembles the library).
    for (int q = tid; q < Q; q += 256) {
        f32x4 e = *reinterpret_cast<const f32x4 *>(t.b2 + q * 4);
        int k = 0;
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
static inline int mydet_se_tail_check(const SeTail &t, int C, int B, int groups) {
    if (!t.gate) return 0;
    if (!t.w1 || !t.b1 || !t.w2t || !t.b2 || !t.hpart) return MYDET_E_BADARG;
    if (((uintptr_t)t.w1 & 15) || ((uintptr_t)t.w2t & 15) || ((uintptr_t)t.b2 & 15) || ((uintptr_t)t.gate & 15) || (C & 3)) return MYDET_E_BADARG;
    if ((uintptr_t)t.hpart & 7) return MYDET_E_BADARG;
    if (t.Cse < 1 || t.Cse > MYDET_SE_MAX_CSE) return MYDET_E_UNSUPP;
    // the share buffer holds the header and one (value, epoch) pair per image, workgroup of the image and hidden unit
    if (B < 1 || groups < 1 || t.hpart_bytes < 4 * ((int64_t)MYDET_SE_EPOCH_WORDS + 2 * (int64_t)B * groups * t.Cse)) return MYDET_E_BADARG;
    return 0;
}
