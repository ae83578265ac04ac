// Shared helpers for the gfx950 kernels (device-side only; no compatibility layer).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mydet.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MYDET_WAVE 64

static inline int mydet_launch_status() { return (int)hipGetLastError(); }

// Compute units of the current device (256 on MI355X), read once.
static inline int mydet_cu_count() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
            n = v;
        else
            n = 256;
    }
    return n;
}

// True the first time it is called for the CURRENT device with this `mask` (one bit per device ordinal): function
// attributes such as the > 64 KiB dynamic-LDS opt-in are per device, so a process that drives several devices must
// set them on each one (one process per GPU is the deployment, but nothing here relies on it).
static inline bool mydet_first_on_device(unsigned long long &mask) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return true;      // unknown ordinal: always opt in
    const unsigned long long bit = 1ull << dev;
    if (mask & bit) return false;
    mask |= bit;
    return true;
}

// The > 64 KiB dynamic-LDS opt-in of `kern`, once per device (`mask` as above): 0, or MYDET_E_UNSUPP when the runtime
// refuses it -- the device is then forgotten again, so the next call retries instead of launching into a later
// "invalid value" error.
template <typename K>
static inline int mydet_lds_opt_in(unsigned long long &mask, K kern, int bytes) {
    if (!mydet_first_on_device(mask)) return 0;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess) return 0;
    (void)hipGetLastError();                                   // the failure is reported by the return code, not left sticky
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) mask &= ~(1ull << dev);
    return MYDET_E_UNSUPP;
}

__device__ __forceinline__ float mydet_sigmoid(float v) { return 1.0f / (1.0f + expf(-v)); }

// Logistic for the swish epilogues of the conv / depthwise / fusion kernels: v_exp_f32 on -v*log2(e) (product formed
// with an fma correction term) and v_rcp_f32 -- 5 instructions instead of ~25.  The hardware ops are 1 ulp; the float32
// exponent argument limits exp to |t|*2^-24 relative, which for the logistic is below 3e-8 absolute for v >= 0 and
// below 6e-7 RELATIVE at v = -10 (where the value is 4.5e-5): swish(v) stays within ~1 ulp of the libm form.
// The decode / SE-gate kernels keep the libm form (score parity and sigmoid collisions are defined by it).
__device__ __forceinline__ float mydet_sigmoid_fast(float v) {
    const float nv = -v;
    const float t = fmaf(nv, 1.44269502162933349609375f, nv * 1.925963033500011079e-8f);   // -v * log2(e), hi + lo
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t));
}

__device__ __forceinline__ float mydet_act(float v, int act) {
    if (act == MYDET_ACT_LEAKY) return v > 0.0f ? v : v * 0.1f;
    if (act == MYDET_ACT_SWISH) return v * mydet_sigmoid_fast(v);
    return v;
}

// Workgroup barrier that orders LDS traffic only: global loads issued before it stay in flight across it (a plain
// __syncthreads() drains them with s_waitcnt vmcnt(0)), which is what lets a kernel prefetch its next tile into
// registers while the current one is computed.
__device__ __forceinline__ void mydet_lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// Bijective XCD-aware remap (8 XCDs, blocks dealt round-robin): blocks that land on one
// XCD get a contiguous range of logical ids, so neighbouring tiles share that XCD's L2.
__device__ __forceinline__ int mydet_xcd_remap(int bid, int nblk) {
    const int xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

// pointwise.hip: LDS-free-x 1x1 conv (Cin <= 240, Cout % 4 == 0); MYDET_E_UNSUPP otherwise.  Library-internal.
int mydet_pw_skinny(const float *x, int64_t ldx, const float *w, const float *scale, const float *shift,
                    const float *residual, int64_t ldr, const float *gate, float *y, int64_t ldy, int B, int HW, int Cin,
                    int Cout, int act, void *stream);
