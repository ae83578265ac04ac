// Shared helpers for the gfx950 kernels (device-side only; no compatibility layer).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mydet.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MYDET_WAVE 64

static inline int mydet_launch_status() { return (int)hipGetLastError(); }

__device__ __forceinline__ float mydet_sigmoid(float v) { return 1.0f / (1.0f + expf(-v)); }

__device__ __forceinline__ float mydet_act(float v, int act) {
    if (act == MYDET_ACT_LEAKY) return v > 0.0f ? v : v * 0.1f;
    if (act == MYDET_ACT_SWISH) return v * mydet_sigmoid(v);
    return v;
}

// Bijective XCD-aware remap (8 XCDs, blocks dealt round-robin): blocks that land on one
// XCD get a contiguous range of logical ids, so neighbouring tiles share that XCD's L2.
__device__ __forceinline__ int mydet_xcd_remap(int bid, int nblk) {
    const int xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}
