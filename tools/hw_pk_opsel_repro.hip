// Minimal reproducer (MI355X / gfx950, ROCm 7.2): v_pk_fma_f32 with op_sel:[0,1,0] -- the LOW result half taking the HIGH dword
// of src1 -- returns wrong low halves in lanes 48-63 while waves of ANOTHER kernel issue back-to-back independent
// v_mfma_f32_16x16x32_bf16 on the same SIMDs.  Registers only: no memory or LDS traffic in either loop.  The same instruction
// without the modifier, or with op_sel_hi:[1,0,1] only, is never wrong; neither is v_fma_f32.   profiles/r06_pk_fma_finding.md
//   hipcc --offload-arch=gfx950 -O3 -o repro tools/hw_pk_opsel_repro.hip && ./repro
// Expected on an affected part:   op_sel:[0,1,0]  alone 0 wrong | beside the MFMA kernel: millions wrong, all in lanes 48-63, x / z only
//                                 no modifier     alone 0 wrong | beside the MFMA kernel: 0 wrong
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// e.xy += w.xy * h, e.zw += w.zw * h for 8 (w, h) pairs per iteration; h sits in the HIGH (OPSEL) or in both dwords of its pair
template <bool OPSEL>
__global__ __launch_bounds__(256) void victim(const float *w, const float *ref, float *out, unsigned *hist, int rounds) {
    const int q = blockIdx.x * 256 + threadIdx.x, nq = gridDim.x * 256;
    f32x2 wl[8], wh[8], hp[8];
    for (int j = 0; j < 8; ++j) {
        const f32x4 t = *reinterpret_cast<const f32x4 *>(w + ((size_t)j * nq + q) * 4);
        const float h = 0.37f + 0.01f * j;
        wl[j] = f32x2{t[0], t[1]}; wh[j] = f32x2{t[2], t[3]};
        hp[j] = OPSEL ? f32x2{-h, h} : f32x2{h, h};
    }
    float rv[4] = {0.f, 0.f, 0.f, 0.f};
    if (ref) for (int c = 0; c < 4; ++c) rv[c] = ref[(size_t)q * 4 + c];
    unsigned cnt[4] = {0, 0, 0, 0};
    float es[4] = {0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < rounds; ++r) {
        f32x2 lo = {0.25f, -0.5f}, hi = {0.125f, 0.75f};
        for (int k = 0; k < 12; ++k)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (OPSEL) {
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(lo) : "v"(wl[j]), "v"(hp[j]));
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(hi) : "v"(wh[j]), "v"(hp[j]));
                } else {
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(lo) : "v"(wl[j]), "v"(hp[j]));
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(hi) : "v"(wh[j]), "v"(hp[j]));
                }
            }
        es[0] = lo[0]; es[1] = lo[1]; es[2] = hi[0]; es[3] = hi[1];
        asm volatile("" : "+v"(es[0]), "+v"(es[1]), "+v"(es[2]), "+v"(es[3]));
        if (ref) for (int c = 0; c < 4; ++c) cnt[c] += __builtin_bit_cast(unsigned, es[c]) != __builtin_bit_cast(unsigned, rv[c]);
    }
    for (int c = 0; c < 4; ++c) {
        out[(size_t)q * 4 + c] = es[c];
        if (cnt[c]) atomicAdd(hist + (threadIdx.x & 63) * 4 + c, cnt[c]);
    }
}

__global__ __launch_bounds__(256) void mfma_busy(float *sink, int iters) {           // six independent accumulators: the pipe never waits
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.3f + 0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.5f - 0.002f * (threadIdx.x * 3 + i)); }
    f32x4 acc[6];
    for (int i = 0; i < 6; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 6; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    float s = 0.f;
    for (int i = 0; i < 6; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) sink[0] = s;
}

template <bool OPSEL>
static void run(const char *name, const float *w, float *ref, float *out, unsigned *hist, float *sink, hipStream_t s1, hipStream_t s2) {
    const int WGS = 64, rounds = 20000;
    hipLaunchKernelGGL(victim<OPSEL>, dim3(WGS), dim3(256), 0, s1, w, (const float *)nullptr, ref, hist, 1);      // reference: alone on the chip
    CHECK(hipDeviceSynchronize());
    for (int beside = 0; beside < 2; ++beside) {
        CHECK(hipMemset(hist, 0, 1024));
        if (beside) hipLaunchKernelGGL(mfma_busy, dim3(768), dim3(256), 0, s2, sink, 150000);                      // ~20 ms, three workgroups per CU
        hipLaunchKernelGGL(victim<OPSEL>, dim3(WGS), dim3(256), 0, s1, w, (const float *)ref, out, hist, rounds);  // ~25-50 ms
        CHECK(hipDeviceSynchronize());
        unsigned h[256], lanes[4] = {0, 0, 0, 0}, comp[4] = {0, 0, 0, 0};
        CHECK(hipMemcpy(h, hist, 1024, hipMemcpyDeviceToHost));
        for (int i = 0; i < 256; ++i) { lanes[i / 64] += h[i]; comp[i & 3] += h[i]; }
        printf("%-16s %-28s wrong results by lane group 0-15 / 16-31 / 32-47 / 48-63: %u %u %u %u   by component x y z w: %u %u %u %u   (of %llu)\n", name,
               beside ? "beside 16x16x32 bf16 MFMAs:" : "alone:", lanes[0], lanes[1], lanes[2], lanes[3], comp[0], comp[1], comp[2], comp[3],
               (unsigned long long)rounds * WGS * 256 * 4);
    }
}

int main() {
    const size_t n = (size_t)8 * 64 * 256 * 4;
    float *hw = (float *)malloc(n * 4), *w, *ref, *out, *sink;
    unsigned *hist;
    srand(1);
    for (size_t i = 0; i < n; ++i) hw[i] = (rand() / (float)RAND_MAX - 0.5f) * 0.6f;
    CHECK(hipMalloc(&w, n * 4)); CHECK(hipMalloc(&ref, 64 * 256 * 16)); CHECK(hipMalloc(&out, 64 * 256 * 16)); CHECK(hipMalloc(&sink, 64)); CHECK(hipMalloc(&hist, 1024));
    CHECK(hipMemcpy(w, hw, n * 4, hipMemcpyHostToDevice));
    hipStream_t s1, s2;
    CHECK(hipStreamCreate(&s1)); CHECK(hipStreamCreate(&s2));
    run<true>("op_sel:[0,1,0]", w, ref, out, hist, sink, s1, s2);
    run<false>("no modifier", w, ref, out, hist, sink, s1, s2);
    return 0;
}
