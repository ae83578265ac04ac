// Stand-alone check (no torch): does a chain of dependent v_pk_fma_f32 in one wave give the results of the same chain of
// v_fma_f32 while waves of ANOTHER kernel on another stream keep the matrix pipe of the same SIMDs busy?
// Found through the squeeze-excite tail (csrc/se_tail.h): its expand conv, compiled to v_pk_fma_f32, produced wrong low halves
// in lanes 48-63 whenever the other batch lane's split-bf16 convs (v_mfma_f32_32x32x16_bf16) ran on the same CUs.
//   hipcc --offload-arch=gfx950 -O2 -o hw_pk_fma_vs_mfma tools/hw_pk_fma_vs_mfma.hip && ./hw_pk_fma_vs_mfma
// Prints, per co-running kernel (none / bf16 MFMA / fp32 MFMA / plain VALU), how many results of the probe loop (fmaf on float4,
// which hipcc turns into v_pk_fma_f32 -- check with --save-temps) differ from the same loop run alone on the chip.
// Result on MI355X (round 5): 0 in every case, also with -DPROBE_FAT (a 174-VGPR victim) and with the aggressor that breaks the real
// kernel outside the model (six independent v_mfma_f32_16x16x32_bf16: tools/r05_pk_repro.py) -- the stand-alone victim does NOT reproduce it.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// the compiler's own code for the same loop (what se_tail.h had): fmaf on float4, left to form packed ops
__global__ __launch_bounds__(256) void probe_compiled_kernel(const float *w, const float *h, int rounds, int K, float *out) {
    __shared__ __attribute__((aligned(16))) float hs[1024];
    for (int i = threadIdx.x; i < K; i += 256) hs[i] = h[i];
    __syncthreads();
    const int q = blockIdx.x * 256 + threadIdx.x;
    f32x4 e = {0.f, 0.f, 0.f, 0.f};
#ifdef PROBE_FAT          // ~170 VGPRs like the finishing workgroup of the squeeze-excite tail: 30 float4 kept live across the loop
    f32x4 fat[30];
#pragma unroll
    for (int i = 0; i < 30; ++i) fat[i] = *reinterpret_cast<const f32x4 *>(w + ((size_t)i * gridDim.x * 256 + q) * 4);
#pragma unroll
    for (int i = 0; i < 30; ++i) asm volatile("" : "+v"(fat[i]));
#endif
    for (int r = 0; r < rounds; ++r) {
        e = f32x4{0.25f, -0.5f, 0.125f, 0.75f};
        for (int k = 0; k < K; k += 8) {
            f32x4 wv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) wv[j] = *reinterpret_cast<const f32x4 *>(w + ((size_t)(k + j) * gridDim.x * 256 + q) * 4);
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int c = 0; c < 4; ++c) e[c] = fmaf(wv[j][c], hs[k + j], e[c]);
        }
        asm volatile("" : "+v"(e));
    }
#ifdef PROBE_FAT
#pragma unroll
    for (int i = 0; i < 30; ++i) asm volatile("" : "+v"(fat[i]));
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 30; ++i) acc += fat[i];
    if (acc[0] == 12345.678f) e[0] += acc[1];
#endif
    *reinterpret_cast<f32x4 *>(out + (size_t)q * 4) = e;
}

template <int KIND>      // 0: v_mfma_f32_32x32x16_bf16, 1: v_mfma_f32_32x32x2_f32, 2: plain VALU fma, 3: SIX independent v_mfma_f32_16x16x32_bf16
__global__ __launch_bounds__(256) void busy_kernel(float *sink, int iters, float seed) {
    extern __shared__ char busy_lds[];              // 72 KB requested at launch: two workgroups per CU, as the split-bf16 conv
    if (iters < 0) busy_lds[threadIdx.x] = 1;
    f32x16 acc0, acc1;
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + 0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.5f - 0.002f * (threadIdx.x * 3 + i)); }
    float fa = seed + 0.001f * threadIdx.x, fb = 0.5f - 0.002f * threadIdx.x;
    float v0 = fa, v1 = fb, v2 = fa * 0.5f, v3 = fb * 0.25f;
    f32x4 a4[6];
    for (int q = 0; q < 6; ++q) a4[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        if (KIND == 3) {
#pragma unroll
            for (int q = 0; q < 6; ++q) a4[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, a4[q], 0, 0, 0);
        } else if (KIND == 0) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc1, 0, 0, 0);
        } else if (KIND == 1) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fb, fa, acc1, 0, 0, 0);
        } else {
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v0) : "v"(fa), "v"(fb));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v1) : "v"(fb), "v"(fa));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v2) : "v"(fa), "v"(fa));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v3) : "v"(fb), "v"(fb));
        }
    }
    float s = v0 + v1 + v2 + v3;
    for (int q = 0; q < 6; ++q) s += a4[q][0] + a4[q][1] + a4[q][2] + a4[q][3];
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    if (s == 12345.678f) sink[0] = s;
}

int main() {
    // ONE probe workgroup per CU (one probe wave per SIMD, as the finishing workgroup of the squeeze-excite tail), the busy kernel's
    // workgroups (two per CU by their LDS request) beside it
    const int WGS = 256, K = 96, rounds = 2000;
    const size_t nw = (size_t)K * WGS * 256 * 4;
    std::vector<float> hw(nw), hh(1024);
    srand(1);
    for (auto &v : hw) v = (rand() / (float)RAND_MAX - 0.5f) * 0.6f;
    for (auto &v : hh) v = (rand() / (float)RAND_MAX - 0.5f);
    float *w, *h, *sink, *out, *out_ref;
    CHECK(hipMalloc(&w, nw * 4)); CHECK(hipMalloc(&h, 1024 * 4)); CHECK(hipMalloc(&sink, 64));
    CHECK(hipMalloc(&out, (size_t)WGS * 256 * 16)); CHECK(hipMalloc(&out_ref, (size_t)WGS * 256 * 16));
    CHECK(hipMemcpy(w, hw.data(), nw * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(h, hh.data(), 1024 * 4, hipMemcpyHostToDevice));
    CHECK(hipFuncSetAttribute((const void *)busy_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024));
    CHECK(hipFuncSetAttribute((const void *)busy_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024));
    CHECK(hipFuncSetAttribute((const void *)busy_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024));
    CHECK(hipFuncSetAttribute((const void *)busy_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024));
    hipStream_t s1, s2;
    CHECK(hipStreamCreate(&s1)); CHECK(hipStreamCreate(&s2));
    // reference of the compiled form: alone on the chip
    hipLaunchKernelGGL(probe_compiled_kernel, dim3(WGS), dim3(256), 0, s1, w, h, 1, K, out_ref);
    CHECK(hipDeviceSynchronize());
    std::vector<float> ref((size_t)WGS * 256 * 4), got(ref.size());
    CHECK(hipMemcpy(ref.data(), out_ref, ref.size() * 4, hipMemcpyDeviceToHost));
    const char *names[] = {"nothing else running", "v_mfma_f32_32x32x16_bf16 on the other stream", "v_mfma_f32_32x32x2_f32 on the other stream", "v_fma_f32 on the other stream",
                           "SIX independent v_mfma_f32_16x16x32_bf16 on the other stream"};
    for (int co = 0; co < 5; ++co) {
        unsigned cmp_bad = 0;
        int cmp_lane[64] = {0};
        for (int rep = 0; rep < 20; ++rep) {
            if (co == 1) hipLaunchKernelGGL(busy_kernel<0>, dim3(1024), dim3(256), 72 * 1024, s2, sink, 60000, 0.3f);
            if (co == 2) hipLaunchKernelGGL(busy_kernel<1>, dim3(1024), dim3(256), 72 * 1024, s2, sink, 15000, 0.3f);
            if (co == 3) hipLaunchKernelGGL(busy_kernel<2>, dim3(1024), dim3(256), 72 * 1024, s2, sink, 200000, 0.3f);
            if (co == 4) hipLaunchKernelGGL(busy_kernel<3>, dim3(1024), dim3(256), 48 * 1024, s2, sink, 40000, 0.3f);
            for (int i = 0; i < 4; ++i) hipLaunchKernelGGL(probe_compiled_kernel, dim3(WGS), dim3(256), 0, s1, w, h, rounds / 20, K, out);
            CHECK(hipDeviceSynchronize());
            CHECK(hipMemcpy(got.data(), out, got.size() * 4, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < got.size(); ++i)
                if (__builtin_bit_cast(unsigned, got[i]) != __builtin_bit_cast(unsigned, ref[i])) { ++cmp_bad; ++cmp_lane[(i / 4) & 63]; }
        }
        printf("%-46s | compiled fmaf loop (v_pk_fma_f32): %u values differ from its run alone\n", names[co], cmp_bad);
        if (cmp_bad) {
            printf("   differing values by lane:");
            for (int l = 0; l < 64; ++l) if (cmp_lane[l]) printf(" %d:%d", l, cmp_lane[l]);
            printf("\n");
        }
    }
    return 0;
}
