// Stand-alone reproducer hunt (no torch, no libmydet) for the round-5 finding: the squeeze-excite tail's expand conv, compiled to
// v_pk_fma_f32, returned wrong LOW halves in lanes 48-63 while waves of ANOTHER kernel ran dense bf16 MFMAs on the same SIMDs
// (profiles/r05_se_tail_debug.txt).  Round 5's probe (tools/hw_pk_fma_vs_mfma.hip) showed nothing -- but it compared only the LAST
// round of its LAST launch, ~10 ms after the 5 ms aggressor had drained.  This one checks EVERY round inside the kernel (bit compare
// with the same form's result computed alone on the chip), proves the overlap with wall-clock stamps, and runs the victim loop in
// several forms so that one ingredient flips per line of output:
//   form 0  fmaf on float4 left to the compiler (v_pk_fma_f32 with op_sel broadcasts, ds_read_b128-fed, 8 loads in flight)
//   form 1  the same loop, explicit v_fma_f32                                     (control: the shipped form)
//   form 2  explicit v_pk_mul_f32 + v_pk_add_f32                                  (is it the fma or any packed op?)
//   form 3  registers only: v_pk_fma_f32 on operands loaded once, no memory / LDS traffic inside the loop
//   form 4  registers only, explicit v_fma_f32                                    (control)
//   forms 5-7  memory-fed, explicit v_pk_fma_f32: 5 no modifier (src1 a real {h, h} pair), 6 op_sel_hi:[1,0,1] only (both halves
//              take the LOW dword of src1), 7 op_sel:[0,1,0] only (both halves take the HIGH dword of src1)
//   forms 8-9  registers only, explicit v_pk_fma_f32: 8 op_sel:[0,1,0], 9 no modifier
//   hipcc --offload-arch=gfx950 -O3 -o hw_pk_probe tools/hw_pk_probe.hip && ./hw_pk_probe
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int K = 96;               // hidden units of the expand conv (Cse of the 1152- and 1920-channel blocks: 48 / 80)

template <int FORM>
__global__ __launch_bounds__(256) void victim_kernel(const float *w, const float *h, const float *ref, float *out, unsigned *hist,
                                                     int rounds, unsigned long long *stamp, float *first_wrong) {
    __shared__ __attribute__((aligned(16))) float hs[K];
    for (int i = threadIdx.x; i < K; i += 256) hs[i] = h[i];
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0) stamp[0] = wall_clock64();
    const int q = blockIdx.x * 256 + threadIdx.x, nq = gridDim.x * 256;
    float rv[4] = {0.f, 0.f, 0.f, 0.f};          // (scalars: __builtin_bit_cast of an ext-vector ELEMENT reads element 0 with this hipcc)
    if (ref) { const f32x4 t = *reinterpret_cast<const f32x4 *>(ref + (size_t)q * 4); rv[0] = t[0]; rv[1] = t[1]; rv[2] = t[2]; rv[3] = t[3]; }
    unsigned cnt[4] = {0, 0, 0, 0};
    bool seen = false;
    f32x4 e = {0.f, 0.f, 0.f, 0.f};
    f32x4 wr[8];
    if (FORM == 3 || FORM == 4 || FORM >= 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) wr[j] = *reinterpret_cast<const f32x4 *>(w + ((size_t)j * nq + q) * 4);
    }
    for (int r = 0; r < rounds; ++r) {
        e = f32x4{0.25f, -0.5f, 0.125f, 0.75f};
        for (int k = 0; k < K; k += 8) {
            f32x4 wv[8];
            float hv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                constexpr bool MEM = FORM < 3 || (FORM >= 5 && FORM <= 7);
                if (MEM) wv[j] = *reinterpret_cast<const f32x4 *>(w + ((size_t)(k + j) * nq + q) * 4);
                else { wv[j] = wr[j]; asm volatile("" : "+v"(wv[j])); }
                hv[j] = MEM ? hs[k + j] : 0.37f + 0.01f * (float)j;
                if (!MEM) asm volatile("" : "+v"(hv[j]));
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (FORM == 0 || FORM == 3) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) e[c] = fmaf(wv[j][c], hv[j], e[c]);
                } else if (FORM == 1 || FORM == 4) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(e[c]) : "v"(wv[j][c]), "v"(hv[j]));
                } else if (FORM >= 5) {
                    f32x2 lo = {e[0], e[1]}, hi = {e[2], e[3]}, wl = {wv[j][0], wv[j][1]}, wh = {wv[j][2], wv[j][3]};
                    if (FORM == 5 || FORM == 9) {
                        f32x2 hp = {hv[j], hv[j]};
                        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(lo) : "v"(wl), "v"(hp));
                        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(hi) : "v"(wh), "v"(hp));
                    } else if (FORM == 6) {
                        f32x2 hp = {hv[j], hv[j ^ 1]};
                        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(lo) : "v"(wl), "v"(hp));
                        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(hi) : "v"(wh), "v"(hp));
                    } else {
                        f32x2 hp = {hv[j ^ 1], hv[j]};
                        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(lo) : "v"(wl), "v"(hp));
                        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(hi) : "v"(wh), "v"(hp));
                    }
                    e = f32x4{lo[0], lo[1], hi[0], hi[1]};
                } else {
                    f32x2 hh = {hv[j], hv[j]}, lo = {e[0], e[1]}, hi = {e[2], e[3]}, wl = {wv[j][0], wv[j][1]}, wh = {wv[j][2], wv[j][3]}, t0, t1;
                    asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t0) : "v"(wl), "v"(hh));
                    asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t1) : "v"(wh), "v"(hh));
                    asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(lo) : "v"(t0));
                    asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(hi) : "v"(t1));
                    e = f32x4{lo[0], lo[1], hi[0], hi[1]};
                }
            }
        }
        float es[4] = {e[0], e[1], e[2], e[3]};
        asm volatile("" : "+v"(es[0]), "+v"(es[1]), "+v"(es[2]), "+v"(es[3]));      // four scalars: one compare per component
        if (ref) {
#pragma unroll
            for (int c = 0; c < 4; ++c) cnt[c] += __builtin_bit_cast(unsigned, es[c]) != __builtin_bit_cast(unsigned, rv[c]);
            if (first_wrong && cnt[0] + cnt[1] + cnt[2] + cnt[3] != 0 && !seen) { seen = true; *reinterpret_cast<f32x4 *>(first_wrong + (size_t)q * 4) = f32x4{es[0], es[1], es[2], es[3]}; }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) stamp[1] = wall_clock64();
    *reinterpret_cast<f32x4 *>(out + (size_t)q * 4) = e;
#pragma unroll
    for (int c = 0; c < 4; ++c)
        if (cnt[c]) atomicAdd(hist + (threadIdx.x & 63) * 4 + c, cnt[c]);
}

// aggressors: 1 six independent v_mfma_f32_16x16x32_bf16 (the one that broke the real kernel), 2 six independent 32x32x16 bf16,
// 3 six independent v_mfma_f32_32x32x2_f32, 4 two DEPENDENT chains of 16x16x32 bf16, 5 plain v_fma_f32
template <int KIND>
__global__ __launch_bounds__(256) void busy_kernel(float *sink, int iters, float seed, unsigned long long *stamp) {
    extern __shared__ char busy_lds[];
    if (iters < 0) busy_lds[threadIdx.x] = 1;
    if (blockIdx.x == 0 && threadIdx.x == 0) stamp[0] = wall_clock64();
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + 0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.5f - 0.002f * (threadIdx.x * 3 + i)); }
    float fa = seed + 0.001f * threadIdx.x, fb = 0.5f - 0.002f * threadIdx.x;
    f32x4 a4[6];
    f32x16 a16[6];
    float v[4] = {fa, fb, fa * 0.5f, fb * 0.25f};
    for (int q = 0; q < 6; ++q) { a4[q] = f32x4{0.f, 0.f, 0.f, 0.f}; for (int i = 0; i < 16; ++i) a16[q][i] = 0.f; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            if (KIND == 1) a4[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, a4[q], 0, 0, 0);
            if (KIND == 2) a16[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, a16[q], 0, 0, 0);
            if (KIND == 3) a16[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, a16[q], 0, 0, 0);
            if (KIND == 4) a4[q & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, a4[q & 1], 0, 0, 0);
            if (KIND == 5) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[q & 3]) : "v"(fa), "v"(fb));
        }
    }
    float s = v[0] + v[1] + v[2] + v[3];
    for (int q = 0; q < 6; ++q) { s += a4[q][0] + a4[q][1] + a4[q][2] + a4[q][3]; for (int i = 0; i < 16; ++i) s += a16[q][i]; }
    if (s == 12345.678f) sink[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) stamp[1] = wall_clock64();
}

typedef void (*victim_fn)(const float *, const float *, const float *, float *, unsigned *, int, unsigned long long *, float *);
typedef void (*busy_fn)(float *, int, float, unsigned long long *);

// registers-only op_sel:[0,1,0] form: every k-iteration repeats the same 8 terms  e[c] += w[j][c] * hp[1], hp = {h[j ^ 1], h[j]},
// h[j] = 0.37 + 0.01 j.  If ONE instruction of the round took the LOW dword of src1 instead (op_sel ignored), the result is off by
// w[j][c] * (h[j ^ 1] - h[j]) = -+0.01 w[j][c] in the halves it computed: name j and the halves for a few wrong lanes.
static void explain(const std::vector<float> &hw, size_t nq, const float *ref_d, const float *fw_d) {
    std::vector<float> ref(nq * 4), fw(nq * 4);
    CHECK(hipMemcpy(ref.data(), ref_d, nq * 16, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(fw.data(), fw_d, nq * 16, hipMemcpyDeviceToHost));
    int shown = 0;
    for (size_t q = 0; q < nq && shown < 6; ++q) {
        if (fw[q * 4] == 0.f && fw[q * 4 + 1] == 0.f) continue;
        // host replay of the round (same fmaf order): want, and what the LOW halves would be had they taken another operand
        float want[4], alt_h[4], alt_w[4];
        const float init[4] = {0.25f, -0.5f, 0.125f, 0.75f};
        for (int c = 0; c < 4; ++c) {
            float a = init[c], b = init[c], d = init[c];
            for (int k = 0; k < K; k += 8)
                for (int j = 0; j < 8; ++j) {
                    const float hj = 0.37f + 0.01f * (float)j, hx = 0.37f + 0.01f * (float)(j ^ 1);
                    const float wc = hw[((size_t)j * nq + q) * 4 + c], wo = hw[((size_t)j * nq + q) * 4 + (c ^ 1)];
                    a = fmaf(wc, hj, a); b = fmaf(wc, hx, b); d = fmaf(wo, hj, d);
                }
            want[c] = a; alt_h[c] = b; alt_w[c] = d;
        }
        printf("      lane-in-wave %zu: got %+.6f %+.6f %+.6f %+.6f | alone %+.6f %+.6f %+.6f %+.6f | host replay %+.6f %+.6f %+.6f %+.6f | if src1 low dword every term %+.6f . %+.6f . | if src0 high dword %+.6f . %+.6f .\n",
               q & 63, fw[q * 4], fw[q * 4 + 1], fw[q * 4 + 2], fw[q * 4 + 3], ref[q * 4], ref[q * 4 + 1], ref[q * 4 + 2], ref[q * 4 + 3], want[0], want[1], want[2], want[3],
               alt_h[0], alt_h[2], alt_w[0], alt_w[2]);
        ++shown;
    }
}

int main(int argc, char **argv) {
    const int WGS = argc > 1 ? atoi(argv[1]) : 64;            // victim workgroups (the real tail: one per image, 8-16 on the chip)
    const size_t nq = (size_t)WGS * 256, nw = (size_t)K * nq * 4;
    std::vector<float> hw(nw), hh(K);
    srand(1);
    for (auto &x : hw) x = (rand() / (float)RAND_MAX - 0.5f) * 0.6f;
    for (auto &x : hh) x = (rand() / (float)RAND_MAX - 0.5f);
    float *w, *h, *sink, *out, *ref;
    unsigned *hist;
    unsigned long long *stamps;
    CHECK(hipMalloc(&w, nw * 4)); CHECK(hipMalloc(&h, K * 4)); CHECK(hipMalloc(&sink, 64));
    CHECK(hipMalloc(&out, nq * 16)); CHECK(hipMalloc(&ref, nq * 16)); CHECK(hipMalloc(&hist, 256 * 4));
    CHECK(hipHostMalloc(&stamps, 4 * 8));
    float *fw;
    CHECK(hipMalloc(&fw, nq * 16));
    CHECK(hipMemcpy(w, hw.data(), nw * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(h, hh.data(), K * 4, hipMemcpyHostToDevice));
    victim_fn victims[10] = {victim_kernel<0>, victim_kernel<1>, victim_kernel<2>, victim_kernel<3>, victim_kernel<4>,
                             victim_kernel<5>, victim_kernel<6>, victim_kernel<7>, victim_kernel<8>, victim_kernel<9>};
    busy_fn busies[6] = {nullptr, busy_kernel<1>, busy_kernel<2>, busy_kernel<3>, busy_kernel<4>, busy_kernel<5>};
    const char *fnames[10] = {"compiled fmaf (v_pk_fma_f32)", "explicit v_fma_f32", "v_pk_mul_f32 + v_pk_add_f32", "registers only, v_pk_fma_f32", "registers only, v_fma_f32",
                              "pk_fma, no modifier", "pk_fma op_sel_hi:[1,0,1]", "pk_fma op_sel:[0,1,0]", "regs only, pk_fma op_sel:[0,1,0]", "regs only, pk_fma no modifier"};
    const char *bnames[6] = {"none", "6 indep 16x16x32 bf16", "6 indep 32x32x16 bf16", "6 indep 32x32x2 f32", "2 dependent 16x16x32 bf16", "v_fma_f32"};
    const int biters[6] = {0, 150000, 80000, 20000, 150000, 600000};      // each ~10-20 ms per resident round
    for (int k = 1; k < 6; ++k) CHECK(hipFuncSetAttribute((const void *)busies[k], hipFuncAttributeMaxDynamicSharedMemorySize, 48 * 1024));
    hipStream_t s1, s2;
    CHECK(hipStreamCreate(&s1)); CHECK(hipStreamCreate(&s2));
    const int f0 = argc > 2 ? atoi(argv[2]) : 0;
    for (int f = f0; f < 10; ++f) {
        const int rounds = (f < 3 || (f >= 5 && f <= 7)) ? 400 : 20000;
        hipLaunchKernelGGL(victims[f], dim3(WGS), dim3(256), 0, s1, w, h, (const float *)nullptr, ref, hist, 1, stamps, (float *)nullptr);     // alone: the reference
        CHECK(hipDeviceSynchronize());
        for (int k = 0; k < (f >= 5 ? 2 : 6); ++k) {
            unsigned total = 0, hh_[256] = {0}, tmp[256];
            double vic_ms = 0, overlap = 0;
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipMemset(hist, 0, 256 * 4));
                CHECK(hipMemset(fw, 0, nq * 16));
                stamps[0] = stamps[1] = stamps[2] = stamps[3] = 0;
                if (k) hipLaunchKernelGGL(busies[k], dim3(768), dim3(256), 48 * 1024, s2, sink, biters[k], 0.3f, stamps + 2);
                hipLaunchKernelGGL(victims[f], dim3(WGS), dim3(256), 0, s1, w, h, (const float *)ref, out, hist, rounds, stamps, fw);
                CHECK(hipDeviceSynchronize());
                CHECK(hipMemcpy(tmp, hist, 256 * 4, hipMemcpyDeviceToHost));
                for (int i = 0; i < 256; ++i) { hh_[i] += tmp[i]; total += tmp[i]; }
                vic_ms += (stamps[1] - stamps[0]) / 1e5;                         // 100 MHz
                if (k) {
                    const double lo = (double)(stamps[0] > stamps[2] ? stamps[0] : stamps[2]), hi = (double)(stamps[1] < stamps[3] ? stamps[1] : stamps[3]);
                    overlap += hi > lo ? (hi - lo) / (double)(stamps[1] - stamps[0]) : 0.0;
                }
            }
            printf("victim %-30s | aggressor %-26s | %8u wrong values of %llu (victim %.2f ms, %.0f %% of it beside the aggressor)\n", fnames[f], bnames[k], total,
                   (unsigned long long)3 * rounds * nq * 4, vic_ms / 3, k ? 100.0 * overlap / 3 : 0.0);
            if (total) {
                unsigned lane16[4] = {0, 0, 0, 0}, comp[4] = {0, 0, 0, 0};
                for (int i = 0; i < 256; ++i) { lane16[(i / 4) / 16] += hh_[i]; comp[i & 3] += hh_[i]; }
                printf("      by lane group 0-15 / 16-31 / 32-47 / 48-63: %u %u %u %u   by component x y z w: %u %u %u %u\n", lane16[0], lane16[1], lane16[2], lane16[3],
                       comp[0], comp[1], comp[2], comp[3]);
                if (f == 8) explain(hw, nq, ref, fw);
            }
            fflush(stdout);
        }
    }
    return 0;
}
