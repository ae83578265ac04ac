// Micro-benchmark (tuning only): how fast can the V workspace of the F(4x4) input transform be WRITTEN, by store pattern?
//   hipcc -O3 --offload-arch=gfx950 tools/micro/store_pattern.hip -o /tmp/store_pattern && /tmp/store_pattern
// V layout: [tile block of 32][Cin/4][9 position groups][4 k][32 tiles] float4.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int TILES = 32, KC = 4, NPG = 9;

__global__ __launch_bounds__(256) void fill_kernel(f32x4 *dst, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 * 9 + threadIdx.x;
#pragma unroll
    for (int g = 0; g < 9; ++g)
        if (i + g * 256 < n) dst[i + g * 256] = f32x4{1.f, 2.f, 3.f, (float)g};
}
// MAP 0: wave = 8 tiles x 8 channels (the shipped kernel); 1: 16 tiles x 4 channels (one k-quad); 2: 32 tiles x 2 channels
template <int MAP, bool NT>
__global__ __launch_bounds__(256) void pat_kernel(f32x4 *v, int nk, int nmb) {
    const int tid = threadIdx.x;
    int sc, st;
    if (MAP == 0) { sc = tid & 7; st = ((tid >> 6) << 3) | ((tid >> 3) & 7); }
    else if (MAP == 1) { const int w = tid >> 6, l = tid & 63; sc = (l >> 4) | ((w & 1) << 2); st = (l & 15) | ((w >> 1) << 4); }
    else { const int w = tid >> 6, l = tid & 63; sc = (l >> 5) | (w << 1); st = l & 31; }
    const int mb = blockIdx.x, c = blockIdx.y * 8 + sc;
    f32x4 *dst = v + (((int64_t)mb * nk + (c >> 2)) * NPG * KC + (c & 3)) * TILES + st;
#pragma unroll
    for (int g = 0; g < NPG; ++g) {
        const f32x4 val = f32x4{(float)g, (float)c, (float)st, 1.f};
        if (NT) __builtin_nontemporal_store(val, dst + g * KC * TILES);
        else dst[g * KC * TILES] = val;
    }
}
// MAP 3: workgroup = one (tile block, k-quad): thread t writes float4 index t + 256 j of the 18 KB run, j = 0..4.5 -> every wave
// instruction writes 1 KB contiguous (what an LDS-staged transform could do)
template <bool NT>
__global__ __launch_bounds__(256) void run_kernel(f32x4 *v, int nk, int nmb) {
    const int tid = threadIdx.x;
    f32x4 *dst = v + ((int64_t)blockIdx.x * nk + blockIdx.y) * (NPG * KC * TILES);
    for (int j = tid; j < NPG * KC * TILES; j += 256) {
        const f32x4 val = f32x4{(float)j, 1.f, 2.f, 3.f};
        if (NT) __builtin_nontemporal_store(val, dst + j);
        else dst[j] = val;
    }
}
template <typename F> float timeit(F f, int n = 20) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) f();
    hipEventRecord(e0);
    for (int i = 0; i < n; ++i) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / n;
}
int main() {
    struct { int B, hw, cin; } shapes[] = {{32, 160, 64}, {32, 80, 128}, {32, 40, 256}, {32, 20, 512}};
    for (auto s : shapes) {
        const int MT = s.B * ((s.hw + 3) / 4) * ((s.hw + 3) / 4), nmb = (MT + TILES - 1) / TILES, nk = s.cin / 4;
        const int64_t n4 = (int64_t)nmb * nk * NPG * KC * TILES;
        f32x4 *v; hipMalloc(&v, n4 * 16);
        const double mb = n4 * 16 / 1e6;
        auto rep = [&](const char *name, float ms) { printf("  %-34s %7.1f us  %5.2f TB/s\n", name, ms * 1e3, mb / ms / 1e6); };
        printf("%d x %dx%d x %d: V = %.0f MB\n", s.B, s.hw, s.hw, s.cin, mb);
        rep("fill (1 KB / wave-instr)", timeit([&] { hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((n4 + 2303) / 2304)), dim3(256), 0, 0, v, n4); }));
        dim3 g(nmb, s.cin / 8);
        rep("map0 8t x 8c  nt", timeit([&] { hipLaunchKernelGGL((pat_kernel<0, true>), g, dim3(256), 0, 0, v, nk, nmb); }));
        rep("map0 8t x 8c  plain", timeit([&] { hipLaunchKernelGGL((pat_kernel<0, false>), g, dim3(256), 0, 0, v, nk, nmb); }));
        rep("map1 16t x 4c nt", timeit([&] { hipLaunchKernelGGL((pat_kernel<1, true>), g, dim3(256), 0, 0, v, nk, nmb); }));
        rep("map1 16t x 4c plain", timeit([&] { hipLaunchKernelGGL((pat_kernel<1, false>), g, dim3(256), 0, 0, v, nk, nmb); }));
        rep("map2 32t x 2c nt", timeit([&] { hipLaunchKernelGGL((pat_kernel<2, true>), g, dim3(256), 0, 0, v, nk, nmb); }));
        rep("map2 32t x 2c plain", timeit([&] { hipLaunchKernelGGL((pat_kernel<2, false>), g, dim3(256), 0, 0, v, nk, nmb); }));
        dim3 g3(nmb, nk);
        rep("run 18 KB / workgroup nt", timeit([&] { hipLaunchKernelGGL((run_kernel<true>), g3, dim3(256), 0, 0, v, nk, nmb); }));
        rep("run 18 KB / workgroup plain", timeit([&] { hipLaunchKernelGGL((run_kernel<false>), g3, dim3(256), 0, 0, v, nk, nmb); }));
        hipFree(v);
    }
    return 0;
}
