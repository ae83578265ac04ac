// Skinny pointwise convolution (1x1, stride 1) for few output channels behind a big map: the MBConv project convs of the
// high-resolution EfficientNet stages (32->16, 16->16 @ H/2, 96->24, 144->24 @ H/4, 144->40, 240->40 @ H/8;
// external/efficientnet/model.py:83-90: x * sigmoid(se) -> _project_conv -> _bn2 -> + inputs) and anything else of
// that shape that reaches mydet_conv2d_igemm_f32.
//
// These layers move 64-600 B per pixel for 16-40 output channels: HBM-bound, and the tiled implicit-GEMM kernel
// spends its time staging the pixel operand through LDS for an N tile that is mostly padding.  Here nothing of x
// touches LDS:
//   * D[channel][pixel] = W . X^T on v_mfma_f32_16x16x4_f32 -- a lane ends up with 4 consecutive channels of one
//     pixel, i.e. 16-byte stores (and 16-byte residual loads);
//   * the pixel operand comes straight from global memory as 16-byte loads: lane (pixel j = lane & 15, kk = lane >> 4)
//     loads x[pixel][16 i + 4 kk .. + 3]; the four MFMAs of chunk i use element e of that vector as their k-slot kk,
//     i.e. MFMA (i, e) sums the logical k = 16 i + 4 kk + e over kk -- a permutation of the K order, matched by the
//     order the weights are laid out in LDS (below); every pixel row is read as whole 64-byte runs;
//   * the weights (at most 256 x 48 floats) are copied to LDS once per workgroup in exactly the per-lane operand
//     order, so an MFMA group's A operand is one conflict-free ds_read_b128;
//   * a wave walks PB blocks of 16 pixels with the loads of the next DEPTH - 1 blocks in flight; a workgroup is four
//     such waves -- short workgroups, many per CU, scheduled by the hardware dispatcher (the persistent variants of the
//     other kernels all lost to that on this chip).
// The squeeze-excite gate multiplies x as it arrives (the reference's `torch.sigmoid(x_squeezed) * x`); the rows of the
// (at most two) images a workgroup touches are staged in LDS.
#include "common.h"

namespace {

struct PwArgs {
    const float *x, *w, *scale, *shift, *res, *gate;
    float *y;
    int64_t ldx, ldr, ldy;
    int M, HW, K, Cout, act;
};

constexpr int PW_WAVES = 4;

template <int KC, int NB, int PB, int DEPTH, bool GATE>
__global__ __launch_bounds__(PW_WAVES * 64) void pw_skinny_kernel(const PwArgs p) {
    constexpr int KP = KC * 16;                    // K rounded up to whole chunks
    constexpr int WG_PIX = PW_WAVES * PB * 16;
    __shared__ __attribute__((aligned(16))) float wl[NB * KC * 64 * 4];
    __shared__ __attribute__((aligned(16))) float gl[GATE ? 2 * KP : 4];
    const int K = p.K;                             // (KC - 1) * 16 < K <= KC * 16, K % 4 == 0
    const int n0 = blockIdx.y * (NB * 16);         // this workgroup's slab of output channels
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int j = lane & 15, kk = lane >> 4;
    const int64_t m_wg = (int64_t)blockIdx.x * WG_PIX;
    const int64_t m_wave = m_wg + (int64_t)wave * PB * 16;

    // pixel rows: unconditional loads at clamped pixel indices (a ragged last workgroup re-reads the last pixel)
    f32x4 a[DEPTH][KC];
    // the last chunk of a K that is not a multiple of 16: lanes whose four k lie beyond K load the row's last four
    // channels instead and are zeroed (the matching weights are zero as well)
    const int k_tail = (KC - 1) * 16 + kk * 4;
    const bool tail_ok = k_tail < K;
    const float *xbase = p.x + kk * 4, *xtail = p.x + (tail_ok ? k_tail : K - 4);
    const int64_t ldx = p.ldx, m_last = (int64_t)p.M - 1;
#define PW_LOAD_BLOCK(pb_, dst_)                                                              \
    {                                                                                         \
        int64_t m_ = m_wave + (pb_) * 16 + j;                                                 \
        m_ = m_ < m_last ? m_ : m_last;                                                       \
        const float *xp_ = xbase + m_ * ldx;                                                  \
        _Pragma("unroll") for (int i_ = 0; i_ < KC - 1; ++i_)(dst_)[i_] = *reinterpret_cast<const f32x4 *>(xp_ + i_ * 16); \
        /* K <= 16: a wave's load covers whole lines that nobody reads again -> streaming load (16->16 @320^2: 61 -> 53 us); */ \
        /* with more chunks a line is shared by two load instructions and the hint costs a refetch (96->24: 41 -> 52 us)   */ \
        if (KC == 1) (dst_)[0] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(xtail + m_ * ldx)); \
        else (dst_)[KC - 1] = *reinterpret_cast<const f32x4 *>(xtail + m_ * ldx);             \
        if (!tail_ok) (dst_)[KC - 1] = f32x4{0.f, 0.f, 0.f, 0.f};                             \
    }
#pragma unroll
    for (int d = 0; d < DEPTH - 1 && d < PB; ++d) PW_LOAD_BLOCK(d, a[d])

    // the prefetched blocks' loads must be issued HERE, not wherever register pressure is lowest (the scheduler
    // would sink them below the previous block's MFMAs)
    __builtin_amdgcn_sched_barrier(0);
    // weights -> LDS in operand order: wl[((nb * KC + i) * 64 + l) * 4 + e] = W[nb * 16 + (l & 15)][16 i + 4 (l >> 4) + e]
    // (all loads of the copy first, then the LDS writes: one L2 round trip, not one per iteration)
    constexpr int WIT = (NB * KC * 64 + PW_WAVES * 64 - 1) / (PW_WAVES * 64);
    f32x4 wt[WIT];
#pragma unroll
    for (int it = 0; it < WIT; ++it) {
        const int idx = min(tid + it * PW_WAVES * 64, NB * KC * 64 - 1);
        const int l = idx & 63, t = idx >> 6, i = t % KC, nb = t / KC;
        const int c = n0 + nb * 16 + (l & 15), k = i * 16 + (l >> 4) * 4;
        wt[it] = *reinterpret_cast<const f32x4 *>(p.w + (int64_t)min(c, p.Cout - 1) * K + min(k, K - 4));
        if (c >= p.Cout || k >= K) wt[it] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int it = 0; it < WIT; ++it) {
        const int idx = tid + it * PW_WAVES * 64;
        if (idx < NB * KC * 64) *reinterpret_cast<f32x4 *>(&wl[idx * 4]) = wt[it];
    }
    const int b0 = (int)(m_wg / p.HW);
    if (GATE) {
        const int nimg = (p.M + p.HW - 1) / p.HW;
        for (int idx = tid; idx < 2 * KP / 4; idx += PW_WAVES * 64) {
            const int which = idx / (KP / 4), q = idx - which * (KP / 4);
            const int b = min(b0 + which, nimg - 1);
            *reinterpret_cast<f32x4 *>(&gl[which * KP + q * 4]) =
                *reinterpret_cast<const f32x4 *>(p.gate + (int64_t)b * K + min(q * 4, K - 4));       // beyond K: x is zero there
        }
    }
    mydet_lds_barrier();              // orders the LDS copies only: the pixel loads stay in flight across it

    const int act = p.act;
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
        if (pb + DEPTH - 1 < PB) PW_LOAD_BLOCK(pb + DEPTH - 1, a[(pb + DEPTH - 1) % DEPTH])
        __builtin_amdgcn_sched_barrier(0);
        f32x4 (&cur)[KC] = a[pb % DEPTH];
        const int64_t m = m_wave + pb * 16 + j;
        if (GATE) {
            // the workgroup's pixels lie in image b0 or b0 + 1 (HW >= WG_PIX is checked on the host)
            const float *gp = gl + (m >= (int64_t)(b0 + 1) * p.HW ? KP : 0) + kk * 4;
#pragma unroll
            for (int i = 0; i < KC; ++i) {
                const f32x4 g = *reinterpret_cast<const f32x4 *>(gp + i * 16);
#pragma unroll
                for (int e = 0; e < 4; ++e) cur[i][e] *= g[e];
            }
        }
        f32x4 acc[NB][2];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb][0] = acc[nb][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < KC; ++i) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const f32x4 wv = *reinterpret_cast<const f32x4 *>(&wl[((nb * KC + i) * 64 + lane) * 4]);
                // two accumulator chains per channel block: consecutive MFMAs never wait on each other's result
                acc[nb][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[0], cur[i][0], acc[nb][0], 0, 0, 0);
                acc[nb][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[1], cur[i][1], acc[nb][1], 0, 0, 0);
                acc[nb][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[2], cur[i][2], acc[nb][0], 0, 0, 0);
                acc[nb][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[3], cur[i][3], acc[nb][1], 0, 0, 0);
            }
        }
        if (m < p.M) {
            float *yp = p.y + m * p.ldy;
            const float *rp = p.res ? p.res + m * p.ldr : nullptr;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int n = n0 + nb * 16 + kk * 4;
                if (n < p.Cout) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[nb][0][e] + acc[nb][1][e];
                    const f32x4 sh = p.shift ? *reinterpret_cast<const f32x4 *>(p.shift + n) : f32x4{0.f, 0.f, 0.f, 0.f};
                    const f32x4 sc = p.scale ? *reinterpret_cast<const f32x4 *>(p.scale + n) : f32x4{1.f, 1.f, 1.f, 1.f};
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = mydet_act(v[e] * sc[e] + sh[e], act);
                    if (rp) {
                        const f32x4 r = *reinterpret_cast<const f32x4 *>(rp + n);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += r[e];
                    }
                    *reinterpret_cast<f32x4 *>(yp + n) = v;
                }
            }
        }
    }
}

template <int KC, int NB, int PB, int DEPTH>
int pw_launch(const PwArgs &a, int slabs, hipStream_t s) {
    constexpr int WG_PIX = PW_WAVES * PB * 16;
    const dim3 grid((unsigned)(((int64_t)a.M + WG_PIX - 1) / WG_PIX), (unsigned)slabs);
    if constexpr (NB <= 3) {
        if (a.gate) {
            hipLaunchKernelGGL((pw_skinny_kernel<KC, NB, PB, DEPTH, true>), grid, dim3(PW_WAVES * 64), 0, s, a);
            return mydet_launch_status();
        }
    }
    if (a.gate) return MYDET_E_UNSUPP;
    hipLaunchKernelGGL((pw_skinny_kernel<KC, NB, PB, DEPTH, false>), grid, dim3(PW_WAVES * 64), 0, s, a);
    return mydet_launch_status();
}

// Output channels are cut into slabs of NB 16-channel blocks (grid.y): a slab's weights (NB * KC KB in operand order)
// must fit LDS several times per CU, and a wave's accumulators its registers.  x is re-read once per slab, from L2.
template <int KC, int PB, int DEPTH>
int pw_launch_nb(const PwArgs &a, hipStream_t s) {
    constexpr int CAP = KC >= 16 ? 2 : KC >= 12 ? 4 : KC >= 10 ? 4 : KC >= 8 ? 5 : 6;
    const int nblocks = (a.Cout + 15) / 16;
    const int slabs = (nblocks + CAP - 1) / CAP;
    const int nb = (nblocks + slabs - 1) / slabs;
    switch (nb) {
        case 1: return pw_launch<KC, 1, PB, DEPTH>(a, slabs, s);
        case 2: return pw_launch<KC, 2, PB, DEPTH>(a, slabs, s);
        case 3: return pw_launch<KC, 3, PB, DEPTH>(a, slabs, s);
        case 4: return pw_launch<KC, 4, PB, DEPTH>(a, slabs, s);
        case 5: if constexpr (CAP >= 5) return pw_launch<KC, 5, PB, DEPTH>(a, slabs, s);
        case 6: if constexpr (CAP >= 6) return pw_launch<KC, 6, PB, DEPTH>(a, slabs, s);
        default: return MYDET_E_UNSUPP;
    }
}

}  // namespace

// Internal to the library (called by mydet_conv2d_igemm_f32, which has validated pointers, alignment and leading
// dimensions): MYDET_E_UNSUPP when the shape is not one this kernel is instantiated for.
int mydet_pw_skinny(const float *x, int64_t ldx, const float *w, const float *scale, const float *shift,
                    const float *residual, int64_t ldr, const float *gate, float *y, int64_t ldy, int B, int HW, int Cin,
                    int Cout, int act, void *stream) {
    const int64_t M = (int64_t)B * HW;
    if ((Cout & 3) || (Cin & 3) || Cin < 4 || Cin > 240 || M >= ((int64_t)1 << 31)) return MYDET_E_UNSUPP;
    // a workgroup covers at most 512 consecutive pixels: two images at most when a map has that many
    if (gate && (HW < 512 || Cout > 48)) return MYDET_E_UNSUPP;
    PwArgs a;
    a.x = x; a.w = w; a.scale = scale; a.shift = shift; a.res = residual; a.gate = gate; a.y = y;
    a.ldx = ldx; a.ldr = ldr; a.ldy = ldy; a.M = (int)M; a.HW = HW; a.K = Cin; a.Cout = Cout; a.act = act;
    hipStream_t s = (hipStream_t)stream;
    switch ((Cin + 15) / 16) {
        case 1: return pw_launch_nb<1, 8, 8>(a, s);
        case 2: return pw_launch_nb<2, 8, 4>(a, s);
        case 3: return pw_launch_nb<3, 4, 2>(a, s);
        case 4: return pw_launch_nb<4, 4, 2>(a, s);
        case 5: return pw_launch_nb<5, 4, 2>(a, s);
        case 6: return pw_launch_nb<6, 4, 2>(a, s);
        case 7: return pw_launch_nb<7, 4, 2>(a, s);
        case 8: return pw_launch_nb<8, 4, 2>(a, s);
        case 9: return pw_launch_nb<9, 4, 2>(a, s);
        case 10: return pw_launch_nb<10, 4, 2>(a, s);
        case 12: return pw_launch_nb<12, 4, 2>(a, s);
        case 15: return pw_launch_nb<15, 4, 2>(a, s);
        default: return MYDET_E_UNSUPP;
    }
}
#undef PW_LOAD_BLOCK
