// 3x3 stride-1 pad-1 convolution by Winograd F(2x2,3x3) on the gfx950 FP32 matrix cores, fully fused:
// input transform while staging, 16 transform-domain GEMMs on v_mfma_f32_32x32x2_f32, output transform +
// BN/activation/residual in the epilogue.  2.25x fewer multiplies than the direct form and no transformed tensor
// ever touches HBM: x is read once per output-channel block, y written once.
//
//   tile  = 2x2 output pixels (4x4 input patch d, origin (2ty-1, 2tx-1), zero outside the image)
//   V     = Bt d B          (adds only; per tile, per input channel)        Bt = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
//   U     = G g Gt          (once per weight, mydet_wino_weights_f32)        G  = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]
//   M_p   = sum_c U_p[n][c] * V_p[c][tile]      for the 16 positions p = 4i+j -- the MFMA part
//   Y     = At M A          (per tile, per output channel)                  At = [1 1 1 0; 0 1 -1 -1]
//   y     = act(Y*scale[n] + shift[n]) + residual
//
// A wave owns 16 output channels x 32 tiles for ALL 16 positions as 16x16 v_mfma_f32_16x16x4_f32 blocks
// (16 positions x 2 tile blocks x 4 = 128 accumulator registers, two waves per SIMD); a workgroup of NW waves owns
// 64 channels x 8*NW tiles (see the two shapes at the kernel).  K = Cin is walked 8 channels at a time through LDS
// slabs laid out in MFMA-fragment order: [position pair][k quarter][row] float4 = {pos 2p, 2p+1 at k; pos 2p, 2p+1
// at k+1}, so one ds_read_b128 feeds two positions; the tile column of V is XOR-swizzled with the k quarter so that
// the fragment reads and the staging stores are both bank-conflict free without padding.
// Every thread stages one channel of one tile's patch (16 dword buffer loads -- a wave covers 8 tiles x the 8
// channels of the slab, one 32-byte run per pixel --, transform in registers, 8 LDS stores) and its share of the
// weights (straight through, global and LDS order coincide): the next slab's loads are issued before the 64 MFMAs
// of the current one.  Out-of-image patch pixels use voffset 0xFFFFFFFF (hardware range check returns 0); the K
// advance rides in the scalar offset.  Fragment reads of position pair p+1 are issued under pair p's MFMAs.
// Output channels are MFMA rows, so a lane ends up with 4 consecutive channels of a pixel: 16-byte stores, and the
// residual loads of a lane are all in flight before its first store.
//
// Replaces the same ATen chain as conv_igemm.hip for the 3x3 layers of ConvBnLeaky / DarkBlock
// (models/modules.py:69-73,94-95) and the dense 3x3 convs of models/backbones.py:183-200, models/rpns.py:155-158.
// float32 throughout; versus the direct form only the association order of the sums differs.
#include <cstdlib>

#include "common.h"

namespace {

constexpr unsigned OOB = 0xFFFFFFFFu;
constexpr int CH = 64;                                 // output channels per workgroup
constexpr int U_BYTES = 8 * 4 * CH * 16;              // one 8-channel slab of U (32 KB)

struct WinoArgs {
    const float *x, *u, *scale, *shift, *res;
    float *y;
    int64_t ldx, ldr, ldy;
    int B, H, W, Cin, Cout, CoutP;
    int TH, TW, MT, ntn, nblk;
    // stream-K schedule (SK kernels): `nwg` persistent workgroups split the items*nk slab iterations evenly; a piece
    // that does not cover its item's whole K writes its output-domain partial sums to ws (slot 2w: piece that starts
    // inside an item, 2w+1: piece that starts an item but does not finish it) for conv_wino_fixup_kernel
    int nwg, nk;
    int skq, skr;             // slab iterations of the stream-K tail = skq * nwg + skr (host side: no 64-bit division on the device)
    float *ws;
    size_t ws_bytes;
};

// first slab iteration of persistent workgroup w: floor(w * total / nwg)
// = w * q + floor(w * r / nwg) with total = q * nwg + r: 32-bit arithmetic (the tail has < nwg items of <= a few hundred slabs;
// the 64-bit divisions this replaces cost a small-grid launch -- batch 1 -- a microsecond of pure latency)
__device__ __forceinline__ int sk_begin(int w, const WinoArgs &p) { return w * p.skq + (int)((unsigned)(w * p.skr) / (unsigned)p.nwg); }

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float *base, int64_t bytes) {
    const uint64_t a = (uint64_t)base;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    const int64_t capped = bytes > 0x7FFFFFF0ll ? 0x7FFFFFF0ll : bytes;
    const int n = __builtin_amdgcn_readfirstlane((int)capped);
    return __builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)hi << 32) | lo), 0, n, 0x00020000);
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// y = act(Y*scale + shift) + residual for one item: lane = tile m0 + 32*wt + 16*blk + fr, its four components =
// channels n0 + 16*wc + 4*fq + (0..3); out[blk][2a+c] = output pixel (2ty+a, 2tx+c).  All residual loads of a
// lane are in flight before its first store.
template <int ACT, bool RES>
__device__ __forceinline__ void wino_epilogue(const WinoArgs &p, const f32x4 (&out)[2][4], int m0, int n0, int b0,
                                              int wc, int wt, int fr, int fq) {
    const int tpi = p.TH * p.TW;
    const int n = n0 + wc * 16 + fq * 4;
    const bool nok = n < p.Cout;                       // Cout % 4 == 0: the four channels stand or fall together
    const int nc = nok ? n : 0;
    const f32x4 scl = p.scale ? *reinterpret_cast<const f32x4 *>(p.scale + nc) : f32x4{1.f, 1.f, 1.f, 1.f};
    const f32x4 sft = p.shift ? *reinterpret_cast<const f32x4 *>(p.shift + nc) : f32x4{0.f, 0.f, 0.f, 0.f};
    const int64_t oimg = (int64_t)p.H * p.W;
    const __amdgpu_buffer_rsrc_t yr = make_rsrc(p.y + b0 * oimg * p.ldy, (p.B - b0) * oimg * p.ldy * 4);
    const __amdgpu_buffer_rsrc_t rr =
        make_rsrc(RES ? p.res + b0 * oimg * p.ldr : p.y, (p.B - b0) * oimg * (RES ? p.ldr : p.ldy) * 4);
    unsigned yo[2][4];                                 // byte offsets of the 2 x 4 output pixels (OOB when masked)
    f32x4 rv[2][4];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
        const int mt = m0 + wt * 32 + blk * 16 + fr;
        const bool tok = mt < p.MT && nok;
        const int mm = mt < p.MT ? mt : p.MT - 1;
        const int b = mm / tpi, r = mm - b * tpi, ty = r / p.TW, tx = r - ty * p.TW;
        const int oy = 2 * ty, ox = 2 * tx;
        const int64_t pix = ((int64_t)(b - b0) * p.H + oy) * p.W + ox;
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const int a = o >> 1, c = o & 1;
            const bool ok = tok && oy + a < p.H && ox + c < p.W;
            const int64_t px = pix + (int64_t)a * p.W + c;
            yo[blk][o] = ok ? (unsigned)((px * p.ldy + n) * 4) : OOB;
            if (RES)
                rv[blk][o] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                           rr, ok ? (unsigned)((px * p.ldr + n) * 4) : OOB, 0, 0));
        }
    }
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            f32x4 v = out[blk][o] * scl + sft;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (ACT == MYDET_ACT_LEAKY) v[e] = v[e] > 0.0f ? v[e] : v[e] * 0.1f;
                if (ACT == MYDET_ACT_SWISH) v[e] = v[e] * mydet_sigmoid_fast(v[e]);
            }
            if (RES) v += rv[blk][o];
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yr, yo[blk][o], 0, 0);
        }
}

// NW = 4: 32 tiles per workgroup, one LDS slab, two barriers per slab, two workgroups per CU -- short-K layers,
//         where a workgroup's prologue and epilogue must hide under its neighbour's MFMAs.
// NW = 8: 64 tiles per workgroup (waves 0-3 / 4-7 take 32 tiles each and share the weights), two LDS slabs, ONE
//         barrier per slab: the next slab's transform and LDS stores are woven into the second half of the
//         current slab's MFMAs, so in steady state the matrix pipe only idles across that barrier.
//
// SK = false: one workgroup per item (item = 8*NW tiles x 64 channels), all of K.
// SK = true : data-parallel rounds + stream-K tail.  The grid is one resident round of G persistent workgroups;
//         each takes floor(items / G) whole items in grid order, and the remaining items (the partial last round
//         of the plain grid) are cut along K: workgroup w owns the slab iterations [w*T/G, (w+1)*T/G) of their
//         item-major, K-minor sequence, so every workgroup does the same amount of matrix work.
//         A piece that covers only part of an item's K leaves its partial
//         outputs in the workspace; conv_wino_fixup_kernel sums the pieces of such items in K order and applies the
//         epilogue (deterministic, no atomics, no waiting inside the kernel).
template <int ACT, bool RES, int NW, bool SK>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void conv_wino_kernel(const WinoArgs p) {
    constexpr int TILES = 8 * NW;                      // 2x2-output tiles per workgroup
    constexpr int V_KS = TILES * 16, V_PS = 4 * V_KS;  // V: bytes between k quarters / position pairs
    constexpr int SLAB = U_BYTES + 8 * V_PS;           // 48 KB (NW = 4) / 64 KB (NW = 8)
    constexpr int NU = 2048 / (64 * NW);               // float4 of weights each thread stages per slab
    constexpr bool DB = NW == 8;
#ifndef WINO_LDS_DIRECT
#define WINO_LDS_DIRECT 0   // measured 1-2 % slower than staging the weights through registers
#endif
    constexpr bool DMA = DB && WINO_LDS_DIRECT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int lid = mydet_xcd_remap(blockIdx.x, SK ? p.nwg : p.nblk);
    const int tpi = p.TH * p.TW;                       // tiles per image
    const int nk = p.Cin >> 3;
    // SK: `rounds` whole items per workgroup in grid order (item = round * G + lid: the workgroups of an XCD then work
    // on neighbouring items at the same time, as in the plain grid, and share their patches / weights in its L2), then
    // an equal share [it, it_end) of the slab iterations of the remaining `nblk - rounds * G` items
    const int rounds = SK ? p.nblk / p.nwg : 1;
    const int tail0 = SK ? rounds * p.nwg : 0;                            // first item of the stream-K tail
    int it = SK ? sk_begin(lid, p) : 0;
    const int it_end = SK ? sk_begin(lid + 1, p) : 0;
    int round = 0;
    const int wc = wave & 3, wt = wave >> 2;           // compute role: channels 16*wc.., tiles 32*wt..32*wt+31
    const int fr = lane & 15, fq = lane >> 4;
    const int slot = wave * 8 + (lane >> 3), kc = lane & 7;   // staging role: tile, channel of the slab
    // Per-piece state.  `setup` derives it for the piece starting at slab iteration `at`; the stream-K loop calls it for
    // the NEXT piece before the epilogue of the current one, so that piece's first slab is already in flight.
    int item, k_lo, k_hi, m0, n0, b0;
    __amdgpu_buffer_rsrc_t xr;
    const __amdgpu_buffer_rsrc_t ur = make_rsrc(p.u, (int64_t)p.Cin * 16 * p.CoutP * 4);
    unsigned off[16], uoff;
    auto setup = [&]() -> bool {                       // next piece of this workgroup, false when there is none
        if (round < rounds) {
            item = SK ? round * p.nwg + lid : lid;
            k_lo = 0; k_hi = nk;
            ++round;
        } else if (SK && it < it_end) {
            const int ti = it / nk;
            item = tail0 + ti;
            k_lo = it - ti * nk;
            k_hi = nk - k_lo < it_end - it ? nk : k_lo + (it_end - it);
            it += k_hi - k_lo;
        } else {
            return false;
        }
        m0 = (item / p.ntn) * TILES;
        n0 = (item % p.ntn) * CH;
        b0 = m0 / tpi;
        // staging: every thread brings one channel of one tile's patch (16 dwords) and NU float4 of weights; a wave
        // covers 8 tiles x the 8 channels of the slab: one 32-byte run per patch pixel and load instruction
        const int64_t img = (int64_t)p.H * p.W * p.ldx;
        xr = make_rsrc(p.x + b0 * img, (p.B - b0) * img * 4);
        const int mt = m0 + slot;
        const int mm = mt < p.MT ? mt : p.MT - 1;
        const int b = mm / tpi, r = mm - b * tpi, ty = r / p.TW, tx = r - ty * p.TW;
        const int iy0 = 2 * ty - 1, ix0 = 2 * tx - 1;
        const int base = (int)(((((int64_t)(b - b0) * p.H + iy0) * p.W + ix0) * p.ldx + kc) * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool ok = mt < p.MT && (unsigned)(iy0 + i) < (unsigned)p.H && (unsigned)(ix0 + j) < (unsigned)p.W;
                off[i * 4 + j] = ok ? (unsigned)(base + (int)(((int64_t)i * p.W + j) * p.ldx * 4)) : OOB;
            }
        uoff = (unsigned)(((tid >> 6) * p.CoutP + n0 + (tid & 63)) * 16);   // U slab float4 q*64*NW + tid
        return true;
    };
    // V slab element (pair, quarter kq = kc/2, tile): this thread owns components {2s, 2s+1}, s = kc & 1.  The tile
    // column is XOR-swizzled with 2*kq: the ds_write_b64 lane groups (2 tiles x 8 channels) and the ds_read_b128
    // lane groups (8 + 8 tiles of two adjacent quarters) then both touch every bank once.
    const unsigned wr_v = U_BYTES + (unsigned)((kc >> 1) * V_KS + ((slot ^ (kc & 6)) * 16) + (kc & 1) * 8);
    // (U slab float4 q*64*NW + tid: global and LDS order coincide)
    const unsigned ustep = __builtin_amdgcn_readfirstlane(16u * NW * (unsigned)p.CoutP);   // bytes between q and q+1
    const unsigned wr_u = (unsigned)tid * 16u;

    // ---- compute role: wave owns channels 16*wc.. and tiles 32*wt..32*wt+31 (two 16x16 blocks)
    const unsigned rd_u = (unsigned)(fq * CH + wc * 16 + fr) * 16u;                  // + position pair * 4*CH*16
    const unsigned rd_v = U_BYTES + (unsigned)(fq * V_KS + ((wt * 32 + fr) ^ (2 * fq)) * 16);   // + pair * V_PS, + 256: block 1

    f32x4 acc[16][2];

    float gv[16];
    f32x4 gu[DMA ? 1 : NU];
    // DB: the weights go global -> LDS directly (buffer_load_dwordx4 ... lds: lane i of a wave lands at base + 16*i,
    // and the slab order in memory IS the LDS order), into the slab that was released at the previous barrier;
    // otherwise through registers, stored after the barrier that frees the only slab.
    auto load_slab = [&](int kt, char *next) {
        const unsigned sv = (unsigned)kt * 32u, su = (unsigned)kt * (512u * (unsigned)p.CoutP);
#pragma unroll
        for (int q = 0; q < 16; ++q)
            gv[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, off[q], sv, 0));
        if constexpr (DMA) {
            char *dst = next + __builtin_amdgcn_readfirstlane(wave) * 1024;
#pragma unroll
            for (int q = 0; q < NU; ++q)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ur, (__attribute__((address_space(3))) void *)(dst + q * (1024 * NW)),
                                                         16, uoff, su + q * ustep, 0, 0);
        } else {
#pragma unroll
            for (int q = 0; q < NU; ++q)
                gu[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ur, uoff, su + q * ustep, 0));
        }
    };
    auto store_slab = [&](char *slab) {
        float t[16];                                   // V = Bt d B (rows, then columns) of this thread's channel
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            t[j] = gv[j] - gv[8 + j]; t[4 + j] = gv[4 + j] + gv[8 + j];
            t[8 + j] = gv[8 + j] - gv[4 + j]; t[12 + j] = gv[4 + j] - gv[12 + j];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float t0 = t[4 * i], t1 = t[4 * i + 1], t2 = t[4 * i + 2], t3 = t[4 * i + 3];
            *reinterpret_cast<f32x2 *>(slab + wr_v + (2 * i) * V_PS) = f32x2{t0 - t2, t1 + t2};
            *reinterpret_cast<f32x2 *>(slab + wr_v + (2 * i + 1) * V_PS) = f32x2{t2 - t1, t1 - t3};
        }
        if constexpr (!DMA) {
#pragma unroll
            for (int q = 0; q < NU; ++q) *reinterpret_cast<f32x4 *>(slab + wr_u + q * (1024 * NW)) = gu[q];
        }
    };
    // MFMAs of position pairs [p0, p1) of one slab; the fragments of pair pp+1 are read under pair pp's MFMAs
    f32x4 fu[2], fv[2][2];
    auto read_frag = [&](const char *slab, int pp, int st) {
        fu[st] = *reinterpret_cast<const f32x4 *>(slab + rd_u + pp * (4 * CH * 16));
        fv[st][0] = *reinterpret_cast<const f32x4 *>(slab + rd_v + pp * V_PS);
        fv[st][1] = *reinterpret_cast<const f32x4 *>(slab + rd_v + pp * V_PS + 256);
    };
    auto mma_pair = [&](int pp, int st) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int blk = 0; blk < 2; ++blk)
                    acc[2 * pp + e][blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                        fu[st][2 * s + e], fv[st][blk][2 * s + e], acc[2 * pp + e][blk], 0, 0, 0);
    };

    if (!setup()) return;                              // (stream-K grids are never larger than the item count)
    load_slab(k_lo, smem + (DB ? (k_lo & 1) * SLAB : 0));
  while (true) {
#pragma unroll
    for (int q = 0; q < 16; ++q)
#pragma unroll
        for (int e = 0; e < 2; ++e) acc[q][e] = f32x4{0.f, 0.f, 0.f, 0.f};
    store_slab(smem + (DB ? (k_lo & 1) * SLAB : 0));
    if (DMA) __builtin_amdgcn_s_waitcnt(0x0F70);       // vmcnt(0): the LDS-direct weight loads have landed
    __syncthreads();
    for (int kt = k_lo; kt < k_hi; ++kt) {
        const char *cur = smem + (DB ? (kt & 1) * SLAB : 0);
        load_slab(kt + 1 < k_hi ? kt + 1 : kt, smem + (DB ? ((kt + 1) & 1) * SLAB : 0));   // past the end: re-load, never consumed
        __builtin_amdgcn_sched_barrier(0);
        read_frag(cur, 0, 0);
#pragma unroll
        for (int pp = 0; pp < 8; ++pp) {
            if (pp + 1 < 8) read_frag(cur, pp + 1, (pp + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);         // keeps the reads above this pair's MFMAs
            if (DB && pp == 4) {                       // the other slab was last read before the previous barrier
#pragma unroll
                for (int q = 0; q < 16; ++q) asm volatile("" : "+v"(gv[q]));  // pins the transform to this point
                store_slab(smem + ((kt + 1) & 1) * SLAB);
            }
            mma_pair(pp, pp & 1);
        }
        if (DMA) __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): next slab's LDS-direct weight loads have landed
        __syncthreads();                               // every wave is done with the slab
        if (!DB) {
            if (kt + 1 < k_hi) {
#pragma unroll
                for (int q = 0; q < 16; ++q) asm volatile("" : "+v"(gv[q]));  // pins the transform below the MFMAs
                store_slab(smem);
            }
            __syncthreads();
        }
    }

    // ---- output transform: lane = tile 32*wt + fr (+16 for block 1), components = channels n0+16wc+4fq+(0..3)
    f32x4 out[2][4];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
        f32x4 s0[4], s1[4];                            // At M: rows (m0+m1+m2), (m1-m2-m3) per column j
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s0[j] = acc[j][blk] + acc[4 + j][blk] + acc[8 + j][blk];
            s1[j] = acc[4 + j][blk] - acc[8 + j][blk] - acc[12 + j][blk];
        }
        out[blk][0] = s0[0] + s0[1] + s0[2];
        out[blk][1] = s0[1] - s0[2] - s0[3];
        out[blk][2] = s1[0] + s1[1] + s1[2];
        out[blk][3] = s1[1] - s1[2] - s1[3];
    }
    const bool whole = !SK || (k_lo == 0 && k_hi == nk), starts = k_lo == 0;
    const int cm0 = m0, cn0 = n0, cb0 = b0;
    const bool more = SK && setup();
    if (more)                                          // next piece: its first slab flies under this epilogue
        load_slab(k_lo, smem + (DB ? (k_lo & 1) * SLAB : 0));
    if (whole) {
        wino_epilogue<ACT, RES>(p, out, cm0, cn0, cb0, wc, wt, fr, fq);
    } else {                                           // partial K: [float4 j][thread], summed by the fixup launch
        f32x4 *dst = reinterpret_cast<f32x4 *>(p.ws) + (int64_t)(2 * lid + (starts ? 1 : 0)) * 8 * (64 * NW) + tid;
#pragma unroll
        for (int j = 0; j < 8; ++j) dst[j * (64 * NW)] = out[j >> 2][j & 3];
    }
    if (!more) break;
  }
}

// Items whose K range was cut by the stream-K schedule: sum the pieces in K order, then the usual epilogue.
// blockIdx.x = item of the stream-K tail (an item that one workgroup summed whole returns at once).
template <int ACT, bool RES, int NW>
__global__ __launch_bounds__(64 * NW) void conv_wino_fixup_kernel(const WinoArgs p) {
    constexpr int TILES = 8 * NW, NT = 64 * NW;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int nk = p.nk;
    const int tail0 = (p.nblk / p.nwg) * p.nwg;        // the stream-K tail starts after the whole rounds
    const int total = p.skq * p.nwg + p.skr;
    const int item = tail0 + (int)blockIdx.x;
    const int first = (int)blockIdx.x * nk;            // the item's slab iterations: [first, first + nk)
    // the first share boundary inside the item: smallest w with floor(w * total / nwg) > first
    const int w = (int)(((int64_t)(first + 1) * p.nwg + total - 1) / total);
    if (w >= p.nwg || sk_begin(w, p) >= first + nk) return;    // not cut
    // blockIdx.y = which of the thread's eight output float4 (2 tile blocks x 2 x 2 pixels) this workgroup sums: a cut item
    // is 17 pieces of 64 KB at batch 1, and ONE workgroup pulling them through one CU took 17 us (the CU's intake, not
    // the loads' latency); eight workgroups per item take 1/8 each
    const int j = blockIdx.y;
    // Everything this workgroup needs is requested before anything is waited for: the epilogue's operands (scale, shift, the
    // residual pixel) and up to 17 pieces (the first one + 16: a batch-1 layer cuts every item 16-17 ways) -- one round trip
    // instead of four dependent ones (first piece, two batches of eight, epilogue operands): 7 -> 5 us per launch at batch 1.
    const int m0 = (item / p.ntn) * TILES, n0 = (item % p.ntn) * CH;
    const int b0 = m0 / (p.TH * p.TW);
    const int wc = wave & 3, wt = wave >> 2, fr = lane & 15, fq = lane >> 4;
    const int n = n0 + wc * 16 + fq * 4;
    const bool nok = n < p.Cout;
    const int nc = nok ? n : 0;
    const f32x4 scl = p.scale ? *reinterpret_cast<const f32x4 *>(p.scale + nc) : f32x4{1.f, 1.f, 1.f, 1.f};
    const f32x4 sft = p.shift ? *reinterpret_cast<const f32x4 *>(p.shift + nc) : f32x4{0.f, 0.f, 0.f, 0.f};
    const int64_t oimg = (int64_t)p.H * p.W;
    const __amdgpu_buffer_rsrc_t yr = make_rsrc(p.y + b0 * oimg * p.ldy, (p.B - b0) * oimg * p.ldy * 4);
    const __amdgpu_buffer_rsrc_t rr =
        make_rsrc(RES ? p.res + b0 * oimg * p.ldr : p.y, (p.B - b0) * oimg * (RES ? p.ldr : p.ldy) * 4);
    unsigned yo;
    f32x4 rv = {0.f, 0.f, 0.f, 0.f};
    {
        const int blk = j >> 2, o = j & 3, tpi = p.TH * p.TW;
        const int mt = m0 + wt * 32 + blk * 16 + fr;
        const int mm = mt < p.MT ? mt : p.MT - 1;
        const int b = mm / tpi, r = mm - b * tpi, ty = r / p.TW, tx = r - ty * p.TW;
        const int oy = 2 * ty + (o >> 1), ox = 2 * tx + (o & 1);
        const bool ok = mt < p.MT && nok && oy < p.H && ox < p.W;
        const int64_t px = ((int64_t)(b - b0) * p.H + oy) * p.W + ox;
        yo = ok ? (unsigned)((px * p.ldy + n) * 4) : OOB;
        if (RES)
            rv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rr, ok ? (unsigned)((px * p.ldr + n) * 4) : OOB, 0, 0));
    }
    const f32x4 *ws = reinterpret_cast<const f32x4 *>(p.ws) + tid + (int64_t)j * NT;
    // first workgroup whose share begins at or beyond the item's end: floor(v * total / nwg) >= X  <=>  v >= X * nwg / total
    // (one 64-bit division instead of one per piece)
    const int64_t xe = (int64_t)(first + nk) * p.nwg;
    int last = (int)((xe + total - 1) / total);
    if (last > p.nwg) last = p.nwg;
    f32x4 sum = ws[(int64_t)(2 * (w - 1) + 1) * 8 * NT];                                               // starts the item
    f32x4 t[16];
#pragma unroll
    for (int u = 0; u < 16; ++u)
        if (w + u < last) t[u] = ws[(int64_t)(2 * (w + u)) * 8 * NT];                                  // uniform
#pragma unroll
    for (int u = 0; u < 16; ++u)                       // added in K order
        if (w + u < last) sum += t[u];
    for (int v0 = w + 16; v0 < last; v0 += 8) {         // more than 17 pieces: eight at a time
        f32x4 t8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (v0 + u < last) t8[u] = ws[(int64_t)(2 * (v0 + u)) * 8 * NT];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (v0 + u < last) sum += t8[u];
    }
    f32x4 v = sum * scl + sft;                         // as wino_epilogue
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (ACT == MYDET_ACT_LEAKY) v[e] = v[e] > 0.0f ? v[e] : v[e] * 0.1f;
        if (ACT == MYDET_ACT_SWISH) v[e] = v[e] * mydet_sigmoid_fast(v[e]);
    }
    if (RES) v += rv;
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yr, yo, 0, 0);
}

// U = G g Gt in float64, rounded once; layout [Cin/8][8 position pairs][4 k quarters][CoutP][4] with the float4 =
// {pos 2p, 2p+1 at k; pos 2p, 2p+1 at k+1}, k = 2*quarter (CoutP = Cout rounded up to 64, zero rows): exactly the LDS slab,
// so a workgroup's share of a slab is 32 contiguous runs of 64 float4.
__global__ void wino_weights_kernel(const float *w, int Cout, int Cin, int CoutP, float *u) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)CoutP * Cin) return;
    const int n = (int)(i / Cin), c = (int)(i - (int64_t)n * Cin);
    double g[3][3], t[4][3];
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) g[a][b] = n < Cout ? (double)w[(((int64_t)n * 3 + a) * 3 + b) * Cin + c] : 0.0;
    for (int b = 0; b < 3; ++b) {
        t[0][b] = g[0][b];
        t[1][b] = 0.5 * (g[0][b] + g[1][b] + g[2][b]);
        t[2][b] = 0.5 * (g[0][b] - g[1][b] + g[2][b]);
        t[3][b] = g[2][b];
    }
    const int kc = c >> 3, kq = (c >> 1) & 3, ks = c & 1;
    for (int a = 0; a < 4; ++a) {
        const double r[4] = {t[a][0], 0.5 * (t[a][0] + t[a][1] + t[a][2]), 0.5 * (t[a][0] - t[a][1] + t[a][2]), t[a][2]};
        for (int b = 0; b < 4; ++b) {
            const int q = a * 4 + b;
            u[((((int64_t)kc * 8 + (q >> 1)) * 4 + kq) * CoutP + n) * 4 + ks * 2 + (q & 1)] = (float)r[b];
        }
    }
}

int forced_sk() {
    static int v = -2;
    if (v == -2) {
        const char *e = getenv("MYDET_WINO_SK");     // tuning only: 0 = never use the stream-K schedule
        v = e ? atoi(e) : -1;
    }
    return v;
}

template <int ACT, bool RES, int NW>
int launch_nw(WinoArgs a, hipStream_t stream) {
    constexpr int TILES = 8 * NW, NT = 64 * NW;
    constexpr int LDS = (U_BYTES + 8 * 4 * TILES * 16) * (NW == 8 ? 2 : 1);
    static unsigned long long attr_set = 0, attr_set_sk = 0;  // > 64 KiB of dynamic LDS needs the opt-in once per device
    if (const int e = mydet_lds_opt_in(attr_set, &conv_wino_kernel<ACT, RES, NW, false>, LDS)) return e;
    if (const int e = mydet_lds_opt_in(attr_set_sk, &conv_wino_kernel<ACT, RES, NW, true>, LDS)) return e;
    a.nblk = (int)(((int64_t)a.MT + TILES - 1) / TILES) * a.ntn;
    a.nk = a.Cin >> 3;
    // stream-K when the grid is at least two resident rounds (below that the plain grid is already one round or
    // its prologue/epilogue overlap is what matters) and the caller gave room for the partial tiles
    const int resident = mydet_cu_count() * (NW == 8 ? 1 : 2);
    const size_t need = (size_t)2 * resident * 8 * NT * sizeof(f32x4);
    // (measured: pays on the 64-tile shape, whose single workgroup per CU exposes the partial last round; not on the
    // 32-tile shape, MYDET_WINO_SK=1 forces it there)
    // ... or a grid that fills less than half of the chip (batch-1 / small-map layers): then ALL items are cut along K.
    // Every persistent workgroup must own at least one slab iteration of the cut part (the fixup sums the pieces of
    // ALL workgroups between two item boundaries), so the grid shrinks to the iteration count when that is smaller.
    const bool big = a.nblk >= 2 * resident, small = a.nblk * 2 <= resident && a.nk >= 8;
    int nwg = resident;
    if (small && (int64_t)a.nblk * a.nk < resident) nwg = a.nblk * a.nk;
    const int64_t tail_total = (int64_t)(a.nblk - (a.nblk / nwg) * nwg) * a.nk;
    const bool covered = tail_total == 0 || tail_total >= nwg;
    if (forced_sk() != 0 && (NW == 8 || forced_sk() == 1) && a.ws && need <= a.ws_bytes && (big || small) && covered) {
        a.nwg = nwg;
        a.skq = (int)(tail_total / nwg); a.skr = (int)(tail_total % nwg);
        hipLaunchKernelGGL((conv_wino_kernel<ACT, RES, NW, true>), dim3(nwg), dim3(NT), LDS, stream, a);
        int rc = mydet_launch_status();
        if (rc || nwg < 2) return rc;
        const int tail_items = a.nblk - (a.nblk / nwg) * nwg;
        if (tail_items == 0) return rc;
        hipLaunchKernelGGL((conv_wino_fixup_kernel<ACT, RES, NW>), dim3(tail_items, 8), dim3(NT), 0, stream, a);
        return mydet_launch_status();
    }
    a.nwg = 0; a.skq = 0; a.skr = 0;
    hipLaunchKernelGGL((conv_wino_kernel<ACT, RES, NW, false>), dim3(a.nblk), dim3(NT), LDS, stream, a);
    return mydet_launch_status();
}

int forced_nw() {
    static int v = -2;
    if (v == -2) {
        const char *e = getenv("MYDET_WINO_NW");     // tuning only: 4 or 8
        v = e ? atoi(e) : -1;
    }
    return v;
}

template <int ACT, bool RES>
int launch_inst(const WinoArgs &a, hipStream_t stream) {
    const int nw = forced_nw() > 0 ? forced_nw() : (a.Cin >= 128 ? 8 : 4);
    return nw == 8 ? launch_nw<ACT, RES, 8>(a, stream) : launch_nw<ACT, RES, 4>(a, stream);
}

}  // namespace

extern "C" int64_t mydet_wino_weights_floats(int Cout, int Cin) {
    if (Cout <= 0 || Cin <= 0 || (Cin & 7)) return 0;
    return (int64_t)16 * Cin * ((Cout + 63) / 64 * 64);
}

extern "C" int mydet_wino_weights_f32(const float *w, int Cout, int Cin, float *u, void *stream) {
    if (!w || !u || Cout <= 0 || Cin <= 0) return MYDET_E_BADARG;
    if (Cin & 7) return MYDET_E_UNSUPP;
    const int CoutP = (Cout + 63) / 64 * 64;
    const int64_t n = (int64_t)CoutP * Cin;
    hipLaunchKernelGGL(wino_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w,
                       Cout, Cin, CoutP, u);
    return mydet_launch_status();
}

extern "C" int mydet_conv2d_wino_f32(const float *x, int64_t ldx, const float *u, const float *scale,
                                     const float *shift, const float *residual, int64_t ldr, void *workspace,
                                     int64_t workspace_bytes, float *y, int64_t ldy, int B, int H, int W, int Cin,
                                     int Cout, int act, void *stream) {
    if (!x || !u || !y || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || act < 0 || act > 2)
        return MYDET_E_BADARG;
    if ((ldx & 3) || ldx < Cin || ldy < Cout || (residual && ldr < Cout)) return MYDET_E_BADARG;
    if (((uintptr_t)x & 15) || ((uintptr_t)u & 15) || ((uintptr_t)y & 15) || (residual && ((uintptr_t)residual & 15)) ||
        (scale && ((uintptr_t)scale & 15)) || (shift && ((uintptr_t)shift & 15)))
        return MYDET_E_BADARG;
    if ((Cin & 7) || (Cout & 3) || (ldy & 3) || (residual && (ldr & 3))) return MYDET_E_UNSUPP;
    WinoArgs a;
    a.x = x; a.u = u; a.scale = scale; a.shift = shift; a.res = residual; a.y = y;
    a.ldx = ldx; a.ldr = residual ? ldr : ldy; a.ldy = ldy;
    a.ws = ((uintptr_t)workspace & 15) ? nullptr : (float *)workspace;
    a.ws_bytes = workspace_bytes > 0 ? (size_t)workspace_bytes : 0;
    a.nwg = 0; a.nk = 0; a.nblk = 0;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.CoutP = (Cout + 63) / 64 * 64;
    a.TH = (H + 1) / 2; a.TW = (W + 1) / 2;
    const int64_t MT = (int64_t)B * a.TH * a.TW;
    if (MT > (int64_t)1 << 30) return MYDET_E_UNSUPP;
    // 32-bit byte offsets inside a workgroup's window: the images its (at most 64) tiles touch
    const int64_t span = 64 / ((int64_t)a.TH * a.TW) + 2;
    const int64_t ldmax = ldx > ldy ? (ldx > a.ldr ? ldx : a.ldr) : (ldy > a.ldr ? ldy : a.ldr);
    if ((int64_t)H * W * ldmax * 4 * span >= 0x7FFFFFF0ll || (int64_t)16 * Cin * a.CoutP * 4 >= 0x7FFFFFF0ll)
        return MYDET_E_UNSUPP;
    a.MT = (int)MT;
    a.ntn = a.CoutP / CH;
    hipStream_t s = (hipStream_t)stream;
    const bool res = residual != nullptr;
    switch (act) {
        case MYDET_ACT_LEAKY: return res ? launch_inst<MYDET_ACT_LEAKY, true>(a, s) : launch_inst<MYDET_ACT_LEAKY, false>(a, s);
        case MYDET_ACT_SWISH: return res ? launch_inst<MYDET_ACT_SWISH, true>(a, s) : launch_inst<MYDET_ACT_SWISH, false>(a, s);
        default: return res ? launch_inst<MYDET_ACT_NONE, true>(a, s) : launch_inst<MYDET_ACT_NONE, false>(a, s);
    }
}
