// 3x3 convolution (pad 1, stride 1 or 2) with float32-exact split-bf16 operands and an LDS-RESIDENT INPUT PATCH (round 6).
//
// conv_igemm_b3_kernel (conv_igemm.hip) gathers, for every one of the 9 taps, the tile's 128 x 16 activations from global
// memory, cuts them into three bfloat16 pieces and stages them: every input element is loaded and split once per tap that
// uses it -- 2.25 times for a stride-2 layer, 9 times for a stride-1 layer -- and that staging, not the matrix pipe, bounds
// the launch (stride-2 128->256 @160^2: 0.36 ms of 0.66 are staging alone; profiles/HISTORY.md, round 5).
// Here a workgroup owns a SPATIAL tile of TH x TW = 8 x 16 output pixels (128 MFMA rows; 16 x 8 / 32 x 4 strip tiles for the remainder
// columns of maps that are 16 n + 8 / + 4 pixels wide, same launch) x BN output channels.  Per
// 16-channel slab it loads the tile's input patch (17 x 33 pixels at stride 2, 10 x 18 at stride 1) ONCE, splits it once into
// three bf16 planes in LDS, and the 9 taps read their MFMA A fragments straight out of that patch at a tap-dependent offset --
// no im2col tile is ever written.  The weights come pre-split (mydet_split_bf16_f32: the planes conv_igemm_b3_kernel uses,
// slab kt = tap * Cin/16 + slab) and never touch LDS: a wave owns 32 output channels (all 128 rows of the tile at BN = 128), and
// its MFMA B fragment of a (slab, tap, plane) is ONE coalesced 1 KB load -- the planes store each 32-row block as the 64 16-byte
// units of exactly that fragment -- requested three taps ahead.  So the only workgroup barriers are the two around the patch
// refresh of a slab (the first form staged the weights through LDS: two barriers per tap, waves waiting 57 % of their cycles).
//   global loads + split arithmetic per output pixel and slab: 561 / 128 = 4.4 pixels (stride 2; 9 before), 180 / 128 = 1.4
//   (stride 1; 9 before).
// Patch layout (one plane; 32 bytes = 16 bf16 per position, the two 16-byte halves swapped where sigma = 1):
//   stride 2: position = py * 36 + (px & 1) * 17 + (px >> 1), sigma = ((px >> 1) >> 3) & 1   (columns split by parity: the 16
//             pixels an MFMA row block reads for one tap are consecutive positions)
//   stride 1: position = py * 24 + px, sigma = (px >> 3) & 1
//   found by exhaustive search (tools/r06/p3_layout_search.py) over (row length, swizzle) against the ds_read_b128 lane groups of MI355X_MICROARCH.md: every
//   fragment read of every tap touches sixteen distinct 16-byte bank slots per lane group (conflict-free); the staging
//   writes of 4 consecutive positions are 128 contiguous bytes.
// Arithmetic: exactly conv_igemm_b3_kernel's (six piece products per k-step, small ones first, float32 accumulation), but
// the K order is (slab, tap) instead of (tap, slab): results agree to float32 round-off, both are held to 2e-5 * max|y|
// against float64 (tests/test_gpu_kernels.py).
// Replaces the ATen conv2d / batch_norm / leaky_relu chain of models/modules.py:76-95 for the stride-2 layers of
// models/backbones.py:14-30 and the 32 -> 64 layer of the first DarkBlock.
#include <cstdlib>

#include "common.h"

namespace {

constexpr unsigned OOB = 0xFFFFFFFFu;
constexpr int P3_COUT_PAD = 256;                 // rows of the weight planes (split_bf16_kernel)

struct P3Args {
    const float *x, *scale, *shift, *res;
    const unsigned short *wsplit;
    float *y;
    int64_t ldx, ldr, ldy;
    int B, H, W, Cin, Cout, Ho, Wo;
    int tx_n, ty_n, ntn, nblk;                   // 8 x 16 tiles per image row / column, channel tiles, workgroups
    int main_tiles, tiles_img;                   // tx_n * ty_n; + the strip tiles of the remainder columns (STRIP != 0)
};

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t p3_rsrc(const void *base, int64_t bytes) {
    const uint64_t a = (uint64_t)base;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    const int64_t capped = bytes > 0x7FFFFFF0ll ? 0x7FFFFFF0ll : bytes;
    const int n = __builtin_amdgcn_readfirstlane((int)capped);
    return __builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)hi << 32) | lo), 0, n, 0x00020000);
}

__device__ __forceinline__ void p3_split3(const f32x4 v, bf16x4 &p0, bf16x4 &p1, bf16x4 &p2) {       // == conv_igemm.hip: split3
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const __bf16 h0 = (__bf16)v[e];
        const float r1 = v[e] - (float)h0;
        const __bf16 h1 = (__bf16)r1;
        const float r2 = r1 - (float)h1;
        p0[e] = h0; p1[e] = h1; p2[e] = (__bf16)r2;
    }
}

// Tile shapes.  SHAPE 0: 8 rows x 16 columns of output pixels (both strides).  SHAPE 1 / 2 (stride 2): 16 x 8 and 32 x 4 STRIP tiles for
// the remainder columns of a map whose width is 16 n + 8 / 16 n + 4 (Darknet-53 at 640^2: the 40- and 20-pixel maps), run by the
// last workgroups of the same launch.  TWL = log2(tile width); PH = patch rows; ROWLEN = positions per patch row; PJ0 = positions
// of the even columns (stride 2: the odd columns follow them).  sigma (the 16-byte half swap of a position) = bit 3 of the column
// index j, plus bit 1 of the patch row for the strip shapes: each found conflict-free by tools/r06/p3_layout_search.py.
template <int S, int SHAPE> struct P3Geom;
template <> struct P3Geom<2, 0> { static constexpr int TH = 8, TWL = 4, PH = 17, ROWLEN = 36, PJ0 = 17; };
template <> struct P3Geom<1, 0> { static constexpr int TH = 8, TWL = 4, PH = 10, ROWLEN = 24, PJ0 = 0; };
template <> struct P3Geom<2, 1> { static constexpr int TH = 16, TWL = 3, PH = 33, ROWLEN = 20, PJ0 = 9; };
template <> struct P3Geom<2, 2> { static constexpr int TH = 32, TWL = 2, PH = 65, ROWLEN = 9, PJ0 = 5; };
template <int SHAPE> __device__ __forceinline__ int p3_sigma(int py, int j) { return SHAPE == 0 ? (j >> 3) & 1 : ((j >> 3) + (py >> 1)) & 1; }

// One tile: image b, output rows oy0.., columns ox0.., output channels n0...  S: stride.  BN: output channels per workgroup
// (64 | 128): waves = (4 / (BN / 32)) row groups x (BN / 32) column blocks of 32.
template <int S, int BN, int SHAPE, int ACT, bool RES>
__device__ __forceinline__ void p3_tile(const P3Args &p, char *patch, const int b, const int oy0, const int ox0, const int n0) {
    typedef P3Geom<S, SHAPE> G;
    constexpr int TH = G::TH, TWL = G::TWL, TW = 1 << TWL, RPB = 32 / TW, ROWB = 32;        // RPB: output rows per 32-row MFMA block
    constexpr int WN = BN / 32, WM = 4 / WN, TM = 4 / WM;       // wave (wm, wn): 32 * TM rows x 32 columns
    constexpr int PH = G::PH, ROWLEN = G::ROWLEN, PJ0 = G::PJ0;
    constexpr int PW = S * (TW - 1) + 3;
    static_assert(TH * TW == 128 && PH == S * (TH - 1) + 3 && PW <= (S == 2 ? 2 * PJ0 - 1 : ROWLEN), "patch geometry");
    constexpr int NPOS = PH * ROWLEN;
    constexpr int PLANE_P = NPOS * ROWB;
    constexpr int NCH = (NPOS * 4 + 255) / 256;      // 16-byte float4 chunks of the patch per thread and slab

    const int tid = threadIdx.x;
    const int iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;

    // ---- descriptors.  x: the buffer starts one row + one pixel before image b, so that a patch origin of (-1, -1) is a
    // non-negative offset (the range check sees the voffset only; those addresses are masked to OOB below and never touched)
    const int64_t img = (int64_t)p.H * p.W * p.ldx;
    const int64_t lead = (int64_t)(p.W + 1) * p.ldx;
    const __amdgpu_buffer_rsrc_t xr = p3_rsrc(p.x + b * img - lead, ((p.B - b) * img + lead) * 4);
    const int CoutP = (p.Cout + P3_COUT_PAD - 1) / P3_COUT_PAD * P3_COUT_PAD;
    const int nsl = p.Cin >> 4;                      // 16-channel slabs
    const __amdgpu_buffer_rsrc_t wr = p3_rsrc(p.wsplit, (int64_t)9 * nsl * 3 * CoutP * 32);

    // ---- patch staging roles: chunk q = tid + 256 i = (position q >> 2, channel quad q & 3)
    unsigned goff[NCH];                              // byte offset of the chunk's pixel (+ quad) in xr, OOB outside the image / padding
    int ldst[NCH];                                   // byte offset of the chunk in a patch plane
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int q = tid + 256 * i, pos = q >> 2, sc = q & 3;
        const int py = pos / ROWLEN, rem = pos - py * ROWLEN;
        int px, sig;
        bool ok = pos < NPOS;
        if (S == 2) {
            const int par = rem >= PJ0 ? 1 : 0, j = rem - par * PJ0;
            px = 2 * j + par;
            sig = p3_sigma<SHAPE>(py, j);
            ok = ok && px < PW && j < (par ? PJ0 - 1 : PJ0);
        } else {
            px = rem;
            sig = p3_sigma<SHAPE>(py, px);
            ok = ok && px < PW;
        }
        const int iy = iy0 + py, ix = ix0 + px;
        ok = ok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        goff[i] = ok ? (unsigned)((((int64_t)iy * p.W + ix) * p.ldx + sc * 4 + lead) * 4) : OOB;
        ldst[i] = pos < NPOS ? pos * ROWB + (((sc >> 1) ^ sig) * 16) + (sc & 1) * 8 : -1;
    }
    // ---- compute role
    const int wave = tid >> 6, lane = tid & 63;
    const int wm = wave / WN, wn = wave % WN;
    const int fr = lane & 31, fh = lane >> 5;
    const int oxl = fr & (TW - 1);
    int apos[TM], apy[TM];                           // patch position / patch row of the lane's row for tap (0, 0)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int oyl = wm * (TM * RPB) + i * RPB + (fr >> TWL);
        apy[i] = S * oyl;
        apos[i] = apy[i] * ROWLEN + oxl;
    }
    // weights: the lane's 16-byte unit of the wave's 32-row block in a (slab, plane) piece of the planes (split_bf16_kernel)
    const unsigned boff = (unsigned)((n0 + wn * 32) * 32 + (2 * fr + (fh ^ ((fr >> 2) & 1))) * 16);
    const unsigned plane_bytes = (unsigned)CoutP * 32u, slab_bytes = 3u * plane_bytes;
    f32x16 acc[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int nch = n0 + wn * 32 + fr;               // the lane's output channel
    const float pscl = p.scale ? p.scale[nch < p.Cout ? nch : 0] : 1.0f;
    const float psft = p.shift ? p.shift[nch < p.Cout ? nch : 0] : 0.0f;

    f32x4 preg[NCH];
    bf16x8 breg[3][3];                               // B fragments of three (slab, tap) steps in flight
    auto load_patch = [&](int cs) {
        const unsigned coff = (unsigned)cs * 64u;    // 16 channels * 4 bytes
#pragma unroll
        for (int i = 0; i < NCH; ++i)
            preg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, (goff[i] != OOB && cs < nsl) ? goff[i] + coff : OOB, 0, 0));
    };
    auto store_patch = [&]() {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            if (ldst[i] < 0) continue;
            bf16x4 q0, q1, q2;
            p3_split3(preg[i], q0, q1, q2);
            char *d = patch + ldst[i];
            *reinterpret_cast<bf16x4 *>(d) = q0;
            *reinterpret_cast<bf16x4 *>(d + PLANE_P) = q1;
            *reinterpret_cast<bf16x4 *>(d + 2 * PLANE_P) = q2;
        }
    };
    auto load_b = [&](int cs, int tap, bf16x8 (&brg)[3]) {             // weights of (slab cs, tap): slab kt = tap * nsl + cs of the planes
        const unsigned kt = (unsigned)(tap * nsl + cs);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
            brg[pl] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wr, cs < nsl ? boff : OOB,
                                                     __builtin_amdgcn_readfirstlane(kt * slab_bytes + (unsigned)pl * plane_bytes), 0));
    };
    auto compute = [&](int tap, const bf16x8 (&bf)[3]) {
        const int kh = tap / 3, kw = tap - kh * 3;
        bf16x8 af[TM][3];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            int pos, sig;
            if (S == 2) {
                pos = apos[i] + kh * ROWLEN + (kw & 1) * PJ0 + (kw >> 1);
                sig = p3_sigma<SHAPE>(apy[i] + kh, oxl + (kw >> 1));
            } else {
                pos = apos[i] + kh * ROWLEN + kw;
                sig = p3_sigma<SHAPE>(apy[i] + kh, oxl + kw);
            }
            const char *a = patch + pos * ROWB + ((fh ^ sig) * 16);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) af[i][pl] = *reinterpret_cast<const bf16x8 *>(a + pl * PLANE_P);
        }
        constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};          // small piece products first
#pragma unroll
        for (int tt = 0; tt < 6; ++tt)
#pragma unroll
            for (int i = 0; i < TM; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][PA[tt]], bf[PB[tt]], acc[i], 0, 0, 0);
    };

    load_patch(0);
    load_b(0, 0, breg[0]);
    load_b(0, 1, breg[1]);
    load_b(0, 2, breg[2]);
    for (int cs = 0; cs < nsl; ++cs) {
        if (cs > 0) __syncthreads();                 // every wave is done with the previous slab's patch
        store_patch();
        load_patch(cs + 1);                          // in flight under the nine taps below
        __syncthreads();
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            compute(tap, breg[tap % 3]);
            const int t3 = tap + 3;
            load_b(cs + t3 / 9, t3 % 9, breg[tap % 3]);
        }
    }

    // ---- epilogue: lane = output channel, register = output pixel of the RPB-row x TW-column block (row r of the block:
    // (r >> TWL, r & (TW - 1))); scale / shift / activation / residual as conv_igemm's
    const int64_t opix = (int64_t)p.Ho * p.Wo;       // (descriptors per image: byte offsets stay inside one image's output)
    const __amdgpu_buffer_rsrc_t yr = p3_rsrc(p.y + b * opix * p.ldy, opix * p.ldy * 4);
    const __amdgpu_buffer_rsrc_t rr = p3_rsrc(RES ? p.res + b * opix * p.ldr : p.y, opix * (RES ? p.ldr : p.ldy) * 4);
    const unsigned ldy4 = (unsigned)p.ldy * 4u, ldr4 = (unsigned)p.ldr * 4u;
    {
        const int n = nch;
        const bool nok = n < p.Cout;
        const float scl = pscl, sft = psft;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int oyb = oy0 + wm * (TM * RPB) + i * RPB;     // first output row of the block
            float rv[16];
            unsigned off_y[16], off_r[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dr = (r & 3) + 8 * (r >> 2) + 4 * fh;
                const int oy = oyb + (dr >> TWL), ox = ox0 + (dr & (TW - 1));
                const bool ok = nok && oy < p.Ho && ox < p.Wo;
                const unsigned pix = (unsigned)(oy * p.Wo + ox);
                off_y[r] = ok ? pix * ldy4 + (unsigned)n * 4u : OOB;
                off_r[r] = ok ? pix * ldr4 + (unsigned)n * 4u : OOB;
                if (RES) rv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, off_r[r], 0, 0));
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[i][r] * scl + sft;
                if (ACT == MYDET_ACT_LEAKY) v = v > 0.0f ? v : v * 0.1f;
                if (ACT == MYDET_ACT_SWISH) v = v * mydet_sigmoid_fast(v);
                if (RES) v += rv[r];
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yr, off_y[r], 0, 0);
            }
        }
    }
}

// The launch: workgroup id -> (channel tile, image, tile of the image); ids of one spatial tile are neighbours (the patch is shared
// through L2).  STRIP != 0: the image's last tiles are strip tiles (shape STRIP) over the columns behind the tx_n * 16 whole ones.
template <int S, int BN, int STRIP, int ACT, bool RES>
__global__ __launch_bounds__(256, 2) void conv_p3_kernel(const P3Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem_p3[];
    const int id = mydet_xcd_remap(blockIdx.x, p.nblk);
    const int nt = id % p.ntn, t = id / p.ntn;
    const int b = t / p.tiles_img, r = t - b * p.tiles_img;
    if constexpr (STRIP != 0) {
        if (r >= p.main_tiles) {                     // (uniform per workgroup)
            p3_tile<S, BN, STRIP, ACT, RES>(p, smem_p3, b, (r - p.main_tiles) * P3Geom<S, STRIP>::TH, p.tx_n * 16, nt * BN);
            return;
        }
    }
    const int ty = r / p.tx_n, tx = r - ty * p.tx_n;
    p3_tile<S, BN, 0, ACT, RES>(p, smem_p3, b, ty * 8, tx * 16, nt * BN);
}

template <int S, int SHAPE> constexpr int p3_lds() { return 3 * P3Geom<S, SHAPE>::PH * P3Geom<S, SHAPE>::ROWLEN * 32; }

template <int S, int BN, int STRIP, int ACT, bool RES>
int p3_launch(const P3Args &p, hipStream_t st) {
    constexpr int L0 = p3_lds<S, 0>(), L1 = p3_lds<S, STRIP>(), LDS = L0 > L1 ? L0 : L1;
    auto kern = &conv_p3_kernel<S, BN, STRIP, ACT, RES>;
    static bool attr = false;
    if (!attr) {
        const hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return (int)e;
        attr = true;
    }
    hipLaunchKernelGGL(kern, dim3(p.nblk), dim3(256), LDS, st, p);
    return mydet_launch_status();
}

template <int S, int BN, int STRIP>
int p3_dispatch(const P3Args &p, int act, bool res, hipStream_t st) {
    if (act == MYDET_ACT_LEAKY) return res ? p3_launch<S, BN, STRIP, MYDET_ACT_LEAKY, true>(p, st) : p3_launch<S, BN, STRIP, MYDET_ACT_LEAKY, false>(p, st);
    if (act == MYDET_ACT_NONE) return res ? p3_launch<S, BN, STRIP, MYDET_ACT_NONE, true>(p, st) : p3_launch<S, BN, STRIP, MYDET_ACT_NONE, false>(p, st);
    return MYDET_E_UNSUPP;
}

}  // namespace

extern "C" int mydet_conv3x3_p3_f32(const float *x, int64_t ldx, const uint16_t *w_planes, const float *scale, const float *shift,
                                    const float *residual, int64_t ldr, float *y, int64_t ldy, int B, int H, int W, int Cin, int Cout,
                                    int stride, int act, void *stream) {
    if (!x || !w_planes || !y || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return MYDET_E_BADARG;
    if ((stride != 1 && stride != 2) || (Cin & 15) || (ldx & 3) || ldx < Cin || ldy < Cout || (residual && ldr < Cout)) return MYDET_E_UNSUPP;
    if (((uintptr_t)x & 15) || ((uintptr_t)w_planes & 15)) return MYDET_E_BADARG;
    P3Args p;
    p.x = x; p.scale = scale; p.shift = shift; p.res = residual; p.wsplit = w_planes; p.y = y;
    p.ldx = ldx; p.ldr = ldr; p.ldy = ldy;
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
    p.Ho = (H + 2 - 3) / stride + 1; p.Wo = (W + 2 - 3) / stride + 1;
    // 32-bit byte offsets inside the kernel, relative to the workgroup's image: one image's input and output stay below 2 GB
    if ((int64_t)(H + 1) * (W + 1) * ldx * 4 > 0x7FFFFFF0ll || (int64_t)p.Ho * p.Wo * (ldy > ldr ? ldy : ldr) * 4 > 0x7FFFFFF0ll) return MYDET_E_UNSUPP;
    const char *fe = getenv("MYDET_P3_FORM");        // (experiments) 1 = 64-channel tiles whatever Cout
    const bool wide = Cout > 64 && !(fe && atoi(fe) == 1);
    const int BN = wide ? 128 : 64;
    // tile plan: whole 8 x 16 tiles; a remainder of 8 / 4 columns (stride 2) goes to 16 x 8 / 32 x 4 strip tiles of the same launch,
    // any other remainder to a ragged last column of 8 x 16 tiles.  MYDET_P3_STRIP=0: always the ragged column (A/B)
    const int rem = p.Wo & 15;
    const char *se = getenv("MYDET_P3_STRIP");
    const int strip = (stride == 2 && !(se && atoi(se) == 0)) ? (rem == 8 ? 1 : rem == 4 ? 2 : 0) : 0;
    p.tx_n = strip ? p.Wo / 16 : (p.Wo + 15) / 16;
    p.ty_n = (p.Ho + 7) / 8; p.ntn = (Cout + BN - 1) / BN;
    p.main_tiles = p.tx_n * p.ty_n;
    p.tiles_img = p.main_tiles + (strip == 1 ? (p.Ho + 15) / 16 : strip == 2 ? (p.Ho + 31) / 32 : 0);
    const int64_t nblk = (int64_t)B * p.tiles_img * p.ntn;
    if (nblk > 0x7FFFFFFF || nblk <= 0) return MYDET_E_UNSUPP;
    p.nblk = (int)nblk;
    hipStream_t st = (hipStream_t)stream;
    const bool res = residual != nullptr;
    if (stride == 2) {
        if (strip == 1) return wide ? p3_dispatch<2, 128, 1>(p, act, res, st) : p3_dispatch<2, 64, 1>(p, act, res, st);
        if (strip == 2) return wide ? p3_dispatch<2, 128, 2>(p, act, res, st) : p3_dispatch<2, 64, 2>(p, act, res, st);
        return wide ? p3_dispatch<2, 128, 0>(p, act, res, st) : p3_dispatch<2, 64, 0>(p, act, res, st);
    }
    return wide ? p3_dispatch<1, 128, 0>(p, act, res, st) : p3_dispatch<1, 64, 0>(p, act, res, st);
}
