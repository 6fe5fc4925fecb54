// One tile per compute unit: 3x3 stride-1 convolution for the mid-resolution layers (40x40 x 128 channels in v10-S), bf16.
//
// Those layers are 15 GFLOP over 51 200 pixels: every tiling of the implicit-GEMM families lands at 26-31 us (~500 TFLOP/s)
// because a 128- or 256-pixel tile re-reads its pixels nine times through 1-KiB LDS-DMA pieces (~900 pieces per CU, and a
// wave gets one piece through per ~240 cycles) and leaves 256 CUs with 200 or 400 tiles. Here the problem is cut into
// EXACTLY one tile per CU - TR x TC output pixels with B*ceil(H/TR)*ceil(W/TC) <= 256 (10 x 20 at 40x40, batch 32) - and
//   * the tile's input patch (TR+2) x (TC+2) x Cin is staged ONCE and stays in LDS (68 KB at Cin = 128),
//   * the weights stream through a 3-slot ring exactly once per CU, one tap row (3 taps x [BN][32]) per stage and barrier,
//     its pieces issued behind the MFMAs of the previous taps,
//   * the pixels of a tile are taken 16 at a time in flattened row-major order: a lane of the MFMA B operand reads its own
//     pixel's halo position, so a fragment may wrap around the tile's row end (no padding to a multiple of 16 columns).
// 8 waves as WGM (pixel groups) x WGN (32 output channels each), WGM * WGN = 8; a wave holds up to FMX pixel fragments x 2
// channel fragments.
#include "common.h"
#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace yp {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void lds_void;

template <int N> __device__ __forceinline__ void wait_vt1() {
    static_assert(N >= 0 && N < 64, "vmcnt range");
    __builtin_amdgcn_s_waitcnt((N & 0xF) | (7 << 4) | (0xF << 8) | (((N >> 4) & 3) << 14));
}
__device__ __forceinline__ int tswz(int row) { return ((row >> 2) & 1) << 1; }

constexpr int T1_FMX = 7;        // pixel fragments per wave (2 waves in M: tiles of up to 224 pixels)
constexpr int T1_NS = 3;         // weight ring slots; a stage is one tap ROW (3 taps x [BN][32]) so that a barrier covers 6*FMX MFMAs per wave

struct Tile1Geo { int TR, TC, tiles_h, tiles_w, nfr, ppc; };   // ppc: 16-pixel pieces per chunk plane of the patch

// debug (YOLOP_T1_CLOCKS=1, the <4,false,false> instantiation only): 100-MHz stamps of workgroup 0's waves - [0] start, [1] patch plane 0 and the
// first weight stages landed (first barrier passed), [2] main loop done, [3] epilogue stores issued
__device__ unsigned long long g_t1_clk[8][4];
__device__ int g_t1_abl;      // timing ablations of the stamped instantiation (YOLOP_T1_ABL, results become wrong): 1 no weight pieces in the loop, 2 no MFMAs, 4 no fragment reads

template <int WGN, bool HAS_RES, bool OUT_F32, bool CLK = false>
__global__ __launch_bounds__(512) void conv_tile1_kernel(const ConvParams p, const Tile1Geo g) {
#define T1_STAMP(i) do { if (CLK && blockIdx.x == 0 && (threadIdx.x & 63) == 0) g_t1_clk[threadIdx.x >> 6][i] = wall_clock64(); } while (0)
    T1_STAMP(0);
    const int abl = CLK ? g_t1_abl : 0;
    constexpr int NW = 8, WGM = NW / WGN, FN = 2;
    constexpr int BN = WGN * FN * 16;
    constexpr int WP = BN * 64 / 1024;                 // weight pieces per k-step
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int nchunk = p.Cin >> 5;
    unsigned char* const Xs = smem;                                        // [nchunk][ppc*16 px][32 ch]
    unsigned char* const Ws = smem + (size_t)nchunk * g.ppc * 1024;       // [NS][BN][32 ch]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WGM, wn = wave / WGM;
    const int fr = lane & 15, fc = lane >> 4;

    const int ntn = (p.Cout + BN - 1) / BN;
    int bid = blockIdx.x;
    const int nt = bid % ntn;
    int t = bid / ntn;
    const int tw = t % g.tiles_w; t /= g.tiles_w;
    const int th = t % g.tiles_h;
    const int b = t / g.tiles_h;
    const int r0 = th * g.TR, c0 = tw * g.TC, n0 = nt * BN;
    const int HC = g.TC + 2;                                                // patch pitch in pixels

    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, (int)p.y_bytes, 0x00020000);

    float bias[FN][4];
#pragma unroll
    for (int a = 0; a < FN; ++a) {
        const int co = n0 + wn * (FN * 16) + a * 16 + fc * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) bias[a][r] = (co + r < p.Cout) ? p.bias[co + r] : 0.f;
    }

    // ---- the input patch: chunk plane 0 now, plane c+1 at the start of chunk c's first stage (two stages before its first use;
    // the counted waits of the weight ring are stricter than needed there, so they cover it) --------------------------------------
    auto issue_x = [&](int ch) {
        const int HR = g.TR + 2, npx = HR * HC;
        for (int pi = wave; pi < g.ppc; pi += NW) {
            const int hp = pi * 16 + (lane >> 2), pc = lane & 3;
            const int c8 = pc ^ tswz(hp);
            const int hy = hp / HC, hx = hp - hy * HC;
            const int hi = r0 - 1 + hy, wi = c0 - 1 + hx;
            const bool ok = hp < npx && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            const unsigned voff = ok ? (unsigned)((((b * p.H + hi) * p.W + wi) * p.x_stride + p.x_coff + ch * 32 + c8 * 8) * 2) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_void*)(Xs + (ch * g.ppc + pi) * 1024), 16, voff, 0, 0, 0);
        }
    };
    issue_x(0);
    // ---- weight ring: stage st = chunk * 3 + ky holds the three taps (ky, 0..2) as [3][BN][32] --------------------------------------
    const int nst = nchunk * 3;
    constexpr int SW = 3 * BN * 64;                                          // bytes per stage
    constexpr int LPW = 3 * WP / 8 > 0 ? (3 * WP + 7) / 8 : 1;               // pieces per wave and stage (3 at BN = 128, 2 at BN = 64)
    unsigned wbase[LPW];
    bool wlive[LPW];
#pragma unroll
    for (int j = 0; j < LPW; ++j) {
        const int q = wave + j * NW;                                         // piece id within the stage: kx = q / WP, rows (q % WP) * 16 ..
        wlive[j] = q < 3 * WP;
        const int kx = q / WP, n = (q % WP) * 16 + (lane >> 2), pc = lane & 3;
        const int c8 = pc ^ tswz(n);
        wbase[j] = (unsigned)(((n0 + n) * p.Kpad + kx * p.Cin + c8 * 8) * 2);
    }
    const bool full = wlive[LPW - 1];                                        // this wave carries LPW pieces per stage (else LPW - 1)
    int it = 0;
    auto issue_piece = [&](int j, int stage) {
        const int ch = stage / 3, ky = stage - ch * 3;
        if (!wlive[j]) return;                                               // (wave-uniform; an out-of-range piece would zero-fill LDS)
        if (CLK && (abl & 1) && stage >= T1_NS - 1) return;
        const unsigned voff = (stage < nst) ? wbase[j] + (unsigned)((ky * 3 * p.Cin + ch * 32) * 2) : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_void*)(Ws + (stage % T1_NS) * SW + (wave + j * NW) * 1024), 16, voff, 0, 0, 0);
    };
#pragma unroll
    for (int s2 = 0; s2 < T1_NS - 1; ++s2) {
#pragma unroll
        for (int j = 0; j < LPW; ++j) issue_piece(j, it);
        ++it;
    }

    // ---- this wave's pixel fragments: flattened pixel -> patch position of tap (0,0) ------------------------------------------------
    const int npix = g.TR * g.TC;
    const int per_wave = (g.nfr + WGM - 1) / WGM;
    const int f0 = wm * per_wave;                                            // first fragment of this wave
    // LDS byte offset of this lane's 16-byte piece of pixel hp WITHOUT the swizzle: L = hp * 64 + fc * 16. The swizzle flips bit 5 where
    // bit 2 of hp is set, and bit 2 of hp is bit 8 of L, so the address of any tap is (L + tap * 64) ^ (((L + tap * 64) >> 3) & 32): one add
    // with a wave-uniform operand + two bit operations per fragment read. (Written as hp arithmetic the compiler spent ~9 VALU
    // instructions per read - 86 per tap and wave, 2600 per wave against 504 MFMAs: PMC showed the VALU issuing 48 % of the kernel's
    // cycles and the matrix pipe busy 30 %.)
    unsigned lbase[T1_FMX];
#pragma unroll
    for (int f = 0; f < T1_FMX; ++f) {
        int pp = (f0 + f) * 16 + fr;
        if (pp >= npix) pp = npix - 1;                                       // padding lanes read a valid pixel, never stored
        const int r = pp / g.TC, c = pp - r * g.TC;
        lbase[f] = (unsigned)((r * HC + c) * 64 + fc * 16);
    }
    unsigned wl[3][FN];                                                      // the same for the weight rows of (kx, a) inside a ring stage
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int a = 0; a < FN; ++a) {
            const int rw = kx * BN + wn * (FN * 16) + a * 16 + fr;
            wl[kx][a] = (unsigned)(rw * 64 + ((fc ^ tswz(rw)) * 16));
        }
    const int myf = max(0, min(per_wave, g.nfr - f0));                       // fragments that exist for this wave (wave-uniform)

    f32x4 acc[FN][T1_FMX];
#pragma unroll
    for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int f = 0; f < T1_FMX; ++f) acc[a][f] = f32x4{bias[a][0], bias[a][1], bias[a][2], bias[a][3]};

    if (full) wait_vt1<(T1_NS - 1) * LPW>();         // the patch has landed (only the ring's weight pieces may still be in flight)
    else wait_vt1<(T1_NS - 1) * (LPW - 1)>();
    __builtin_amdgcn_s_barrier();
    T1_STAMP(1);

    // Main loop, software-pipelined by one tap: the nine fragment reads of tap t+1 are issued BEFORE the 14 MFMAs of tap t, into a second
    // register set. (Left to the compiler, a pair of MFMAs followed each fragment read at a distance of one or two reads: every pair
    // waited out its own LDS round trip, and with both waves of a SIMD in the same phase behind the stage barrier the stage took the SUM of
    // its LDS time and its MFMA time - 2900 cycles for 1728 + 1344; in-kernel stamps: main loop 15.9 us of a 20-us workgroup.) The pixel
    // fragments of the next STAGE's first tap are prefetched across the barrier as well (the patch is resident; a chunk plane lands two
    // stages before its first use); its weight fragments are in the ring slot the barrier releases and are read right behind it.
    bf16x8 xf[2][T1_FMX], wf[2][FN];
    auto read_x = [&](bf16x8 (&dst)[T1_FMX], int stage, int kx) {
        const int ch = stage / 3, ky = stage - ch * 3;
        const unsigned char* xs = Xs + (size_t)ch * g.ppc * 1024;
        const unsigned tapb = (unsigned)((ky * HC + kx) * 64);
        if (CLK && (abl & 4)) return;
#pragma unroll
        for (int f = 0; f < T1_FMX; ++f) {
            const unsigned L = lbase[f] + tapb;
            dst[f] = *(const bf16x8*)(xs + (L ^ ((L >> 3) & 32u)));
        }
    };
    auto read_w = [&](bf16x8 (&dst)[FN], int stage, int kx) {
        const unsigned char* ws = Ws + (stage % T1_NS) * SW;
        if (CLK && (abl & 4)) return;
#pragma unroll
        for (int a = 0; a < FN; ++a) dst[a] = *(const bf16x8*)(ws + wl[kx][a]);
    };
    auto stage = [&](int st, auto Pc) {
        constexpr int P = decltype(Pc)::value;           // register set that holds this stage's first tap
        if (full) wait_vt1<(T1_NS - 2) * LPW>();         // this wave's pieces of stage st have landed
        else wait_vt1<(T1_NS - 2) * (LPW - 1)>();
        __builtin_amdgcn_s_barrier();
        const int ch = st / 3, ky = st - ch * 3;
        if (ky == 0 && ch + 1 < nchunk) issue_x(ch + 1);
        read_w(wf[P], st, 0);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            constexpr int dummy = 0; (void)dummy;
            const int cur = (P + kx) & 1, nxt = cur ^ 1;
            if (kx < 2) { read_w(wf[nxt], st, kx + 1); read_x(xf[nxt], st, kx + 1); }
            else if (st + 1 < nst) read_x(xf[nxt], st + 1, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (!(CLK && (abl & 2))) {
#pragma unroll
                for (int f = 0; f < T1_FMX; ++f)
#pragma unroll
                    for (int a = 0; a < FN; ++a) acc[a][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[cur][a], xf[cur][f], acc[a][f], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            // one piece of stage st + NS - 1 behind each tap's MFMAs: the ~300-cycle issue stall runs under their execution; the
            // slot it overwrites (stage st - 1) was finished with before this stage's barrier
            if (kx < LPW) issue_piece(kx, it);
        }
        ++it;
    };
    read_x(xf[0], 0, 0);
    for (int st = 0; st + 1 < nst; st += 2) {
        stage(st, std::integral_constant<int, 0>{});
        stage(st + 1, std::integral_constant<int, 1>{});
    }
    if (nst & 1) stage(nst - 1, std::integral_constant<int, 0>{});

    // ---- epilogue ---------------------------------------------------------------------------------------------------------------
    T1_STAMP(2);
#pragma unroll
    for (int f = 0; f < T1_FMX; ++f) {
        if (f >= myf) continue;
        const int pp = (f0 + f) * 16 + fr;
        const int r = pp / g.TC, c = pp - r * g.TC;
        const int ho = r0 + r, wo = c0 + c;
        const bool pix_ok = pp < npix && ho < p.Ho && wo < p.Wo;
        const unsigned m = (unsigned)((b * p.Ho + ho) * p.Wo + wo);
#pragma unroll
        for (int a = 0; a < FN; ++a) {
            const int co = n0 + wn * (FN * 16) + a * 16 + fc * 4;
            const bool ok = pix_ok && co < p.Cout;
            float v[4] = {acc[a][f][0], acc[a][f][1], acc[a][f][2], acc[a][f][3]};
            if (p.act == ACT_SILU) silu4_packed(v);
            if (HAS_RES) {
                const uint2 rr = ok ? *(const uint2*)((const __bf16*)p.res + (size_t)m * p.res_stride + p.res_coff + co) : make_uint2(0u, 0u);
                v[0] += __uint_as_float(rr.x << 16); v[1] += __uint_as_float(rr.x & 0xffff0000u);
                v[2] += __uint_as_float(rr.y << 16); v[3] += __uint_as_float(rr.y & 0xffff0000u);
            }
            if (OUT_F32) {
                const unsigned off = ok ? (m * (unsigned)p.y_stride + (unsigned)(p.y_coff + co)) * 4u : OOB;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, make_float4(v[0], v[1], v[2], v[3])), yrs, off, 0, 0);
            } else {
                const unsigned off = ok ? (m * (unsigned)p.y_stride + (unsigned)(p.y_coff + co)) * 2u : OOB;
                __attribute__((aligned(8))) __bf16 o[4] = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                __builtin_amdgcn_raw_buffer_store_b64(*(const __attribute__((ext_vector_type(2))) unsigned*)o, yrs, off, 0, 0);
            }
        }
    }
    T1_STAMP(3);
#undef T1_STAMP
}

// ---------------------------------------------------------------------------------------------------------------
// Wide-wave-tile form (configuration 2, round 4): the same one-tile-per-CU problem with the roles split. PMC and the in-kernel stamps put
// conv_tile1's main loop at 20.7 k cycles of LDS fragment reads against 16.1 k of MFMA per CU and layer (a wave tile of 7 pixel x 2 channel
// fragments reads 9 fragments per 14 MFMAs), plus a ~300-cycle issue stall per weight piece inside the consumers. Here
//   * waves 0-3 (one per SIMD) are CONSUMERS with 7 x 4 fragment tiles - 112 pixels x 64 output channels, 11 fragment reads per 28 MFMAs:
//     44 KB instead of 72 KB of LDS reads per tap and CU, below the 448 cycles the tap's MFMAs take;
//   * waves 4-7 (one per SIMD) are LOADERS: they issue the patch planes and the weight ring's pieces (6 per wave and stage) and nothing
//     else, so no consumer ever waits on a DMA issue slot;
//   * one workgroup barrier per stage (tap row) hands a landed stage to the consumers and a finished slot back to the loaders.
template <int FMW, bool HAS_RES, bool OUT_F32>
__global__ __launch_bounds__(512) void conv_tile1w_kernel(const ConvParams p, const Tile1Geo g) {
    constexpr int BN = 128, FN = 4, FM = FMW;
    static_assert(FMW == T1_FMX, "geometry");
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int nchunk = p.Cin >> 5;
    unsigned char* const Xs = smem;                                        // [nchunk][ppc*16 px][32 ch]
    unsigned char* const Ws = smem + (size_t)nchunk * g.ppc * 1024;       // [NS][3 taps][BN][32 ch]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fc = lane >> 4;
    int t = blockIdx.x;                                                     // (BN = Cout tile: launched only for Cout <= 128)
    const int tw = t % g.tiles_w; t /= g.tiles_w;
    const int th = t % g.tiles_h;
    const int b = t / g.tiles_h;
    const int r0 = th * g.TR, c0 = tw * g.TC;
    const int HC = g.TC + 2;
    const int nst = nchunk * 3;
    constexpr int SW = 3 * BN * 64;                                         // bytes per stage: 24 pieces
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, (int)p.y_bytes, 0x00020000);

    if (wave >= 4) {
        // ================================================= loaders =================================================
        const int lw = wave - 4;
        // patch plane ch: ppc pieces of 16 pixels, dealt round the four loader waves; every loader issues the same NUMBER of instructions
        // per plane (the surplus ones read out of range into the plane's padding rows), so that the counted waits below hold for all
        const int xpw = (g.ppc + 3) >> 2;
        auto issue_x = [&](int ch) {
            const int HR = g.TR + 2, npx = HR * HC;
            for (int j = 0; j < xpw; ++j) {
                const int pi = min(lw + 4 * j, g.ppc - 1);              // (a surplus instruction rewrites the plane's last piece with the same bytes)
                const int hp = pi * 16 + (lane >> 2), pc = lane & 3;
                const int c8 = pc ^ tswz(hp);
                const int hy = hp / HC, hx = hp - hy * HC;
                const int hi = r0 - 1 + hy, wi = c0 - 1 + hx;
                const bool ok = hp < npx && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
                const unsigned voff = ok ? (unsigned)((((b * p.H + hi) * p.W + wi) * p.x_stride + p.x_coff + ch * 32 + c8 * 8) * 2) : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_void*)(Xs + (ch * g.ppc + pi) * 1024), 16, voff, 0, 0, 0);
            }
        };
        unsigned wbase[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int q = lw + 4 * j;                                       // piece of the stage: tap kx = q / 8, rows (q % 8) * 16 ..
            const int kx = q >> 3, n = (q & 7) * 16 + (lane >> 2), pc = lane & 3;
            const int c8 = pc ^ tswz(n);
            wbase[j] = (unsigned)((n * p.Kpad + kx * p.Cin + c8 * 8) * 2);
        }
        auto issue_stage = [&](int stage) {
            const int ch = stage / 3, ky = stage - ch * 3;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const unsigned voff = (stage < nst) ? wbase[j] + (unsigned)((ky * 3 * p.Cin + ch * 32) * 2) : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_void*)(Ws + (stage % T1_NS) * SW + (lw + 4 * j) * 1024), 16, voff, 0, 0, 0);
            }
        };
        // issue order: X0 W0 W1 | (stage 0 barrier) X1 W2 | (stage 1) W3 | (stage 2) W4 | (stage 3) X2 W5 | ...
        issue_x(0);
        issue_stage(0);
        issue_stage(1);
        for (int st = 0; st < nst; ++st) {
            // stage st (and every plane issued before it) has landed once at most the 6 pieces of stage st + 1 are in flight
            wait_vt1<6>();
            __builtin_amdgcn_s_barrier();
            const int ch = st / 3, ky = st - ch * 3;
            if (ky == 0 && ch + 1 < nchunk) issue_x(ch + 1);                // lands two stages before its first use: the wait above covers it (in-order counter)
            issue_stage(st + 2);                                            // slot of stage st - 1: every consumer passed this barrier, so it is done with it
        }
        wait_vt1<0>();
        return;
    }

    // ================================================= consumers =================================================
    const int wm = wave & 1, wn = wave >> 1;
    float bias[FN][4];
#pragma unroll
    for (int a = 0; a < FN; ++a) {
        const int co = wn * (FN * 16) + a * 16 + fc * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) bias[a][r] = (co + r < p.Cout) ? p.bias[co + r] : 0.f;
    }
    const int npix = g.TR * g.TC;
    const int per_wave = (g.nfr + 1) >> 1;
    const int f0 = wm * per_wave;
    unsigned lbase[FM];
#pragma unroll
    for (int f = 0; f < FM; ++f) {
        int pp = (f0 + f) * 16 + fr;
        if (pp >= npix) pp = npix - 1;
        const int r = pp / g.TC, c = pp - r * g.TC;
        lbase[f] = (unsigned)((r * HC + c) * 64 + fc * 16);
    }
    unsigned wl[3][FN];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int a = 0; a < FN; ++a) {
            const int rw = kx * BN + wn * (FN * 16) + a * 16 + fr;
            wl[kx][a] = (unsigned)(rw * 64 + ((fc ^ tswz(rw)) * 16));
        }
    const int myf = max(0, min(per_wave, g.nfr - f0));
    f32x4 acc[FN][FM];
#pragma unroll
    for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int f = 0; f < FM; ++f) acc[a][f] = f32x4{bias[a][0], bias[a][1], bias[a][2], bias[a][3]};

    // Fragment reads are inline-asm ds_read_b128 with COUNTED lgkmcnt waits. Left to the compiler, the wait in front of a tap's MFMAs was
    // lgkmcnt(0): it also waited for the next tap's reads, issued just before - the software pipeline did not overlap anything (ablations
    // on the first form of this kernel: 8.5 us of MFMA and 6.3 us of fragment reads both fully exposed in a 28.6-us launch).
    bf16x8 xf[2][FM], wf[2][FN];
    const unsigned xs_l = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)Xs;
    const unsigned ws_l = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)Ws;
    auto lds_rd = [](bf16x8& dst, unsigned addr) { asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr)); };
    auto read_tap = [&](bf16x8 (&xd)[FM], bf16x8 (&wd)[FN], int stage, int kx) {       // 4 + 7 = 11 reads
        const int ch = stage / 3, ky = stage - ch * 3;
        const unsigned wsb = ws_l + (unsigned)((stage % T1_NS) * SW);
#pragma unroll
        for (int a = 0; a < FN; ++a) lds_rd(wd[a], wsb + wl[kx][a]);
        const unsigned xsb = xs_l + (unsigned)(ch * g.ppc * 1024);
        const unsigned tapb = (unsigned)((ky * HC + kx) * 64);
#pragma unroll
        for (int f = 0; f < FM; ++f) {
            const unsigned L = lbase[f] + tapb;
            lds_rd(xd[f], xsb + (L ^ ((L >> 3) & 32u)));
        }
    };
    static_assert(FM == 7 && FN == 4, "operand lists of the waits");
    // the set (xs, ws) has landed once at most `N` younger reads are outstanding; the registers pass through the wait, so no use moves above it
#define T1W_WAIT(N, xs_, ws_) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(xs_[0]), "+v"(xs_[1]), "+v"(xs_[2]), "+v"(xs_[3]), "+v"(xs_[4]), "+v"(xs_[5]), "+v"(xs_[6]), \
                                           "+v"(ws_[0]), "+v"(ws_[1]), "+v"(ws_[2]), "+v"(ws_[3]) : : "memory")
    auto mfma_tap = [&](bf16x8 (&xd)[FM], bf16x8 (&wd)[FN]) {
#pragma unroll
        for (int f = 0; f < FM; ++f)
#pragma unroll
            for (int a = 0; a < FN; ++a) acc[a][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wd[a], xd[f], acc[a][f], 0, 0, 0);
    };
    for (int st = 0; st < nst; ++st) {
        __builtin_amdgcn_s_barrier();
        read_tap(xf[0], wf[0], st, 0);
        read_tap(xf[1], wf[1], st, 1);
        T1W_WAIT(11, xf[0], wf[0]);
        mfma_tap(xf[0], wf[0]);
        read_tap(xf[0], wf[0], st, 2);              // (in-order issue: behind the MFMAs that read set 0)
        T1W_WAIT(11, xf[1], wf[1]);
        mfma_tap(xf[1], wf[1]);
        T1W_WAIT(0, xf[0], wf[0]);
        mfma_tap(xf[0], wf[0]);
    }
#undef T1W_WAIT
    // ---- epilogue ---------------------------------------------------------------------------------------------------------------
#pragma unroll
    for (int f = 0; f < FM; ++f) {
        if (f >= myf) continue;
        const int pp = (f0 + f) * 16 + fr;
        const int r = pp / g.TC, c = pp - r * g.TC;
        const int ho = r0 + r, wo = c0 + c;
        const bool pix_ok = pp < npix && ho < p.Ho && wo < p.Wo;
        const unsigned m = (unsigned)((b * p.Ho + ho) * p.Wo + wo);
#pragma unroll
        for (int a = 0; a < FN; ++a) {
            const int co = wn * (FN * 16) + a * 16 + fc * 4;
            const bool ok = pix_ok && co < p.Cout;
            float v[4] = {acc[a][f][0], acc[a][f][1], acc[a][f][2], acc[a][f][3]};
            if (p.act == ACT_SILU) silu4_packed(v);
            if (HAS_RES) {
                const uint2 rr = ok ? *(const uint2*)((const __bf16*)p.res + (size_t)m * p.res_stride + p.res_coff + co) : make_uint2(0u, 0u);
                v[0] += __uint_as_float(rr.x << 16); v[1] += __uint_as_float(rr.x & 0xffff0000u);
                v[2] += __uint_as_float(rr.y << 16); v[3] += __uint_as_float(rr.y & 0xffff0000u);
            }
            if (OUT_F32) {
                const unsigned off = ok ? (m * (unsigned)p.y_stride + (unsigned)(p.y_coff + co)) * 4u : OOB;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, make_float4(v[0], v[1], v[2], v[3])), yrs, off, 0, 0);
            } else {
                const unsigned off = ok ? (m * (unsigned)p.y_stride + (unsigned)(p.y_coff + co)) * 2u : OOB;
                __attribute__((aligned(8))) __bf16 o[4] = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                __builtin_amdgcn_raw_buffer_store_b64(*(const __attribute__((ext_vector_type(2))) unsigned*)o, yrs, off, 0, 0);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// host: pick TR x TC so that the whole problem is one round of at most 256 workgroups
static bool tile1_geometry(const ConvParams& p, int BN, Tile1Geo& g, size_t& lds) {
    const int WGM = 8 / (BN / 32);
    const int B = p.M / (p.Ho * p.Wo);
    const int ntn = (p.Cout + BN - 1) / BN;
    const int nchunk = p.Cin / 32;
    double best = 0;
    bool found = false;
    for (int TR = 1; TR <= p.Ho; ++TR)
        for (int TC = 4; TC <= p.Wo; ++TC) {
            const int px = TR * TC;
            if (px > WGM * T1_FMX * 16) break;
            const int tiles = B * ((p.Ho + TR - 1) / TR) * ((p.Wo + TC - 1) / TC) * ntn;
            if (tiles > 256) continue;
            const int ppc = ((TR + 2) * (TC + 2) + 15) / 16;
            const size_t sh = (size_t)nchunk * ppc * 1024 + (size_t)T1_NS * 3 * BN * 64;
            if (sh > 160 * 1024) continue;
            const int nfr = (px + 15) / 16;
            const int per_wave = (nfr + WGM - 1) / WGM;                    // fragments of the busiest pixel group
            // time ~ per_wave MFMA pairs per k-step; prefer fewer, then more CUs used
            const double score = 1000.0 / per_wave + tiles / 256.0 + (double)B * p.Ho * p.Wo * ntn / ((double)tiles * nfr * 16) * 0.1;
            if (score > best) { best = score; g = Tile1Geo{TR, TC, (p.Ho + TR - 1) / TR, (p.Wo + TC - 1) / TC, nfr, ppc}; lds = sh; found = true; }
        }
    return found;
}

int conv_tile1_num_cfgs() { return 3; }
const char* conv_tile1_kernel_name(int c) { return c == 0 ? "conv_tile1_kernel<4>" : c == 1 ? "conv_tile1_kernel<2>" : "conv_tile1w_kernel<7>"; }

bool conv_tile1_cfg_valid(const ConvParams& p, int c) {
    if (c < 0 || c >= 3) return false;
    if (p.ks != 3 || p.stride != 1 || p.pad != 1 || p.up != 1 || (p.Cin % 32) != 0 || p.Kpad != 9 * p.Cin || p.x2_C > 0) return false;
    if (p.x_bytes >= (1ull << 31) || p.w_bytes >= (1ull << 31) || p.y_bytes >= (1ull << 31)) return false;
    if ((p.Cout & 3) || (p.y_stride & 3) || (p.y_coff & 3) || (p.res && ((p.res_stride & 3) || (p.res_coff & 3)))) return false;
    if (p.res && p.out_f32) return false;
    const int BN = c == 1 ? 64 : 128;
    if (BN > (p.Cout + 31) / 32 * 32) return false;
    if (c == 2 && p.Cout > 128) return false;                // (the wide form has no output-channel tiling)
    Tile1Geo g;
    size_t lds;
    if (!tile1_geometry(p, BN, g, lds)) return false;
    return true;                                   // (whether one round of tiles pays is the autotuner's call)
}

template <int WGN, bool HAS_RES, bool OUT_F32>
static hipError_t launch_tile1_var(const ConvParams& p, hipStream_t st) {
    constexpr int BN = WGN * 32;
    Tile1Geo g;
    size_t sh;
    if (!tile1_geometry(p, BN, g, sh)) return hipErrorInvalidValue;
    const int B = p.M / (p.Ho * p.Wo);
    const int tiles = B * g.tiles_h * g.tiles_w * ((p.Cout + BN - 1) / BN);
    auto kern = conv_tile1_kernel<WGN, HAS_RES, OUT_F32>;
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
        if (e != hipSuccess) return e;
        attr = true;
    }
    if constexpr (WGN == 4 && !HAS_RES && !OUT_F32) {
        static const bool clocks = [] { const char* v = std::getenv("YOLOP_T1_CLOCKS"); return v && *v == '1'; }();
        if (clocks) {
            auto kc = conv_tile1_kernel<4, false, false, true>;
            { const char* v = std::getenv("YOLOP_T1_ABL"); const int ab = v ? atoi(v) : 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_t1_abl), &ab, sizeof(int)); }
            static bool attr_c = false;
            if (!attr_c) { (void)hipFuncSetAttribute((const void*)kc, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)); attr_c = true; }
            hipLaunchKernelGGL(kc, dim3(tiles), dim3(512), sh, st, p, g);
            hipError_t e = hipStreamSynchronize(st);
            if (e != hipSuccess) return e;
            unsigned long long h[8][4];
            (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_t1_clk), sizeof(h));
            double ph[3] = {0, 0, 0};
            for (int w = 0; w < 8; ++w) for (int i = 0; i < 3; ++i) ph[i] += (double)(h[w][i + 1] - h[w][i]) / 100.0 / 8.0;
            fprintf(stderr, "[tile1 clocks] Cin %d Cout %d %dx%d tiles %d (TR %d TC %d): fill %.2f us, main loop %.2f us, epilogue %.2f us (workgroup 0, mean of 8 waves)\n",
                    p.Cin, p.Cout, p.Ho, p.Wo, tiles, g.TR, g.TC, ph[0], ph[1], ph[2]);
            return hipSuccess;
        }
    }
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(512), sh, st, p, g);
    return hipGetLastError();
}

template <bool HAS_RES, bool OUT_F32>
static hipError_t launch_tile1w_var(const ConvParams& p, hipStream_t st) {
    Tile1Geo g;
    size_t sh;
    if (!tile1_geometry(p, 128, g, sh)) return hipErrorInvalidValue;
    const int B = p.M / (p.Ho * p.Wo);
    const int tiles = B * g.tiles_h * g.tiles_w;
    auto kern = conv_tile1w_kernel<T1_FMX, HAS_RES, OUT_F32>;
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
        if (e != hipSuccess) return e;
        attr = true;
    }
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(512), sh, st, p, g);
    return hipGetLastError();
}

hipError_t launch_conv_tile1(const ConvParams& p, int c, hipStream_t st) {
    if (c == 2) {
        if (p.out_f32) return launch_tile1w_var<false, true>(p, st);
        if (p.res) return launch_tile1w_var<true, false>(p, st);
        return launch_tile1w_var<false, false>(p, st);
    }
    if (c == 0) {
        if (p.out_f32) return launch_tile1_var<4, false, true>(p, st);
        if (p.res) return launch_tile1_var<4, true, false>(p, st);
        return launch_tile1_var<4, false, false>(p, st);
    }
    if (p.out_f32) return launch_tile1_var<2, false, true>(p, st);
    if (p.res) return launch_tile1_var<2, true, false>(p, st);
    return launch_tile1_var<2, false, false>(p, st);
}

}  // namespace yp
