// Persistent, weight-resident form of the halo-tiled bf16 3x3 stride-1 convolution (see conv_halo.hip for the tiling).
//
// PMC profile of conv_halo.hip on the 64->64 @80x80 layer (profiles/r01_*): a wave lived ~30k cycles for ~4.6k cycles
// of MFMA - every workgroup paid its own prologue (address set-up, first-load latency) and store tail, one workgroup
// per CU, nothing overlapped. Here a workgroup is PERSISTENT: it keeps the whole weight slab of its output-channel
// block in LDS (loaded once) and walks pixel tiles j, j+G, j+2G, ...; the halo chunks of the NEXT tile are already in
// flight (LDS-DMA ring of NSH slots, lead NSH-1 chunks) while the current tile computes and stores.
//
// vmcnt accounting. Per wave the vector-memory queue holds, in order, halo pieces (LH per chunk) and epilogue stores
// (S = FM*FN per tile, issued UNCONDITIONALLY through a buffer descriptor: a lane that has nothing to store uses an
// out-of-range offset, so the count is exact). Iteration g does: wait(g) ; barrier ; issue(g+NSH-1) ; compute(g) ;
// [stores if g ends a tile]. Ops younger than L(g) at wait(g): L(g+1..g+NSH-2) plus the stores of every tile end
// among the previous NSH-1 iterations -> s_waitcnt vmcnt((NSH-2)*LH + k*S), k in {0..NSH-1}, selected per iteration.
// A residual tile (Bottleneck add) is read with ordinary loads in the epilogue; the compiler's own wait for them can
// only over-wait (drain older prefetches), never under-wait.
#include "common.h"

namespace yp {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ float silu_fast(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
template <int N> __device__ __forceinline__ void wait_vmc() {
    static_assert(N >= 0 && N < 64, "vmcnt range");
    __builtin_amdgcn_s_waitcnt((N & 0xF) | (7 << 4) | (0xF << 8) | (((N >> 4) & 3) << 14));
}
__device__ __forceinline__ int pswz(int row) { return ((row >> 2) & 1) << 1; }

template <int FM, int FN, int WGM, int WGN, int NSH, bool HAS_RES, bool OUT_F32>
__global__ __launch_bounds__(WGM * WGN * 64) void conv_halo_p_kernel(const ConvParams p, const int tiles_h, const int tiles_w,
                                                                    const int ntiles, const int G) {
    constexpr int NW = WGM * WGN;
    constexpr int TH = WGM * FM, BN = WGN * FN * 16;
    constexpr int HP = (TH + 2) * 18;
    constexpr int H_INSTR = (HP * 4 + 63) / 64;
    constexpr int LH = (H_INSTR + NW - 1) / NW;
    constexpr int HB = H_INSTR * 1024;
    constexpr int S = FM * FN;                        // stores per wave per tile
    constexpr unsigned OOB = 0x80000000u;
    static_assert(NSH == 3, "wait selection below is written for a 3-slot ring");
    static_assert((NSH - 2) * LH + 2 * S < 64, "vmcnt immediate");

    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char* const Hs = smem;                   // NSH halo slots
    unsigned char* const dump = smem + NSH * HB;      // 1 KiB landing zone of padding loads
    unsigned char* const Wres = dump + 1024;          // resident weights: [(chunk*9 + tap)][BN][32] bf16

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WGM, wn = wave / WGM;
    const int fr = lane & 15, fc = lane >> 4;
    const int nchunk = p.Cin >> 5;

    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, j = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int nt = bid % ntiles, j0 = bid / ntiles;
    const int n0 = nt * BN;
    const int B = p.M / (p.Ho * p.Wo);
    const int num_tiles = B * tiles_h * tiles_w;

    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, (int)p.y_bytes, 0x00020000);

    // bias first: its loads retire under the one-time vmcnt(0) below, so no compiler wait lands inside the tile loop
    float bias[FN][4];
#pragma unroll
    for (int a = 0; a < FN; ++a) {
        const int co = n0 + wn * (FN * 16) + a * 16 + fc * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) bias[a][r] = (co + r < p.Cout) ? p.bias[co + r] : 0.f;
    }

    // ---- resident weights: one pass of LDS-DMA pieces, row rg = (chunk*9 + tap)*BN + n -----------------------------
    {
        const int rows = 9 * nchunk * BN;
        const int ninstr = rows >> 4;
        for (int ii = wave; ii < ninstr; ii += NW) {
            const int s = ii * 64 + lane;
            const int rg = s >> 2, pc = s & 3;
            const int c8 = pc ^ pswz(rg);
            const int n = rg % BN, q = rg / BN;
            const int tap = q % 9, ch = q / 9;
            const unsigned voff = (unsigned)(((n0 + n) * p.Kpad + tap * p.Cin + ch * 32 + c8 * 8) * 2);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_void*)(Wres + ii * 1024), 16, voff, 0, 0, 0);
        }
    }

    // ---- issue side: halo pieces of (tile it_tile, chunk it_c) --------------------------------------------------------
    unsigned hconst[LH];
    auto set_tile = [&](int tile) {
        int t = tile;
        const int tw = t % tiles_w; t /= tiles_w;
        const int th = t % tiles_h;
        const int b = t / tiles_h;
        const int h0 = th * TH, w0 = tw * 16;
#pragma unroll
        for (int j = 0; j < LH; ++j) {
            const int ii = wave * LH + j;
            const int s = ii * 64 + lane;
            const int hp = s >> 2, pc = s & 3;
            const int c8 = pc ^ pswz(hp);
            const int hy = hp / 18, hx = hp - hy * 18;
            const int hi = h0 - 1 + hy, wi = w0 - 1 + hx;
            const bool ok = (tile < num_tiles) && (ii < H_INSTR) && (hp < HP) && ((unsigned)hi < (unsigned)p.H) && ((unsigned)wi < (unsigned)p.W);
            hconst[j] = ok ? (unsigned)((((b * p.H + hi) * p.W + wi) * p.x_stride + p.x_coff) * 2 + c8 * 16) : OOB;
        }
    };
    int it_tile = j0, it_c = 0, it_slot = 0;
    set_tile(it_tile);
    auto issue_next = [&]() {
        unsigned char* dst = Hs + it_slot * HB;
        const unsigned coff = (unsigned)it_c * 64u;
#pragma unroll
        for (int j = 0; j < LH; ++j) {
            const int ii = wave * LH + j;
            const unsigned voff = (hconst[j] == OOB) ? OOB : hconst[j] + coff;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_void*)((ii < H_INSTR) ? dst + ii * 1024 : dump), 16, voff, 0, 0, 0);
        }
        it_slot = (it_slot + 1 == NSH) ? 0 : it_slot + 1;
        if (++it_c == nchunk) {
            it_c = 0;
            it_tile += G;
            set_tile(it_tile);
        }
    };

#pragma unroll
    for (int s = 0; s < NSH - 1; ++s) issue_next();
    wait_vmc<0>();                      // weights + first chunks landed (once per workgroup)
    __builtin_amdgcn_s_barrier();

    int rd_slot = 0;
    unsigned epmask = 0;                // bit k: iteration (current-1-k) ended a tile
    bool first_iter = true;
    for (int tile = j0; tile < num_tiles; tile += G) {
        f32x4 acc[FN][FM];
#pragma unroll
        for (int a = 0; a < FN; ++a)
#pragma unroll
            for (int r = 0; r < FM; ++r) acc[a][r] = f32x4{bias[a][0], bias[a][1], bias[a][2], bias[a][3]};   // bias rides in the accumulator

        for (int c = 0; c < nchunk; ++c) {
            if (!first_iter) {
                const int k = __builtin_popcount(epmask & ((1u << (NSH - 1)) - 1u));
                if (k == 0) wait_vmc<(NSH - 2) * LH>();
                else if (k == 1) wait_vmc<(NSH - 2) * LH + S>();
                else wait_vmc<(NSH - 2) * LH + 2 * S>();
                __builtin_amdgcn_s_barrier();
            }
            first_iter = false;
            issue_next();
            epmask <<= 1;

            const unsigned char* hsl = Hs + rd_slot * HB;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                bf16x8 wf[3][FN];
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int a = 0; a < FN; ++a) {
                        const int rw = (c * 9 + ky * 3 + kx) * BN + wn * (FN * 16) + a * 16 + fr;
                        wf[ky][a] = *(const bf16x8*)(Wres + swz64((unsigned)(rw * 64 + fc * 16)));
                    }
#pragma unroll
                for (int hy = 0; hy < FM + 2; ++hy) {
                    const int hp = (wm * FM + hy) * 18 + kx + fr;
                    const bf16x8 xf = *(const bf16x8*)(hsl + swz64((unsigned)(hp * 64 + fc * 16)));
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) {
                        const int r = hy - ky;
                        if (r >= 0 && r < FM) {
#pragma unroll
                            for (int a = 0; a < FN; ++a)
                                acc[a][r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ky][a], xf, acc[a][r], 0, 0, 0);
                        }
                    }
                }
            }
            rd_slot = (rd_slot + 1 == NSH) ? 0 : rd_slot + 1;
        }

        // ---- epilogue of `tile`: exactly S buffer stores per wave ---------------------------------------------------------
        {
            int t = tile;
            const int tw = t % tiles_w; t /= tiles_w;
            const int th = t % tiles_h;
            const int b = t / tiles_h;
            const int wo = tw * 16 + fr;
            // residual tile first (ordinary loads, all in flight together; the compiler waits once before the first use)
            uint2 rres[FM][FN];
            if (HAS_RES) {
#pragma unroll
                for (int r = 0; r < FM; ++r) {
                    const int ho = th * TH + wm * FM + r;
                    const bool pix_ok = (ho < p.Ho) && (wo < p.Wo);
                    const unsigned m = (unsigned)((b * p.Ho + ho) * p.Wo + wo);
#pragma unroll
                    for (int a = 0; a < FN; ++a) {
                        const int co = n0 + wn * (FN * 16) + a * 16 + fc * 4;
                        rres[r][a] = (pix_ok && co < p.Cout)
                                         ? *(const uint2*)((const __bf16*)p.res + (size_t)m * p.res_stride + p.res_coff + co)
                                         : make_uint2(0u, 0u);
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < FM; ++r) {
                const int ho = th * TH + wm * FM + r;
                const bool pix_ok = (ho < p.Ho) && (wo < p.Wo);
                const unsigned m = (unsigned)((b * p.Ho + ho) * p.Wo + wo);
#pragma unroll
                for (int a = 0; a < FN; ++a) {
                    const int co = n0 + wn * (FN * 16) + a * 16 + fc * 4;
                    const bool ok = pix_ok && (co < p.Cout);
                    float v[4] = {acc[a][r][0], acc[a][r][1], acc[a][r][2], acc[a][r][3]};
                    if (p.act == ACT_SILU) silu4_packed(v);
                    if (HAS_RES) {
                        const uint2 rr = rres[r][a];
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
        epmask |= 1u;
    }
    wait_vmc<0>();
}

// ---------------------------------------------------------------------------------------------------------------
struct HaloPCfg { int FM, FN, WGM, WGN; const char* name; };
static const HaloPCfg kHaloP[] = {
    {4, 2, 4, 1, "conv_halo_p_kernel<4,2,4,1,3>"},   // 0: 16x16 px x 32 ch, 4 waves
    {8, 2, 2, 2, "conv_halo_p_kernel<8,2,2,2,3>"},   // 1: 16x16 px x 64 ch, 4 waves
    {4, 2, 4, 2, "conv_halo_p_kernel<4,2,4,2,3>"},   // 2: 16x16 px x 64 ch, 8 waves
    {4, 2, 2, 2, "conv_halo_p_kernel<4,2,2,2,3>"},   // 3:  8x16 px x 64 ch, 4 waves
    {4, 2, 2, 1, "conv_halo_p_kernel<4,2,2,1,3>"},   // (2 waves: not instantiated)
    {2, 2, 4, 1, "conv_halo_p_kernel<2,2,4,1,3>"},   // 4:  8x16 px x 32 ch, 4 waves
};
constexpr int kNumHaloP = 5;
static const int kHaloPIdx[kNumHaloP] = {0, 1, 2, 3, 5};

int conv_halo_p_num_cfgs() { return kNumHaloP; }

static size_t halo_p_lds(const HaloPCfg& k, int Cin) {
    const int NW = k.WGM * k.WGN, TH = k.WGM * k.FM, BN = k.WGN * k.FN * 16;
    (void)NW;
    const int HP = (TH + 2) * 18, H_INSTR = (HP * 4 + 63) / 64;
    return (size_t)3 * H_INSTR * 1024 + 1024 + (size_t)9 * (Cin / 32) * BN * 64;
}

bool conv_halo_p_cfg_valid(const ConvParams& p, int c) {
    if (c < 0 || c >= kNumHaloP) return false;
    if (p.ks != 3 || p.stride != 1 || p.pad != 1 || p.up != 1 || (p.Cin % 32) != 0 || (p.Kpad != 9 * p.Cin)) return false;
    if (p.x_bytes >= (1ull << 31) || p.w_bytes >= (1ull << 31) || p.y_bytes >= (1ull << 31)) return false;
    // the exact-count epilogue only has the vector store form
    if ((p.Cout & 3) || (p.y_stride & 3) || (p.y_coff & 3) || (p.res && ((p.res_stride & 3) || (p.res_coff & 3)))) return false;
    if (p.res && p.out_f32) return false;
    const HaloPCfg& k = kHaloP[kHaloPIdx[c]];
    const int BN = k.WGN * k.FN * 16, TH = k.WGM * k.FM;
    const int cpad = (p.Cout + 31) / 32 * 32;
    if (BN > cpad) return false;
    if (halo_p_lds(k, p.Cin) > 160 * 1024) return false;
    const long covered = (long)((p.Ho + TH - 1) / TH * TH) * ((p.Wo + 15) / 16 * 16);
    if (covered * 2 > (long)p.Ho * p.Wo * 3) return false;
    return true;
}

const char* conv_halo_p_kernel_name(int c) { return kHaloP[kHaloPIdx[c]].name; }

template <int FM, int FN, int WGM, int WGN, bool HAS_RES, bool OUT_F32>
static hipError_t launch_halo_p_var(const ConvParams& p, const HaloPCfg& k, hipStream_t st) {
    constexpr int TH = WGM * FM, BN = WGN * FN * 16;
    const size_t sh = halo_p_lds(k, p.Cin);
    const int B = p.M / (p.Ho * p.Wo);
    const int tiles_h = (p.Ho + TH - 1) / TH, tiles_w = (p.Wo + 15) / 16, ntiles = (p.Cout + BN - 1) / BN;
    const int num_tiles = B * tiles_h * tiles_w;
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(2, (160 * 1024) / sh));
    int G = (256 * per_cu) / ntiles;
    if (G < 1) G = 1;
    if (G > num_tiles) G = num_tiles;
    auto kern = conv_halo_p_kernel<FM, FN, WGM, WGN, 3, HAS_RES, OUT_F32>;
    static size_t attr = 0;
    if (sh > attr) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
        if (e != hipSuccess) return e;
        attr = 160 * 1024;
    }
    hipLaunchKernelGGL(kern, dim3(G * ntiles), dim3(WGM * WGN * 64), sh, st, p, tiles_h, tiles_w, ntiles, G);
    return hipGetLastError();
}

template <int FM, int FN, int WGM, int WGN>
static hipError_t launch_halo_p_one(const ConvParams& p, const HaloPCfg& k, hipStream_t st) {
    if (p.out_f32) return launch_halo_p_var<FM, FN, WGM, WGN, false, true>(p, k, st);     // (fp32 logits never carry a residual)
    if (p.res) return launch_halo_p_var<FM, FN, WGM, WGN, true, false>(p, k, st);
    return launch_halo_p_var<FM, FN, WGM, WGN, false, false>(p, k, st);
}

hipError_t launch_conv_halo_p(const ConvParams& p, int c, hipStream_t st) {
    const HaloPCfg& k = kHaloP[kHaloPIdx[c]];
    switch (c) {
        case 0: return launch_halo_p_one<4, 2, 4, 1>(p, k, st);
        case 1: return launch_halo_p_one<8, 2, 2, 2>(p, k, st);
        case 2: return launch_halo_p_one<4, 2, 4, 2>(p, k, st);
        case 3: return launch_halo_p_one<4, 2, 2, 2>(p, k, st);
        default: return launch_halo_p_one<2, 2, 4, 1>(p, k, st);
    }
}

}  // namespace yp
