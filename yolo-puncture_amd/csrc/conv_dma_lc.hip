// Loader / consumer form of the persistent LDS-DMA implicit-GEMM convolution (bf16, see conv_dma_p.hip for the ring).
//
// Why. PMC + microbenchmarks (profiles/r01_micro_dma_bench.txt): one wave gets one 1-KiB LDS-DMA piece
// through about every 240 cycles (the chip-wide rate grows with the number of ISSUING waves: 9 / 17 / 32 / 40 / 43 B/clk/CU
// at 2 / 4 / 8 / 12 / 16 waves), and while a wave is issuing its pieces it issues no MFMA. In conv_dma_p every wave does
// both: 3-4 pieces per k-step block it for roughly a thousand cycles around 256 cycles of its own matrix work, and the
// two waves of a SIMD do so in lockstep behind the per-k-step barrier (32 % matrix-pipe busy).
//
// Here the roles are split. A workgroup has NC consumer waves (MFMA only: fragment reads, matrix work, epilogue) and NL
// loader waves (LDS-DMA only). Per k-step g and one workgroup barrier:
//     loader:    wait until ITS pieces of stage g have landed (counted vmcnt) ; barrier(g) ; issue its pieces of stage g+NS-1
//     consumer:  barrier(g) ; read the fragments of stage g ; MFMAs                       [; epilogue stores at a tile end]
// Stage g+NS-1 overwrites the slot of stage g-1, whose last fragment reads were consumed by MFMAs before the consumers
// reached barrier(g). Loaders have nothing but LDS-DMA on their vector-memory queue, so the wait is vmcnt((NS-2)*pieces per stage);
// consumers have nothing but their own stores and never wait for them inside the loop.
#include "common.h"
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace yp {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ float silu_lc(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
template <int N> __device__ __forceinline__ void wait_vml() {
    static_assert(N >= 0 && N < 64, "vmcnt range");
    __builtin_amdgcn_s_waitcnt((N & 0xF) | (7 << 4) | (0xF << 8) | (((N >> 4) & 3) << 14));
}
template <int BK> __device__ __forceinline__ int swz_l(int row) {
    return BK == 32 ? (((row >> 2) & 1) << 1) : ((row >> 1) & 7);
}

template <int BM, int BN, int WGM, int WGN, int NL, int BK, int NS, bool HAS_RES, bool OUT_F32>
__global__ __launch_bounds__((WGM * WGN + NL) * 64) void conv_dma_lc_kernel(const ConvParams p, const int mtiles, const int ntiles, const int G) {
    constexpr int NC = WGM * WGN;
    constexpr int CPR = BK / 8;
    constexpr int RB = BK * 2;
    constexpr int A_INSTR = BM * CPR / 64;
    constexpr int W_INSTR = BN * CPR / 64;
    constexpr int PIECES = A_INSTR + W_INSTR;          // 1-KiB pieces per stage; loader lw takes pieces lw, lw+NL, lw+2NL, ...
    constexpr int MAXP = (PIECES + NL - 1) / NL;
    constexpr int FULL = PIECES - (MAXP - 1) * NL;     // loaders [0, FULL) carry MAXP pieces, the others MAXP-1
    constexpr int SB = (BM + BN) * RB;
    constexpr int WM = BM / WGM, WN = BN / WGN, FM = WM / 16, FN = WN / 16;
    constexpr int KSUB = BK / 32;
    static_assert(MAXP >= 1 && (NS - 2) * MAXP < 64, "vmcnt immediate");
    constexpr unsigned OOB = 0x80000000u;

    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, j = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int nt = bid % ntiles, j0 = bid / ntiles;
    const int n0 = nt * BN;
    const int nk = p.Kpad / BK;

    if (wave >= NC) {
        // ================================================= loader =================================================
        const int lw = wave - NC;
        const int HoWo = p.Ho * p.Wo;
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t xrs2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x2_C > 0 ? p.x2 : p.x), 0, (int)(p.x2_C > 0 ? p.x2_bytes : p.x_bytes), 0x00020000);
        const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)p.w_bytes, 0x00020000);
        // per-piece constants: piece q < A_INSTR is a pixel piece (16 rows of the im2col tile), else a weight piece
        unsigned pconst[MAXP], pmask[MAXP], pconst2[MAXP];     // pconst2: the pixel piece's rows in the folded-upsample source
        auto set_tile = [&](int mt) {
#pragma unroll
            for (int j = 0; j < MAXP; ++j) {
                const int q = lw + j * NL;
                if (q >= A_INSTR) continue;
                const int s = q * 64 + lane;
                const int row = s / CPR, pc = s - row * CPR;
                const int c = pc ^ swz_l<BK>(row);
                const int m = mt * BM + row;
                unsigned mask = 0, base = 0;
                if (mt < mtiles && m < p.M) {
                    if (p.ks == 1) {
                        base = (unsigned)(m * p.x_stride + p.x_coff) * 2u;
                        mask = 1u;
                        if (p.x2_C > 0) {
                            const int b = m / HoWo, r = m - b * HoWo;
                            const int ho = r / p.Wo, wo = r - ho * p.Wo;
                            pconst2[j] = (unsigned)(((b * p.x2_H + (ho >> 1)) * p.x2_W + (wo >> 1)) * p.x2_stride + p.x2_coff) * 2u + (unsigned)c * 16u;
                        }
                    } else {
                        const int b = m / HoWo, r = m - b * HoWo;
                        const int ho = r / p.Wo, wo = r - ho * p.Wo;
                        const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
                        base = (unsigned)(((b * p.H + hi0) * p.W + wi0) * p.x_stride + p.x_coff) * 2u;
                        for (int ky = 0; ky < p.ks; ++ky)
                            for (int kx = 0; kx < p.ks; ++kx)
                                if ((unsigned)(hi0 + ky) < (unsigned)p.H && (unsigned)(wi0 + kx) < (unsigned)p.W)
                                    mask |= 1u << (ky * p.ks + kx);
                    }
                }
                pconst[j] = base + (unsigned)c * 16u;
                pmask[j] = mask;
            }
        };
#pragma unroll
        for (int j = 0; j < MAXP; ++j) {
            const int q = lw + j * NL;
            if (q >= A_INSTR && q < PIECES) {
                const int s = (q - A_INSTR) * 64 + lane;
                const int row = s / CPR, pc = s - row * CPR;
                const int c = pc ^ swz_l<BK>(row);
                pconst[j] = (unsigned)(((n0 + row) * p.Kpad + c * 8) * 2);
                pmask[j] = 0;
            }
        }
        int it_tile = j0, it_kt = 0, it_slot = 0;
        int is_tap = 0, is_ky = 0, is_kx = 0, is_kc = 0;
        set_tile(it_tile);
        auto issue_next = [&]() {
            const unsigned tapoff = (unsigned)(((is_ky * p.W + is_kx) * p.x_stride + is_kc) * 2);
            unsigned char* sbase = smem + it_slot * SB;
            const bool live = it_tile < mtiles;
            const bool src2 = p.x2_C > 0 && is_kc < p.x2_C;   // this k-step's channels come from the low-resolution source
#pragma unroll
            for (int j = 0; j < MAXP; ++j) {
                const int q = lw + j * NL;              // wave-uniform
                if (q < A_INSTR) {
                    const unsigned voff = ((pmask[j] >> is_tap) & 1u) ? ((src2 ? pconst2[j] : pconst[j]) + tapoff) : OOB;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(src2 ? xrs2 : xrs, (lds_void*)(sbase + q * 1024), 16, voff, 0, 0, 0);
                } else if (q < PIECES) {
                    const unsigned voff = live ? (pconst[j] + (unsigned)(it_kt * BK) * 2u) : OOB;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_void*)(sbase + BM * RB + (q - A_INSTR) * 1024), 16, voff, 0, 0, 0);
                }
            }
            it_slot = (it_slot + 1 == NS) ? 0 : it_slot + 1;
            is_kc += BK;
            if (is_kc >= p.Cin) {
                is_kc = 0;
                ++is_tap;
                if (++is_kx == p.ks) { is_kx = 0; ++is_ky; }
            }
            if (++it_kt == nk) {
                it_kt = 0; is_tap = 0; is_ky = 0; is_kx = 0; is_kc = 0;
                it_tile += G;
                set_tile(it_tile);
            }
        };
#pragma unroll
        for (int s = 0; s < NS - 1; ++s) issue_next();
        unsigned long long clk[3] = {0, 0, 0}, last = p.clk ? __builtin_amdgcn_s_memtime() : 0ull;
#define LC_STAMP(i) if (p.clk) { const unsigned long long now = __builtin_amdgcn_s_memtime(); clk[i] += now - last; last = now; }
        for (int tile = j0; tile < mtiles; tile += G) {
            for (int kt = 0; kt < nk; ++kt) {
                if (lw < FULL) wait_vml<(NS - 2) * MAXP>();   // this wave's pieces of stage g have landed
                else wait_vml<(NS - 2) * (MAXP - 1)>();
                LC_STAMP(0)
                __builtin_amdgcn_s_barrier();          // barrier(g): stage g is published, slot of stage g-1 is free
                LC_STAMP(1)
                issue_next();                          // stage g+NS-1
                LC_STAMP(2)
            }
        }
        wait_vml<0>();
        if (p.clk && lane == 0)
            for (int i = 0; i < 3; ++i) p.clk[((size_t)blockIdx.x * (NC + NL) + wave) * 3 + i] = clk[i];
        return;
    }

    // =================================================== consumer ===================================================
    const int wm = wave % WGM, wn = wave / WGM;
    const int fr = lane & 15, fc = lane >> 4;
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, (int)p.y_bytes, 0x00020000);
    float bias[FN][4];
#pragma unroll
    for (int a = 0; a < FN; ++a) {
        const int co = n0 + wn * WN + a * 16 + fc * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) bias[a][r] = (co + r < p.Cout) ? p.bias[co + r] : 0.f;
    }
    int aoff[KSUB], woff[KSUB];
#pragma unroll
    for (int ss = 0; ss < KSUB; ++ss) {
        const int ra = wm * WM + fr, rw = wn * WN + fr;
        aoff[ss] = ra * RB + (((ss * 4 + fc) ^ swz_l<BK>(ra)) * 16);
        woff[ss] = BM * RB + rw * RB + (((ss * 4 + fc) ^ swz_l<BK>(rw)) * 16);
    }
    f32x4 acc[FN][FM];
    int rslot = 0;
    unsigned long long clk[3] = {0, 0, 0}, last = p.clk ? __builtin_amdgcn_s_memtime() : 0ull;
    for (int tile = j0; tile < mtiles; tile += G) {
#pragma unroll
        for (int a = 0; a < FN; ++a)
#pragma unroll
            for (int b = 0; b < FM; ++b) acc[a][b] = f32x4{bias[a][0], bias[a][1], bias[a][2], bias[a][3]};
        for (int kt = 0; kt < nk; ++kt) {
            LC_STAMP(2)
            __builtin_amdgcn_s_barrier();              // barrier(g)
            LC_STAMP(0)
            const unsigned char* sb = smem + rslot * SB;
#pragma unroll
            for (int ss = 0; ss < KSUB; ++ss) {
                bf16x8 wf[FN], xf[FM];
#pragma unroll
                for (int a = 0; a < FN; ++a) wf[a] = *(const bf16x8*)(sb + woff[ss] + a * 16 * RB);
#pragma unroll
                for (int b = 0; b < FM; ++b) xf[b] = *(const bf16x8*)(sb + aoff[ss] + b * 16 * RB);
#pragma unroll
                for (int a = 0; a < FN; ++a)
#pragma unroll
                    for (int b = 0; b < FM; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[a], xf[b], acc[a][b], 0, 0, 0);
            }
            rslot = (rslot + 1 == NS) ? 0 : rslot + 1;
        }
        // ---- epilogue ----------------------------------------------------------------------------------------------
        const int m0 = tile * BM;
        uint2 rres[FM][FN];
        if (HAS_RES) {
#pragma unroll
            for (int b = 0; b < FM; ++b) {
                const int m = m0 + wm * WM + b * 16 + fr;
#pragma unroll
                for (int a = 0; a < FN; ++a) {
                    const int co = n0 + wn * WN + a * 16 + fc * 4;
                    rres[b][a] = (m < p.M && co < p.Cout)
                                     ? *(const uint2*)((const __bf16*)p.res + (size_t)m * p.res_stride + p.res_coff + co)
                                     : make_uint2(0u, 0u);
                }
            }
        }
#pragma unroll
        for (int b = 0; b < FM; ++b) {
            const int m = m0 + wm * WM + b * 16 + fr;
#pragma unroll
            for (int a = 0; a < FN; ++a) {
                const int co = n0 + wn * WN + a * 16 + fc * 4;
                const bool ok = (m < p.M) && (co < p.Cout);
                float v[4] = {acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]};
                if (p.act == ACT_SILU) silu4_packed(v);
                if (HAS_RES) {
                    const uint2 rr = rres[b][a];
                    v[0] += __uint_as_float(rr.x << 16); v[1] += __uint_as_float(rr.x & 0xffff0000u);
                    v[2] += __uint_as_float(rr.y << 16); v[3] += __uint_as_float(rr.y & 0xffff0000u);
                }
                if (OUT_F32) {
                    const unsigned off = ok ? ((unsigned)m * (unsigned)p.y_stride + (unsigned)(p.y_coff + co)) * 4u : OOB;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, make_float4(v[0], v[1], v[2], v[3])), yrs, off, 0, 0);
                } else {
                    const unsigned off = ok ? ((unsigned)m * (unsigned)p.y_stride + (unsigned)(p.y_coff + co)) * 2u : OOB;
                    __attribute__((aligned(8))) __bf16 o[4] = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                    __builtin_amdgcn_raw_buffer_store_b64(*(const __attribute__((ext_vector_type(2))) unsigned*)o, yrs, off, 0, 0);
                }
            }
        }
    }
    if (p.clk && lane == 0)
        for (int i = 0; i < 3; ++i) p.clk[((size_t)blockIdx.x * (NC + NL) + wave) * 3 + i] = clk[i];
}

// ---------------------------------------------------------------------------------------------------------------
struct LcCfg { int BM, BN, NC, NL, BK, NS; const char* name; };
static const LcCfg kL[] = {
    {128, 128, 8, 8, 64, 4, "conv_dma_lc_kernel<128,128,4,2,8,64,4>"},   // 0: 16 waves
    {128, 128, 4, 8, 32, 4, "conv_dma_lc_kernel<128,128,2,2,8,32,4>"},   // 1: 12 waves, 32-deep k-steps (Cin % 64 != 0)
    {128, 64, 4, 8, 64, 4, "conv_dma_lc_kernel<128,64,2,2,8,64,4>"},     // 2: 12 waves
    {128, 64, 8, 8, 64, 4, "conv_dma_lc_kernel<128,64,4,2,8,64,4>"},     // 3
    {128, 128, 4, 8, 64, 4, "conv_dma_lc_kernel<128,128,2,2,8,64,4>"},   // 4: 12 waves, one consumer per SIMD
    {256, 64, 8, 8, 64, 3, "conv_dma_lc_kernel<256,64,4,2,8,64,3>"},     // 5
    {64, 64, 4, 4, 64, 4, "conv_dma_lc_kernel<64,64,2,2,4,64,4>"},       // 6: 8 waves, two workgroups per CU
    {128, 64, 4, 12, 64, 4, "conv_dma_lc_kernel<128,64,2,2,12,64,4>"},   // 7: 16 waves, 24 pieces over 12 loaders
    {64, 128, 4, 12, 64, 4, "conv_dma_lc_kernel<64,128,2,2,12,64,4>"},   // 8
};
constexpr int kNumL = (int)(sizeof(kL) / sizeof(kL[0]));
int conv_dma_lc_num_cfgs() { return kNumL; }
const char* conv_dma_lc_kernel_name(int c) { return kL[c].name; }

bool conv_dma_lc_cfg_valid(const ConvParams& p, int c) {
    if (c < 0 || c >= kNumL) return false;
    if ((p.Cin % 32) != 0 || (p.Kpad % 32) != 0 || p.ks > 3 || p.up != 1) return false;
    if (p.x_bytes >= (1ull << 31) || p.w_bytes >= (1ull << 31) || p.y_bytes >= (1ull << 31)) return false;
    if ((p.Cout & 3) || (p.y_stride & 3) || (p.y_coff & 3) || (p.res && ((p.res_stride & 3) || (p.res_coff & 3)))) return false;
    if (p.res && p.out_f32) return false;
    const LcCfg& k = kL[c];
    if (k.BK == 64 && ((p.Cin % 64) != 0 || (p.Kpad % 64) != 0)) return false;
    if (p.x2_C > 0 && (p.ks != 1 || (p.x2_C % k.BK) != 0 || p.x2_bytes >= (1ull << 31))) return false;
    const int cpad = (p.Cout + 31) / 32 * 32;
    if (k.BN >= 2 * cpad) return false;
    // the weight rows of the block must exist (packed weights are padded to 128 rows)
    return true;
}

template <int BM, int BN, int WGM, int WGN, int NL, int BK, int NS, bool HAS_RES, bool OUT_F32>
static hipError_t launch_lc_var(const ConvParams& p, hipStream_t st) {
    const int mtiles = (p.M + BM - 1) / BM, ntiles = (p.Cout + BN - 1) / BN;
    const size_t sh = (size_t)NS * (BM + BN) * BK * 2;
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>((WGM * WGN + NL) >= 12 ? 1 : 2, (160 * 1024) / sh));
    int G = (256 * per_cu) / ntiles;
    if (G < 1) G = 1;
    if (G > mtiles) G = mtiles;
    auto kern = conv_dma_lc_kernel<BM, BN, WGM, WGN, NL, BK, NS, HAS_RES, OUT_F32>;
    static bool attr = false;
    if (!attr && sh > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        if (e != hipSuccess) return e;
        attr = true;
    }
    static const bool clocks = [] { const char* v = std::getenv("YOLOP_LC_CLOCKS"); return v && *v == '1'; }();   // debug: per-phase s_memtime sums
    if (clocks) {
        constexpr int NWV = WGM * WGN + NL;
        ConvParams q = p;
        const size_t n = (size_t)G * ntiles * NWV * 3;
        if (hipMalloc((void**)&q.clk, n * 8) != hipSuccess) return hipErrorOutOfMemory;
        (void)hipMemset(q.clk, 0, n * 8);
        hipLaunchKernelGGL(kern, dim3(G * ntiles), dim3(NWV * 64), sh, st, q, mtiles, ntiles, G);
        (void)hipStreamSynchronize(st);
        std::vector<unsigned long long> h(n);
        (void)hipMemcpy(h.data(), q.clk, n * 8, hipMemcpyDeviceToHost);
        (void)hipFree(q.clk);
        const double steps = (double)mtiles / G * (p.Kpad / BK);
        auto avg = [&](int w, int i) { double s = 0; for (int g = 0; g < G * ntiles; ++g) s += (double)h[((size_t)g * NWV + w) * 3 + i]; return s / (G * ntiles) / steps; };
        fprintf(stderr, "[lc clocks] %dx%d tile, M=%d K=%d Cout=%d, %d k-steps/tile, %.2f tiles per workgroup; ticks per k-step: consumer 0: barrier wait %.0f, reads+MFMA(+epilogue) %.0f"
                        " | loader 0: dma wait %.0f, barrier wait %.0f, issue %.0f\n",
                BM, BN, p.M, p.Kpad, p.Cout, p.Kpad / BK, (double)mtiles / G, avg(0, 0), avg(0, 2), avg(WGM * WGN, 0), avg(WGM * WGN, 1), avg(WGM * WGN, 2));
        return hipGetLastError();
    }
    hipLaunchKernelGGL(kern, dim3(G * ntiles), dim3((WGM * WGN + NL) * 64), sh, st, p, mtiles, ntiles, G);
    return hipGetLastError();
}
template <int BM, int BN, int WGM, int WGN, int NL, int BK, int NS>
static hipError_t launch_lc_one(const ConvParams& p, hipStream_t st) {
    if (p.out_f32) return launch_lc_var<BM, BN, WGM, WGN, NL, BK, NS, false, true>(p, st);
    if (p.res) return launch_lc_var<BM, BN, WGM, WGN, NL, BK, NS, true, false>(p, st);
    return launch_lc_var<BM, BN, WGM, WGN, NL, BK, NS, false, false>(p, st);
}

hipError_t launch_conv_dma_lc(const ConvParams& p, int c, hipStream_t st) {
    switch (c) {
        case 0: return launch_lc_one<128, 128, 4, 2, 8, 64, 4>(p, st);
        case 1: return launch_lc_one<128, 128, 2, 2, 8, 32, 4>(p, st);
        case 2: return launch_lc_one<128, 64, 2, 2, 8, 64, 4>(p, st);
        case 3: return launch_lc_one<128, 64, 4, 2, 8, 64, 4>(p, st);
        case 4: return launch_lc_one<128, 128, 2, 2, 8, 64, 4>(p, st);
        case 5: return launch_lc_one<256, 64, 4, 2, 8, 64, 3>(p, st);
        case 6: return launch_lc_one<64, 64, 2, 2, 4, 64, 4>(p, st);
        case 7: return launch_lc_one<128, 64, 2, 2, 12, 64, 4>(p, st);
        default: return launch_lc_one<64, 128, 2, 2, 12, 64, 4>(p, st);
    }
}

}  // namespace yp
