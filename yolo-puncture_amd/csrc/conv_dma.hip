// bf16 dense convolution, implicit GEMM on MFMA 16x16x32 with an LDS-DMA operand pipeline (gfx950).
//
// Same math and fragment orientation as conv_igemm.hip (weights = A operand, pixels = B operand, a lane ends
// up with 4 consecutive output channels of one pixel), but the operand tiles go global -> LDS directly with
// `buffer_load_dwordx4 ... lds` (no VGPR staging): each wave instruction deposits 64 x 16 B linearly in LDS while
// every lane supplies its own source address, so the im2col gather, the zero padding (out-of-range buffer
// offsets return 0) and the bank-conflict swizzle (applied to the SOURCE chunk, guide rule 21) are all folded
// into the address of the load. NS ring slots keep NS-1 k-steps of loads in flight across raw s_barriers with a
// counted s_waitcnt vmcnt (never 0 in the loop); loads for steps past the end are issued out of range so that
// the count stays uniform.
//
// Replaces the `Conv` / 1x1 layers of the ultralytics graph inside `.predict` (reference yolo_seg/app.py:91),
// blocks per SURVEY.md Appendix A.2 [U]. Used when Cin % 32 == 0 in bf16 mode; everything else (fp32 mode,
// Cin = 16 / 80) runs conv_igemm.hip.
#include "common.h"
#include <cstdlib>

namespace yp {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ float silu2(float x) { return x / (1.0f + __expf(-x)); }

// s_waitcnt vmcnt(N) only (gfx9 encoding: vmcnt[3:0] | expcnt[6:4]=7 | lgkmcnt[11:8]=15 | vmcnt_hi[15:14])
template <int N> __device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt range");
    __builtin_amdgcn_s_waitcnt((N & 0xF) | (7 << 4) | (0xF << 8) | (((N >> 4) & 3) << 14));
}

// XOR mask applied to the 16-B chunk index within a row. 64-B rows (BK=32): conflict-free for ds_read_b128 of ANY
// 16 consecutive rows; 128-B rows (BK=64): conflict-free for 16-row aligned fragments. Derivations: DESIGN.md.
template <int BK> __device__ __forceinline__ int swz_mask(int row) {
    return BK == 32 ? (((row >> 2) & 1) << 1) : ((row >> 1) & 7);
}

template <int BM, int BN, int WGM, int WGN, int BK, int NS>
__global__ __launch_bounds__(WGM * WGN * 64) void conv_dma_kernel(const ConvParams p, const int mtiles, const int ntiles) {
    constexpr int NW = WGM * WGN;             // waves per workgroup (4 or 8)
    constexpr int CPR = BK / 8;               // 16-B chunks per LDS row
    constexpr int RB = BK * 2;                // bytes per LDS row
    constexpr int A_INSTR = BM * CPR / 64;    // 1-KiB wave-instructions per stage for the pixel tile
    constexpr int W_INSTR = BN * CPR / 64;
    constexpr int A_IPW = A_INSTR / NW;
    constexpr int W_IPW = (W_INSTR + NW - 1) / NW;
    constexpr int LPW = A_IPW + W_IPW;        // loads per wave per stage (uniform)
    constexpr int SB = (BM + BN) * RB;        // stage bytes
    constexpr int WM = BM / WGM, WN = BN / WGN, FM = WM / 16, FN = WN / 16;
    constexpr int KSUB = BK / 32;
    static_assert((NW == 4 || NW == 8) && A_INSTR % NW == 0 && A_IPW >= 1, "tile/wave layout");
    constexpr unsigned OOB = 0x80000000u;

    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];   // NS stages + 1 KiB dump slot

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WGM, wn = wave / WGM;

    // XCD-aware bijective remap of the linear block id (blocks b, b+8, ... share an XCD / L2): each XCD walks a
    // contiguous run of tiles, n-tile fastest so the CTAs that share a pixel tile are neighbours.
    int bid = blockIdx.x;
    {
        const int nwg = mtiles * ntiles, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, j = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int mt = bid / ntiles, nt = bid - mt * ntiles;
    const int m0 = mt * BM, n0 = nt * BN;
    const int HoWo = p.Ho * p.Wo;

    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)p.w_bytes, 0x00020000);

    // ---- per-lane constants of the A (pixel) loads: row, byte offset of its (hi0,wi0) corner, tap validity ----
    unsigned aconst[A_IPW], amask[A_IPW];
#pragma unroll
    for (int j = 0; j < A_IPW; ++j) {
        const int s = (wave * A_IPW + j) * 64 + lane;
        const int row = s / CPR, pc = s - row * CPR;
        const int c = pc ^ swz_mask<BK>(row);
        const int m = m0 + row;
        unsigned mask = 0, base = 0;
        if (m < p.M) {
            const int b = m / HoWo, r = m - b * HoWo;
            const int ho = r / p.Wo, wo = r - ho * p.Wo;
            const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
            base = (unsigned)(((b * p.H + hi0) * p.W + wi0) * p.x_stride + p.x_coff) * 2u;
            for (int ky = 0; ky < p.ks; ++ky)
                for (int kx = 0; kx < p.ks; ++kx)
                    if ((unsigned)(hi0 + ky) < (unsigned)p.H && (unsigned)(wi0 + kx) < (unsigned)p.W)
                        mask |= 1u << (ky * p.ks + kx);
        }
        aconst[j] = base + (unsigned)c * 16u;
        amask[j] = mask;
    }
    unsigned wconst[W_IPW];
#pragma unroll
    for (int j = 0; j < W_IPW; ++j) {
        const int ii = wave * W_IPW + j;
        const int s = ii * 64 + lane;
        const int row = s / CPR, pc = s - row * CPR;
        const int c = pc ^ swz_mask<BK>(row);
        wconst[j] = (ii < W_INSTR) ? (unsigned)(((n0 + row) * p.Kpad + c * 8) * 2) : OOB;
    }

    // ---- scalar k-step state for the NEXT stage to issue -----------------------------------------------------
    int is_tap = 0, is_ky = 0, is_kx = 0, is_kc = 0, is_k0 = 0;
    auto issue = [&](int slot) {
        const unsigned tapoff = (unsigned)(((is_ky * p.W + is_kx) * p.x_stride + is_kc) * 2);
        unsigned char* sbase = smem + slot * SB;
#pragma unroll
        for (int j = 0; j < A_IPW; ++j) {
            const unsigned voff = ((amask[j] >> is_tap) & 1u) ? (aconst[j] + tapoff) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_void*)(sbase + (wave * A_IPW + j) * 1024), 16, voff, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < W_IPW; ++j) {
            const int ii = wave * W_IPW + j;
            unsigned char* dst = (ii < W_INSTR) ? (sbase + BM * RB + ii * 1024) : (smem + NS * SB);
            const unsigned voff = (wconst[j] == OOB || is_k0 >= p.Kpad) ? OOB : (wconst[j] + (unsigned)is_k0 * 2u);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_void*)dst, 16, voff, 0, 0, 0);
        }
        // advance by BK k
        is_k0 += BK;
        is_kc += BK;
        if (is_kc >= p.Cin) {
            is_kc = 0;
            ++is_tap;
            if (++is_kx == p.ks) { is_kx = 0; ++is_ky; }
        }
        if (is_tap > 30) is_tap = 30;   // dummy stages past the end stay out of range (mask has <= 9 bits)
    };

    // ---- accumulators, bias ------------------------------------------------------------------------------------
    f32x4 acc[FN][FM];
#pragma unroll
    for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int b = 0; b < FM; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment read offsets (row part + swizzled chunk), constant per lane
    const int fr = lane & 15, fc = lane >> 4;
    int aoff[KSUB], woff[KSUB];
#pragma unroll
    for (int ss = 0; ss < KSUB; ++ss) {
        const int ra = wm * WM + fr, rw = wn * WN + fr;
        aoff[ss] = ra * RB + (((ss * 4 + fc) ^ swz_mask<BK>(ra)) * 16);
        woff[ss] = BM * RB + rw * RB + (((ss * 4 + fc) ^ swz_mask<BK>(rw)) * 16);
    }

    const int nk = p.Kpad / BK;
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) issue(s);

    int rslot = 0, wslot = NS - 1;
    for (int kt = 0; kt < nk; ++kt) {
        wait_vmcnt<(NS - 2) * LPW>();
        __builtin_amdgcn_s_barrier();
        issue(wslot);
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
        wslot = (wslot + 1 == NS) ? 0 : wslot + 1;
    }
    wait_vmcnt<0>();   // retire the out-of-range tail loads before the LDS is released

    // ---- epilogue: +bias, SiLU, +residual, store 4 consecutive couts of one pixel per lane --------------------------
    const bool vec_ok = ((p.Cout & 3) == 0) && ((p.y_stride & 3) == 0) && ((p.y_coff & 3) == 0) &&
                        (p.res == nullptr || (((p.res_stride & 3) == 0) && ((p.res_coff & 3) == 0)));
    float bias[FN][4];
#pragma unroll
    for (int a = 0; a < FN; ++a) {
        const int co = n0 + wn * WN + a * 16 + fc * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) bias[a][r] = (co + r < p.Cout) ? p.bias[co + r] : 0.f;
    }
#pragma unroll
    for (int b = 0; b < FM; ++b) {
        const int m = m0 + wm * WM + b * 16 + fr;
        if (m >= p.M) continue;
        size_t opix = (size_t)m;
        if (p.up != 1) {
            const int bb = m / HoWo, r = m - bb * HoWo;
            const int ho = r / p.Wo, wo = r - ho * p.Wo;
            opix = ((size_t)bb * (p.Ho * p.up) + ho * p.up + p.oy) * (size_t)(p.Wo * p.up) + wo * p.up + p.ox;
        }
#pragma unroll
        for (int a = 0; a < FN; ++a) {
            const int co = n0 + wn * WN + a * 16 + fc * 4;
            if (co >= p.Cout) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float t = acc[a][b][r] + bias[a][r];
                if (p.act == ACT_SILU) t = silu2(t);
                v[r] = t;
            }
            if (p.res) {
                const __bf16* rp = (const __bf16*)p.res + (size_t)m * p.res_stride + p.res_coff + co;
                if (vec_ok) {
                    const uint2 rr = *(const uint2*)rp;
                    v[0] += __uint_as_float(rr.x << 16); v[1] += __uint_as_float(rr.x & 0xffff0000u);
                    v[2] += __uint_as_float(rr.y << 16); v[3] += __uint_as_float(rr.y & 0xffff0000u);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (co + r < p.Cout) v[r] += (float)rp[r];
                }
            }
            if (p.out_f32) {
                float* yp = (float*)p.y + opix * p.y_stride + p.y_coff + co;
                if (vec_ok) *(float4*)yp = make_float4(v[0], v[1], v[2], v[3]);
                else
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (co + r < p.Cout) yp[r] = v[r];
            } else {
                __bf16* yp = (__bf16*)p.y + opix * p.y_stride + p.y_coff + co;
                if (vec_ok) {
                    __attribute__((aligned(8))) __bf16 o[4] = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                    *(uint2*)yp = *(const uint2*)o;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (co + r < p.Cout) yp[r] = (__bf16)v[r];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// configurations + launch. p.cfg >= 0 selects one explicitly (the engine's plan-time autotuner times the valid ones
// per layer and keeps the fastest); p.cfg < 0 uses the static heuristic.
// ---------------------------------------------------------------------------------------------------------------
struct DmaCfg { int BM, BN, NW, BK, NS; const char* name; };
static const DmaCfg kCfgs[] = {
    {128, 32, 4, 32, 4, "conv_dma_kernel<128,32,4,1,32,4>"},     // 0
    {128, 64, 4, 32, 4, "conv_dma_kernel<128,64,2,2,32,4>"},     // 1
    {128, 128, 4, 32, 4, "conv_dma_kernel<128,128,2,2,32,4>"},   // 2
    {64, 64, 4, 32, 4, "conv_dma_kernel<64,64,2,2,32,4>"},       // 3
    {256, 64, 8, 32, 4, "conv_dma_kernel<256,64,4,2,32,4>"},     // 4
    {256, 128, 8, 32, 4, "conv_dma_kernel<256,128,4,2,32,4>"},   // 5
    {256, 128, 8, 64, 3, "conv_dma_kernel<256,128,4,2,64,3>"},   // 6
    {128, 128, 4, 64, 3, "conv_dma_kernel<128,128,2,2,64,3>"},   // 7
    {128, 64, 4, 64, 3, "conv_dma_kernel<128,64,2,2,64,3>"},     // 8
    {256, 64, 8, 64, 3, "conv_dma_kernel<256,64,4,2,64,3>"},     // 9
    {128, 256, 8, 32, 4, "conv_dma_kernel<128,256,2,4,32,4>"},   // 10
    {256, 32, 8, 32, 4, "conv_dma_kernel<256,32,8,1,32,4>"},     // 11
    {128, 256, 8, 64, 3, "conv_dma_kernel<128,256,2,4,64,3>"},   // 12
    {64, 128, 4, 64, 3, "conv_dma_kernel<64,128,2,2,64,3>"},     // 13
};
constexpr int kNumCfgs = (int)(sizeof(kCfgs) / sizeof(kCfgs[0]));

int conv_dma_num_cfgs() { return kNumCfgs; }

bool conv_dma_supported(const ConvParams& p) {
    return (p.Cin % 32) == 0 && (p.Kpad % 32) == 0 && p.x_bytes < (1ull << 31) && p.w_bytes < (1ull << 31) && p.ks <= 3;
}

bool conv_dma_cfg_valid(const ConvParams& p, int c) {
    if (c < 0 || c >= kNumCfgs || !conv_dma_supported(p)) return false;
    const DmaCfg& k = kCfgs[c];
    if (k.BK == 64 && ((p.Cin % 64) != 0 || (p.Kpad % 64) != 0)) return false;
    const int cpad = (p.Cout + 31) / 32 * 32;
    if (k.BN > 32 && k.BN >= 2 * cpad) return false;            // more than half the tile would be padding
    if (k.BN == 32 && p.Cout > 32) return false;
    return true;
}

static int dma_heuristic(const ConvParams& p) {
    if (p.Cout <= 32) return 0;
    const long ctas128 = (long)((p.M + 127) / 128) * ((p.Cout + 127) / 128);
    if ((p.Cout % 128) == 0 && ctas128 >= 1024) return 2;
    const long ctas64 = (long)((p.M + 127) / 128) * ((p.Cout + 63) / 64);
    if (ctas64 >= 512) return 1;
    return 3;
}
static int g_force_cfg = -1;   // debug/test override (yp_debug_force_conv_cfg)
void conv_dma_force_cfg(int c) { g_force_cfg = c; }
int conv_dma_forced_cfg() { return g_force_cfg; }
static int g_dbg_ablate = 0;
void conv_set_debug_ablation(int v) { g_dbg_ablate = v; }
int conv_debug_ablation() { return g_dbg_ablate; }
bool tile_balance_enabled(int family) { static const int mask = [] { const char* v = std::getenv("YOLOP_BALANCE"); return v ? atoi(v) : 6; }(); return (mask & family) != 0; }
static int dma_choice(const ConvParams& p) {
    if (conv_dma_cfg_valid(p, g_force_cfg)) return g_force_cfg;
    return conv_dma_cfg_valid(p, p.cfg) ? p.cfg : dma_heuristic(p);
}

const char* conv_dma_kernel_name(const ConvParams& p) { return kCfgs[dma_choice(p)].name; }

template <int BM, int BN, int WGM, int WGN, int BK, int NS>
static hipError_t launch_one(const ConvParams& p, hipStream_t st) {
    const int mtiles = (p.M + BM - 1) / BM, ntiles = (p.Cout + BN - 1) / BN;
    const size_t sh = (size_t)NS * (BM + BN) * BK * 2 + 1024;
    auto kern = conv_dma_kernel<BM, BN, WGM, WGN, BK, NS>;
    static bool attr = false;
    if (!attr && sh > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        if (e != hipSuccess) return e;
        attr = true;
    }
    hipLaunchKernelGGL(kern, dim3(mtiles * ntiles), dim3(WGM * WGN * 64), sh, st, p, mtiles, ntiles);
    return hipGetLastError();
}

hipError_t launch_conv_dma(const ConvParams& p, hipStream_t st) {
    switch (dma_choice(p)) {
        case 0: return launch_one<128, 32, 4, 1, 32, 4>(p, st);
        case 1: return launch_one<128, 64, 2, 2, 32, 4>(p, st);
        case 2: return launch_one<128, 128, 2, 2, 32, 4>(p, st);
        case 3: return launch_one<64, 64, 2, 2, 32, 4>(p, st);
        case 4: return launch_one<256, 64, 4, 2, 32, 4>(p, st);
        case 5: return launch_one<256, 128, 4, 2, 32, 4>(p, st);
        case 6: return launch_one<256, 128, 4, 2, 64, 3>(p, st);
        case 7: return launch_one<128, 128, 2, 2, 64, 3>(p, st);
        case 8: return launch_one<128, 64, 2, 2, 64, 3>(p, st);
        case 9: return launch_one<256, 64, 4, 2, 64, 3>(p, st);
        case 10: return launch_one<128, 256, 2, 4, 32, 4>(p, st);
        case 11: return launch_one<256, 32, 8, 1, 32, 4>(p, st);
        case 12: return launch_one<128, 256, 2, 4, 64, 3>(p, st);
        default: return launch_one<64, 128, 2, 2, 64, 3>(p, st);
    }
}

}  // namespace yp
