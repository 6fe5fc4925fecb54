// "Pixels direct" form of the bf16 1x1 convolution (configuration ids 800+).
//
// Why: the LDS-DMA GEMM kernels pull BOTH operands of a 128x128x64 k-step (32 KB) through the DMA path
// of a CU, which delivers 17-43 B/clk depending on how many waves issue - 1000-1900 cycles for 512 cycles of matrix work - and the
// per-wave operand reads keep the LDS array half busy. A 1x1 convolution's pixel operand needs no staging at all: in NHWC a pixel
// row IS the k-contiguous B-fragment row of `v_mfma_f32_16x16x32_bf16` (lane (r, kq) wants 16 bytes of pixel r at channel 8*kq), and a
// wave that owns its pixels exclusively reads every activation byte exactly once. So here
//   * the PIXEL operand goes global -> registers in fragment layout (one buffer_load_b128 per lane, fragment and 32-deep k-substep),
//     NS-1 k-steps ahead, never through LDS;
//   * only the WEIGHT block [BN][64] of the k-step travels by LDS-DMA into an NS-slot ring shared by the 8 waves (16-32 KB per k-step
//     instead of 32-48), and a wave computes PXW x FN accumulator tiles from PXW pixel fragments + FN weight fragments per substep:
//     256 px x 256 couts per workgroup = the FLOP-per-delivered-byte of a 256x256 tile with half of the bytes bypassing the LDS.
// One tile per workgroup (no stores inside the k-loop, so the vmcnt arithmetic is the ring's alone); the nearest-x2 upsample fold
// (channels below x2_C come from the low-resolution tensor at (ho >> 1, wo >> 1)) is a per-k-step choice of the lane's row base.
#include "common.h"

namespace yp {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void lds_void;

template <int N> __device__ __forceinline__ void px_wait_vmc() {
    static_assert(N >= 0 && N < 64, "vmcnt range");
    __builtin_amdgcn_s_waitcnt((N & 0xF) | (7 << 4) | (0xF << 8) | (((N >> 4) & 3) << 14));
}

template <int PXW, int FN, int WGM, int WGN, bool HAS_RES, bool OUT_F32>
__global__ __launch_bounds__(WGM * WGN * 64) void conv_pxd_kernel(const ConvParams p, const int mtiles, const int BMe) {
    constexpr int NW = WGM * WGN;
    constexpr int BM = WGM * PXW * 16, BN = WGN * FN * 16;
    constexpr int BK = 64, NS = 3;
    constexpr int W_INSTR = BN * 8 / 64;               // 1-KiB weight pieces per k-step (8 chunks of 16 B per 128-B row)
    constexpr int LW = (W_INSTR + NW - 1) / NW;        // pieces per wave
    constexpr int SLOT = W_INSTR * 1024;
    constexpr int PL = PXW * 2;                        // pixel loads per wave and k-step
    constexpr unsigned OOB = 0x80000000u;
    static_assert(LW + 2 * PL < 64, "vmcnt immediate");

    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char* const dump = smem + NS * SLOT;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WGM, wn = wave / WGM;
    const int fr = lane & 15, fc = lane >> 4;

    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, j = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int mt = bid % mtiles, nt = bid / mtiles;
    // BMe <= BM rows of a tile are in use: the host sizes the tiles so that the launch is whole rounds of workgroups (a 40x40 map at batch 32
    // is 200 tiles of 256 pixels on 256 CUs; 256 tiles of 200 pixels keep every CU busy and finish a fifth earlier)
    const int m0 = mt * BMe, n0 = nt * BN;
    const int nk = (p.Kpad + BK - 1) / BK;            // Kpad % 32 == 0: a trailing half step reads zero pixels (its weight bytes are then irrelevant)

    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, (int)p.y_bytes, 0x00020000);

    // this lane's pixel rows: byte offset of (pixel, channel 8*fc) in x, and in the low-resolution source of a folded upsample
    unsigned xoff[PXW], xoff2[PXW];
#pragma unroll
    for (int f = 0; f < PXW; ++f) {
        const int ri = (wm * PXW + f) * 16 + fr;
        const int m = m0 + ri;
        xoff[f] = OOB; xoff2[f] = OOB;
        if (ri < BMe && m < p.M) {
            xoff[f] = (unsigned)(m * p.x_stride + p.x_coff) * 2u + (unsigned)fc * 16u;
            if (p.x2_C > 0) {
                const int HoWo = p.Ho * p.Wo;
                const int b = m / HoWo, r = m - b * HoWo;
                const int ho = r / p.Wo, wo = r - ho * p.Wo;
                xoff2[f] = (unsigned)(((b * p.x2_H + (ho >> 1)) * p.x2_W + (wo >> 1)) * p.x2_stride + p.x2_coff) * 2u + (unsigned)fc * 16u;
            }
        }
    }
    // this wave's weight pieces: piece ii covers rows 8*ii .. 8*ii+7 of the [BN][64] block; lane -> (row, 16-B chunk), chunk swizzled
    unsigned wconst[LW];
#pragma unroll
    for (int j = 0; j < LW; ++j) {
        const int ii = wave * LW + j;
        const int s = ii * 64 + lane;
        const int row = s >> 3, pc = s & 7;
        const int c = pc ^ ((row >> 1) & 7);
        wconst[j] = (ii < W_INSTR) ? (unsigned)(((n0 + row) * p.Kpad + c * 8) * 2) : OOB;
    }
    auto issue_w = [&](int kt, int slot) {
        unsigned char* dst = smem + slot * SLOT;
        const bool live = kt < nk;
#pragma unroll
        for (int j = 0; j < LW; ++j) {
            const int ii = wave * LW + j;
            const unsigned voff = (live && wconst[j] != OOB) ? wconst[j] + (unsigned)(kt * BK) * 2u : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_void*)((ii < W_INSTR) ? dst + ii * 1024 : dump), 16, voff, 0, 0, 0);
        }
    };
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    // The pixel loads are inline asm on purpose: the compiler's own vmcnt bookkeeping cannot see the LDS-DMA pieces and, across the
    // back edge of the unrolled loop, falls back to waiting for (almost) everything in flight before the first use of a register set
    // (vmcnt(4) instead of vmcnt(12) every third step). With the loads opaque, the counted wait below - which passes the set's
    // registers through as operands, so no use can be scheduled above it - is the only one.
    auto make_rs = [](const void* ptr, size_t bytes) -> u32x4 {      // raw buffer descriptor: base, stride 0, num_records, flags (as make_buffer_rsrc)
        const unsigned long long a = (unsigned long long)ptr;
        u32x4 r;
        r[0] = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a);
        r[1] = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((a >> 32) & 0xffffu));
        r[2] = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)bytes);
        r[3] = 0x00020000u;
        return r;
    };
    const u32x4 xrs_v = make_rs(p.x, p.x_bytes), xrs2_v = make_rs(p.x2_C > 0 ? p.x2 : p.x, p.x2_C > 0 ? p.x2_bytes : p.x_bytes);
    auto load_px = [&](int kt, u32x4 (&dst)[PXW][2]) {
        const bool live = kt < nk;
        const int k0 = kt * BK;
        const bool src2 = p.x2_C > 0 && k0 < p.x2_C;
        const u32x4 rs = src2 ? xrs2_v : xrs_v;           // (uniform choice: scalar registers)
#pragma unroll
        for (int f = 0; f < PXW; ++f)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const unsigned base = src2 ? xoff2[f] : xoff[f];
                const unsigned voff = (live && base != OOB && k0 + s * 32 < p.Kpad) ? base + (unsigned)(k0 + s * 32) * 2u : OOB;
                asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(dst[f][s]) : "v"(voff), "s"(rs) : "memory");
            }
    };
    auto wait_set = [&](u32x4 (&cur)[PXW][2]) {
        if constexpr (PXW == 2)
            asm volatile("s_waitcnt vmcnt(%4)" : "+v"(cur[0][0]), "+v"(cur[0][1]), "+v"(cur[1][0]), "+v"(cur[1][1]) : "n"(LW + 2 * PL) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(%2)" : "+v"(cur[0][0]), "+v"(cur[0][1]) : "n"(LW + 2 * PL) : "memory");
    };

    f32x4 acc[FN][PXW];
    {
#pragma unroll
        for (int a = 0; a < FN; ++a) {
            const int co = n0 + wn * (FN * 16) + a * 16 + fc * 4;
            f32x4 b4;
#pragma unroll
            for (int r = 0; r < 4; ++r) b4[r] = (co + r < p.Cout) ? p.bias[co + r] : 0.f;
#pragma unroll
            for (int f = 0; f < PXW; ++f) acc[a][f] = b4;          // bias rides in the accumulator
        }
    }

    // ---- prologue: weight steps 0 and 1, pixel steps 0, 1 and 2 in flight ----------------------------------------------------------
    // Issue order per wave:  w0 px0 w1 px1 px2 | w2 px3 | w3 px4 | ...   step g = { wait ; barrier ; w(g+2) ; compute(g) ; px(g+3) }.
    // What step g needs (w(g), px(g)) is older than exactly LW + 2*PL later operations at its wait, every step.
    u32x4 pxa[PXW][2], pxb[PXW][2], pxc[PXW][2];
    issue_w(0, 0); load_px(0, pxa);
    issue_w(1, 1); load_px(1, pxb);
    load_px(2, pxc);

    // three NAMED register sets rotate (the loop is unrolled by three): a set is refilled by the loads of step g+3 right after the
    // MFMAs of step g have read it - no register copies, which would make the compiler wait for the loads in flight
    auto step = [&](int g, u32x4 (&cur)[PXW][2]) {
        wait_set(cur);
        __builtin_amdgcn_s_barrier();
        issue_w(g + 2, (g + 2) % NS);                     // overwrites the slot of step g-1, whose reads were consumed before this barrier
        const unsigned char* ws = smem + (g % NS) * SLOT;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            constexpr int AH = FN > 8 ? 8 : FN;           // weight fragments held at a time
#pragma unroll
            for (int a0 = 0; a0 < FN; a0 += AH) {
                bf16x8 wf[AH];
#pragma unroll
                for (int a = 0; a < AH; ++a) {
                    const int row = wn * (FN * 16) + (a0 + a) * 16 + fr;
                    wf[a] = *(const bf16x8*)(ws + row * 128 + (((s * 4 + fc) ^ ((row >> 1) & 7)) * 16));
                }
#pragma unroll
                for (int f = 0; f < PXW; ++f) {
                    const bf16x8 xf = __builtin_bit_cast(bf16x8, cur[f][s]);
#pragma unroll
                    for (int a = 0; a < AH; ++a) acc[a0 + a][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[a], xf, acc[a0 + a][f], 0, 0, 0);
                }
            }
        }
        load_px(g + 3, cur);
    };
    for (int g = 0; g < nk; g += 3) {
        step(g, pxa);
        if (g + 1 < nk) step(g + 1, pxb);
        if (g + 2 < nk) step(g + 2, pxc);
    }

    // the loads issued for steps past the end (out-of-range offsets: they return zeros) still write their registers when they land:
    // drain them while the three sets are formally alive, or the compiler could hand those registers to the epilogue
    if constexpr (PXW == 2)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(pxa[0][0]), "+v"(pxa[0][1]), "+v"(pxa[1][0]), "+v"(pxa[1][1]), "+v"(pxb[0][0]), "+v"(pxb[0][1]), "+v"(pxb[1][0]),
                     "+v"(pxb[1][1]), "+v"(pxc[0][0]), "+v"(pxc[0][1]), "+v"(pxc[1][0]), "+v"(pxc[1][1]) : : "memory");
    else
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(pxa[0][0]), "+v"(pxa[0][1]), "+v"(pxb[0][0]), "+v"(pxb[0][1]), "+v"(pxc[0][0]), "+v"(pxc[0][1]) : : "memory");

    // ---- epilogue --------------------------------------------------------------------------------------------------------------
#pragma unroll
    for (int f = 0; f < PXW; ++f) {
        const int ri = (wm * PXW + f) * 16 + fr;
        const int m = m0 + ri;
        const bool pix_ok = ri < BMe && m < p.M;
#pragma unroll
        for (int a = 0; a < FN; ++a) {
            const int co = n0 + wn * (FN * 16) + a * 16 + fc * 4;
            const bool ok = pix_ok && (co < p.Cout);
            float v[4] = {acc[a][f][0], acc[a][f][1], acc[a][f][2], acc[a][f][3]};
            if (p.act == ACT_SILU) silu4_packed(v);
            if (HAS_RES && ok) {
                const uint2 rr = *(const uint2*)((const __bf16*)p.res + (size_t)m * p.res_stride + p.res_coff + co);
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

// ---------------------------------------------------------------------------------------------------------------
struct PxdCfg { int PXW, FN, WGM, WGN; const char* name; };
static const PxdCfg kPxd[] = {
    {2, 16, 8, 1, "conv_pxd_kernel<2,16,8,1>"},   // 0: 256 px x 256 couts
    {2, 8, 8, 1, "conv_pxd_kernel<2,8,8,1>"},     // 1: 256 px x 128 couts
    {2, 8, 4, 2, "conv_pxd_kernel<2,8,4,2>"},     // 2: 128 px x 256 couts
    {1, 8, 8, 1, "conv_pxd_kernel<1,8,8,1>"},     // 3: 128 px x 128 couts
    {1, 8, 4, 2, "conv_pxd_kernel<1,8,4,2>"},     // 4:  64 px x 256 couts
    {2, 4, 8, 1, "conv_pxd_kernel<2,4,8,1>"},     // 5: 256 px x  64 couts
    {1, 16, 8, 1, "conv_pxd_kernel<1,16,8,1>"},   // 6: 128 px x 256 couts, one pixel fragment per wave
    {1, 4, 8, 1, "conv_pxd_kernel<1,4,8,1>"},     // 7: 128 px x  64 couts
};
constexpr int kNumPxd = (int)(sizeof(kPxd) / sizeof(kPxd[0]));

int conv_pxd_num_cfgs() { return kNumPxd; }

bool conv_pxd_cfg_valid(const ConvParams& p, int c) {
    if (c < 0 || c >= kNumPxd) return false;
    const PxdCfg& k = kPxd[c];
    if (p.ks != 1 || p.stride != 1 || p.up != 1 || p.w2 || (p.Cin % 32) != 0 || p.Kpad != p.Cin) return false;
    if ((p.x_stride & 7) || (p.x_coff & 7)) return false;                                  // 16-byte fragment loads
    if (p.x2_C > 0 && ((p.x2_C % 64) != 0 || (p.x2_stride & 7) || (p.x2_coff & 7))) return false;
    if (p.x_bytes >= (1ull << 31) || p.w_bytes >= (1ull << 31) || p.y_bytes >= (1ull << 31) || (p.x2_C > 0 && p.x2_bytes >= (1ull << 31))) return false;
    if ((p.Cout & 3) || (p.y_stride & 3) || (p.y_coff & 3) || (p.res && ((p.res_stride & 3) || (p.res_coff & 3)))) return false;
    if (p.res && p.out_f32) return false;
    const int BN = k.WGN * k.FN * 16;
    const int cpad = (p.Cout + 63) / 64 * 64;
    if (BN > 64 && BN >= 2 * cpad) return false;                                             // more than half the tile would be padding
    const int ntiles = (p.Cout + BN - 1) / BN;
    if ((p.Cout + 127) / 128 * 128 < ntiles * BN) return false;                             // (the packed matrix has rows up to the next multiple of 128)
    return true;
}

const char* conv_pxd_kernel_name(int c) { return kPxd[c].name; }

template <int PXW, int FN, int WGM, int WGN, bool HAS_RES, bool OUT_F32>
static hipError_t launch_pxd_var(const ConvParams& p, hipStream_t st) {
    constexpr int BM = WGM * PXW * 16, BN = WGN * FN * 16;
    const int ntiles = (p.Cout + BN - 1) / BN;
    // whole rounds of 256 workgroups: the rounds the full tiles need, then the tile height that fills exactly those rounds
    int BMe = BM;
    if (tile_balance_enabled(1)) {
        const long full = (long)((p.M + BM - 1) / BM) * ntiles;
        const long rounds = (full + 255) / 256;
        const long lanes = rounds * 256 / ntiles;                  // pixel tiles that fit those rounds
        if (lanes > 0) BMe = (int)((p.M + lanes - 1) / lanes);
        if (BMe > BM) BMe = BM;
        if (BMe < 16) BMe = 16;
    }
    const int mtiles = (p.M + BMe - 1) / BMe;
    const size_t sh = (size_t)3 * (BN * 8 / 64) * 1024 + 1024;
    auto kern = conv_pxd_kernel<PXW, FN, WGM, WGN, HAS_RES, OUT_F32>;
    static bool attr = false;
    if (!attr && sh > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        if (e != hipSuccess) return e;
        attr = true;
    }
    hipLaunchKernelGGL(kern, dim3(mtiles * ntiles), dim3(WGM * WGN * 64), sh, st, p, mtiles, BMe);
    return hipGetLastError();
}

template <int PXW, int FN, int WGM, int WGN>
static hipError_t launch_pxd_one(const ConvParams& p, hipStream_t st) {
    if (p.out_f32) return launch_pxd_var<PXW, FN, WGM, WGN, false, true>(p, st);
    if (p.res) return launch_pxd_var<PXW, FN, WGM, WGN, true, false>(p, st);
    return launch_pxd_var<PXW, FN, WGM, WGN, false, false>(p, st);
}

hipError_t launch_conv_pxd(const ConvParams& p, int c, hipStream_t st) {
    switch (c) {
        case 0: return launch_pxd_one<2, 16, 8, 1>(p, st);
        case 1: return launch_pxd_one<2, 8, 8, 1>(p, st);
        case 2: return launch_pxd_one<2, 8, 4, 2>(p, st);
        case 3: return launch_pxd_one<1, 8, 8, 1>(p, st);
        case 4: return launch_pxd_one<1, 8, 4, 2>(p, st);
        case 5: return launch_pxd_one<2, 4, 8, 1>(p, st);
        case 6: return launch_pxd_one<1, 16, 8, 1>(p, st);
        default: return launch_pxd_one<1, 4, 8, 1>(p, st);
    }
}

}  // namespace yp
