// Pointwise 1x1 convolution -> per-channel spatial operator, one workgroup per (image, slice of NS output channels): the form the
// small-map part of the network takes (YOLOv10-S at 640x640: the 20x20 layers model.8 / 9 / 10 / 22 and the P5 class branch; SURVEY A.2
// [U] blocks CIB, SPPF, PSA, executed inside `.predict`, reference yolo_seg/app.py:91).
//
// Why this shape (round 4). At M = B * 20 * 20 = 12 800 pixels a 1x1 layer is 100-400 tiles of the LDS-DMA GEMM kernels on 256 CUs: two
// rounds of workgroups of 4-16 k-steps each, 13-28 us per layer for 2-7 us of operand ingest (profiles/r03_per_op_table.txt: 0.49 ms over
// 27 launches, <= 0.17 of the roofline), and every depthwise conv / pool behind it is one more 10-28-us launch over a 13-26-MB tensor.
// Depthwise and pooling operators mix pixels but not channels, a 1x1 convolution mixes channels but not pixels. A workgroup that owns ALL
// pixels of one image and a SLICE of the output channels can therefore run the 1x1 GEMM for its slice and then the spatial operator on
// its own result, which never leaves the chip:
//   GEMM     C[HW px][NS ch] = X[HW][K] * W[NS][K]^T; both operands travel by LDS-DMA in whole 128-byte lines (8 pixel rows x 64 channels per
//            1-KiB piece) through rings shared by the 16 waves - two slots each for the pixel rows of fragments 0..15 and 16..31, three for
//            the [NS][64] weight block of a k-step -, every wave owns up to two 16-pixel fragments and all NS/16 channel fragments.
//            Ingest per workgroup: HW*K*2 B of pixels (through L2: the NS-slices of an image run on the same XCD) + NS*K*2 B of weights.
//   epilogue bias + SiLU + bf16 -> (optionally) the pointwise result's own tensor in HBM, and an NHWC image of the slice in LDS with a
//            zero halo;
//   spatial  depthwise 3x3 / 7x7 (+ bias, SiLU, residual) on fp32 VALU - a thread = one 5x3 (3x3) output patch x 2 channels, v_pk_fma_f32,
//            taps in (ky, kx) order like the stand-alone kernel - or SPPF's three chained 5x5 max-pools (separable, order-preserving int16
//            keys as in sppf_pool3_bf16_kernel); only this result is written.
// SP = 0 is the plain GEMM in the same decomposition (the 1x1 layers between the fused pairs).
#include "common.h"
#include <cstdlib>
#include <algorithm>

namespace yp {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((address_space(3))) void lds_void;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short short2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned ps_key2(unsigned d) {               // bf16 pair -> order-preserving int16 pair (an involution)
    const unsigned s = (d >> 15) & 0x00010001u;
    return d ^ ((s << 15) - s);
}
__device__ __forceinline__ uint4 ps_key4(const uint4 v) { return make_uint4(ps_key2(v.x), ps_key2(v.y), ps_key2(v.z), ps_key2(v.w)); }
__device__ __forceinline__ unsigned ps_max2(unsigned a, unsigned b) {
    const short2v r = __builtin_elementwise_max(__builtin_bit_cast(short2v, a), __builtin_bit_cast(short2v, b));
    return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ uint4 ps_max4(const uint4 a, const uint4 b) { return make_uint4(ps_max2(a.x, b.x), ps_max2(a.y, b.y), ps_max2(a.z, b.z), ps_max2(a.w, b.w)); }
__device__ __forceinline__ float ps_silu(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f)); }

// phase stamps (core clock, s_memtime) of workgroup 0 / wave 0 in the last launch: [0] start, [1] prologue issued, [2] GEMM done, [3] epilogue
// done, [4] behind the barrier, [5] spatial stage done; [8 + g] at the top of k-step g (g < 16). yp_debug_pwsp_clocks reads them.
__device__ unsigned long long g_ps_clk[32];
#define PS_STAMP(i) do { if (blockIdx.x == 0 && tid == 0) g_ps_clk[i] = __builtin_readcyclecounter(); } while (0)
hipError_t pwsp_read_clocks(unsigned long long* out32) { return hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_ps_clk), 32 * sizeof(unsigned long long)); }

constexpr int PS_MF = 2;                 // 16-pixel fragments per wave: 16 waves x 2 x 16 = 512 pixels per image at most
constexpr int PS_NW = 16;                // waves per workgroup: the rate at which a CU takes operands in grows with the number of waves that issue loads
constexpr int PS_NT = PS_NW * 64;

template <int NS, int SP>
__global__ __launch_bounds__(PS_NT) void pwsp_kernel(const PwSpParams p) {
    constexpr int FN = NS / 16;
    constexpr int BK = 64, NSLOT = 3;
    constexpr int W_INSTR = NS / 8;                    // 1-KiB weight pieces per k-step ([NS][64] bf16, 8 rows per piece)
    constexpr int SLOT = W_INSTR * 1024;
    constexpr unsigned OOB = 0x80000000u;
    constexpr int KS = SP == 1 ? 3 : SP == 2 ? 7 : 1, PAD = KS / 2;
    constexpr int PIX = SP == 3 ? NS * 2 : NS * 2 + 8; // bytes per pixel of the LDS image (depthwise: +8 so that 16 pixels spread over the banks)
    constexpr int DW_NP = NS / 2, DW_G = PS_NT / DW_NP; // depthwise stage: channel pairs, thread groups (32 at NS = 64, 64 at NS = 32)
    constexpr int DW_PH = NS == 64 ? 5 : 3, DW_PW = 3;   // output patch of a thread (20 x 20: 4 x 7 = 28 patches of 5 x 3, 7 x 7 = 49 of 3 x 3)
    static_assert(SP != 3 || NS == 32, "the pool form keeps 64-byte pixel rows");

    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char* const dump = smem + NSLOT * SLOT;
    unsigned char* const img = smem + NSLOT * SLOT + 1024;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fc = lane >> 4;
    const int HW = p.H * p.W;
    const int nsl = p.C1 / NS;

    // image -> XCD: workgroup ids go round the 8 XCDs, so the slices of one image take ids of one residue class and share an L2
    int b, sl;
    {
        const int bid = blockIdx.x, nwg = gridDim.x;
        if ((p.B & 7) == 0) { const int xcd = bid & 7, j = bid >> 3; b = (j / nsl) * 8 + xcd; sl = j % nsl; }
        else { b = bid / nsl; sl = bid % nsl; }
        (void)nwg;
    }
    PS_STAMP(0);
    const int n0 = sl * NS;
    const bool spatial = SP != 0 && n0 >= p.sp_c0 && n0 + NS <= p.sp_c0 + p.Csp;
    const int nk = (p.Kpad1 + BK - 1) / BK;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w1, 0, (int)p.w1_bytes, 0x00020000);

    // the spatial stage's image in LDS: whole output patches + halo
    const int dgx = (p.W + DW_PW - 1) / DW_PW, dgy = (p.H + DW_PH - 1) / DW_PH;
    const int IW = (SP == 1 || SP == 2) ? dgx * DW_PW + 2 * PAD : p.W, IH = (SP == 1 || SP == 2) ? dgy * DW_PH + 2 * PAD : p.H;
    // ---- GEMM: BOTH operands by LDS-DMA in whole 128-byte lines ---------------------------------------------------------------------------
    // (first form of this kernel: the pixel operand global -> registers in fragment layout as conv_pxd does. A fragment's 16 lanes of a quad
    //  group are 16 different pixel rows, 64 bytes of each per instruction - half-used cache lines: 19 B/clk/CU of ingest, 3.3 k cycles per
    //  k-step of 64 KB whatever the number of issuing waves (8 or 16), unchanged with the MFMAs removed or with the slices of an image started
    //  at different k. A DMA piece is 8 rows x 128 B.)
    // Stages: A_g = weight block of k-step g + the pixel rows of fragments 0..15, B_g = the pixel rows of fragments 16... Two slots each for
    // A and B pixels, three for the weights (B_g still reads block g while A_g+2 is issued). Per wave and stage 3 (A) / 2 (B) pieces,
    // issued whether they exist or not (the rest go to a dump slot with an out-of-range source), so that one counted wait fits every step:
    //   (g,A): wait vmcnt(5) [B_g, A_g+1 may fly] ; barrier ; issue B_g+1 ; weight fragments of g -> registers ; MFMAs of fragment `wave`
    //   (g,B): wait vmcnt(5) [A_g+1, B_g+1]       ; barrier ; issue A_g+2 ; MFMAs of fragment 16 + wave
    constexpr int PXS = 256 * 128;                         // one pixel slot: 256 rows of 128 B
    unsigned char* const wring = smem;                     // [3][NS][128 B]
    unsigned char* const pxa = img;                        // [2][PXS]  (the spatial stage's image is laid over these slots after the GEMM)
    unsigned char* const pxb = img + 2 * PXS;              // [2][PXS]
    const int nmf = (HW + 15) >> 4;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
    // this lane's role in a pixel piece: row lane >> 3 of the piece's 8 rows, 16-byte position lane & 7, which holds source chunk
    // (lane & 7) ^ ((row >> 1) & 7) - the swizzle the fragment reads undo
    // (K % 64 == 32 - the 288- / 96-channel layers of v10-M / -N: the last k-step's upper four chunks lie behind the row's K channels, in the
    // next pixel / the next weight row. Both operands read zeros there: `kc` = the lane's chunk, compared against K per step.)
    unsigned pa_off[2], pb_off[2];
    int p_kc[2], w_kc;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (wave * 2 + j) * 8 + (lane >> 3);                     // row within the slot = pixel (A) / pixel - 256 (B)
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        p_kc[j] = c * 8;
        pa_off[j] = (row < HW && row < 256) ? (unsigned)((b * HW + row) * p.x_stride + p.x_coff + c * 8) * 2u : OOB;
        pb_off[j] = (row + 256 < HW) ? (unsigned)((b * HW + row + 256) * p.x_stride + p.x_coff + c * 8) * 2u : OOB;
    }
    unsigned wconst;
    {
        const int s_ = wave * 64 + lane;
        const int row = s_ >> 3, pc = s_ & 7;
        const int c = pc ^ ((row >> 1) & 7);
        w_kc = c * 8;
        wconst = (wave < W_INSTR) ? (unsigned)(((n0 + row) * p.Kpad1 + c * 8) * 2) : OOB;
    }
    auto kbyte = [&](int kt) { return (unsigned)(kt * BK) * 2u; };
    auto issue_A = [&](int kt) {
        const bool live = kt < nk;
        const unsigned kb = kbyte(kt);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_void*)((wave < W_INSTR) ? wring + (kt % NSLOT) * SLOT + wave * 1024 : dump), 16,
                                                 (live && wconst != OOB && kt * BK + w_kc < p.K) ? wconst + kb : OOB, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_void*)(pxa + (kt & 1) * PXS + (wave * 2 + j) * 1024), 16,
                                                     (live && pa_off[j] != OOB && kt * BK + p_kc[j] < p.K && !(p.dbg & 2)) ? pa_off[j] + kb : OOB, 0, 0, 0);
    };
    auto issue_B = [&](int kt) {
        const bool live = kt < nk;
        const unsigned kb = kbyte(kt);
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_void*)(pxb + (kt & 1) * PXS + (wave * 2 + j) * 1024), 16,
                                                     (live && pb_off[j] != OOB && kt * BK + p_kc[j] < p.K && !(p.dbg & 2)) ? pb_off[j] + kb : OOB, 0, 0, 0);
    };

    f32x4 acc[FN][PS_MF];
#pragma unroll
    for (int a = 0; a < FN; ++a) {
        const float4 b4 = *(const float4*)(p.bias1 + n0 + a * 16 + fc * 4);     // bias rides in the accumulator
#pragma unroll
        for (int f = 0; f < PS_MF; ++f) acc[a][f] = f32x4{b4.x, b4.y, b4.z, b4.w};
    }
    const bool hasA = wave < nmf, hasB = wave + 16 < nmf;                        // (wave-uniform)
    issue_A(0); issue_B(0); issue_A(1);
    PS_STAMP(1);
    // fragment read: row r of a [rows][128 B] block, k-substep s, this lane's 8 k (chunk s * 4 + fc)
    const unsigned fro = (unsigned)(fr * 128), frs = (unsigned)((fr >> 1) & 7);
    for (int g = 0; g < nk; ++g) {
        if (g < 16) PS_STAMP(8 + g);
        __builtin_amdgcn_s_waitcnt(5 | (7 << 4) | (0xF << 8));                   // vmcnt(5)
        __builtin_amdgcn_s_barrier();
        issue_B(g + 1);
        const unsigned char* ws = wring + (g % NSLOT) * SLOT;
        bf16x8 wf[2][FN];
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
            for (int a = 0; a < FN; ++a) wf[s_][a] = *(const bf16x8*)(ws + a * 2048 + fro + (((unsigned)(s_ * 4 + fc) ^ frs) << 4));   // (row = a * 16 + fr: (row >> 1) & 7 = (fr >> 1) & 7)
        if (hasA && !(p.dbg & 1)) {
            const unsigned char* ps = pxa + (g & 1) * PXS + wave * 2048;
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_) {
                const bf16x8 xf = *(const bf16x8*)(ps + fro + (((unsigned)(s_ * 4 + fc) ^ frs) << 4));
#pragma unroll
                for (int a = 0; a < FN; ++a) acc[a][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s_][a], xf, acc[a][0], 0, 0, 0);
            }
        }
        __builtin_amdgcn_s_waitcnt(5 | (7 << 4) | (0xF << 8));                   // vmcnt(5)
        __builtin_amdgcn_s_barrier();
        issue_A(g + 2);
        if (hasB && !(p.dbg & 1)) {
            const unsigned char* ps = pxb + (g & 1) * PXS + wave * 2048;
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_) {
                const bf16x8 xf = *(const bf16x8*)(ps + fro + (((unsigned)(s_ * 4 + fc) ^ frs) << 4));
#pragma unroll
                for (int a = 0; a < FN; ++a) acc[a][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s_][a], xf, acc[a][1], 0, 0, 0);
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (0xF << 8));                       // the zero-fill pieces of the stages past the end
    __syncthreads();                                                             // every wave is done with the pixel slots: the image goes over them
    // ---- the LDS image (whole output patches + halo) starts as zeros: the depthwise padding ---------------------------------------------
    if (SP == 1 || SP == 2) {
        if (spatial) {
            const int n16 = (IH * IW * PIX + 15) >> 4;
            for (int i = tid; i < n16; i += PS_NT) *(uint4*)(img + (size_t)i * 16) = make_uint4(0, 0, 0, 0);
            __syncthreads();
        }
    }

    PS_STAMP(2);
    // ---- epilogue: activation, bf16; to HBM when the pointwise result has readers of its own, into the LDS image for the spatial stage ----
#pragma unroll
    for (int f = 0; f < PS_MF; ++f) {
        const int m = (wave + PS_NW * f) * 16 + fr;
        const bool ok = m < HW;
        const int py = ok ? m / p.W : 0, px = ok ? m - py * p.W : 0;
#pragma unroll
        for (int a = 0; a < FN; ++a) {
            float v[4] = {acc[a][f][0], acc[a][f][1], acc[a][f][2], acc[a][f][3]};
            if (p.act1 == ACT_SILU) silu4_packed(v);
            const int co = a * 16 + fc * 4;
            if (SP == 0 && p.res1 && ok) {
                const uint2 rr = *(const uint2*)((const __bf16*)p.res1 + (size_t)(b * HW + m) * p.res1_stride + p.res1_coff + n0 + co);
                v[0] += __uint_as_float(rr.x << 16); v[1] += __uint_as_float(rr.x & 0xffff0000u);
                v[2] += __uint_as_float(rr.y << 16); v[3] += __uint_as_float(rr.y & 0xffff0000u);
            }
            __attribute__((aligned(8))) __bf16 o[4] = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
            if (ok && p.y1) *(uint2*)((__bf16*)p.y1 + (size_t)(b * HW + m) * p.y1_stride + p.y1_coff + n0 + co) = *(const uint2*)o;
            if (SP != 0 && ok && spatial) *(uint2*)(img + (size_t)((py + PAD) * IW + px + PAD) * PIX + co * 2) = *(const uint2*)o;
        }
    }
    PS_STAMP(3);
    if (SP == 0) return;
    if (!spatial || (p.dbg & 4)) return;                    // (workgroup-uniform)
    __syncthreads();
    PS_STAMP(4);

    const int cs0 = n0 - p.sp_c0;                           // first channel of this slice within the spatial operator's channel range
    if constexpr (SP == 1 || SP == 2) {
        // ---- depthwise KS x KS: a thread = one PH x PW output patch x 2 channels ----------------------------------------------------
        typedef __attribute__((address_space(3))) const unsigned* lds_u32p;
        typedef __attribute__((address_space(3))) const f32x2* lds_f2p;
        typedef __attribute__((address_space(3))) f32x2* lds_f2w;
        const int cp = tid % DW_NP, grp = tid / DW_NP;
        const int c = cs0 + cp * 2;                         // channel pair (c, c + 1) of the depthwise operator
        const unsigned img_l = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)img;
        const unsigned wl_l = img_l + (unsigned)(((IH * IW * PIX) + 15) & ~15);
        // 3x3: the nine weight pairs of the thread's channels live in registers. 7x7: 49 pairs (98 registers) beside 25 accumulator pairs
        // and an input row do not fit in the 256 registers two waves per SIMD leave a wave - they are unpacked to fp32 pairs in LDS once
        // per workgroup and read per (ky, kx) (one ds_read_b64 per PW packed FMAs)
        f32x2 wk[KS == 3 ? 9 : 1];
        if constexpr (KS == 3) {
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const unsigned d = *(const unsigned*)((const __bf16*)p.wd + (size_t)t * p.Csp + c);
                wk[t] = f32x2{__uint_as_float(d << 16), __uint_as_float(d & 0xffff0000u)};
            }
        } else {
            for (int t = tid; t < KS * KS * DW_NP; t += PS_NT) {
                const unsigned d = *(const unsigned*)((const __bf16*)p.wd + (size_t)(t / DW_NP) * p.Csp + cs0 + (t % DW_NP) * 2);
                *(lds_f2w)(size_t)(wl_l + (unsigned)t * 8u) = f32x2{__uint_as_float(d << 16), __uint_as_float(d & 0xffff0000u)};
            }
            __syncthreads();
        }
        const f32x2 bias = f32x2{p.biasd[c], p.biasd[c + 1]};
        const int npatch = dgx * dgy;
        for (int q = grp; q < npatch; q += DW_G) {
            const int qy = q / dgx, qx = q - qy * dgx;
            const int oy0 = qy * DW_PH, ox0 = qx * DW_PW;
            f32x2 o[DW_PH][DW_PW];
#pragma unroll
            for (int i = 0; i < DW_PH; ++i)
#pragma unroll
                for (int j = 0; j < DW_PW; ++j) o[i][j] = bias;
            // the image in LDS has IH x IW pixels (whole patches + halo, zero where the map has no pixel): no clamping below
            unsigned rowa = img_l + (unsigned)((oy0 * IW + ox0) * PIX + cp * 4);
#pragma unroll
            for (int r = 0; r < DW_PH + KS - 1; ++r) {
                unsigned wa = wl_l + (unsigned)cp * 8u;
                asm volatile("" : "+v"(wa));                // (opaque per input row: the rows' weight reads must not be merged into 49 live pairs)
                f32x2 xin[DW_PW + KS - 1];
#pragma unroll
                for (int cc = 0; cc < DW_PW + KS - 1; ++cc) {
                    const unsigned d = *(lds_u32p)(size_t)(rowa + (unsigned)(cc * PIX));
                    xin[cc] = f32x2{__uint_as_float(d << 16), __uint_as_float(d & 0xffff0000u)};
                }
#pragma unroll
                for (int i = 0; i < DW_PH; ++i) {
                    const int ky = r - i;
                    if (ky < 0 || ky >= KS) continue;
#pragma unroll
                    for (int kx = 0; kx < KS; ++kx) {
                        f32x2 w;
                        if constexpr (KS == 3) w = wk[ky * 3 + kx];
                        else w = *(lds_f2p)(size_t)(wa + (unsigned)((ky * KS + kx) * DW_NP * 8));
#pragma unroll
                        for (int j = 0; j < DW_PW; ++j) o[i][j] = __builtin_elementwise_fma(xin[j + kx], w, o[i][j]);
                    }
                }
                rowa += (unsigned)(IW * PIX);
                // one input row at a time: the accumulators pass through an empty asm statement, so this row's FMAs cannot be sunk behind
                // the next rows' loads (the optimiser otherwise gathers all 121 input pairs first and spills)
#pragma unroll
                for (int i = 0; i < DW_PH; ++i) {
                    asm volatile("" : "+v"(o[i][0]), "+v"(o[i][1]), "+v"(o[i][2]));
                }
            }
            const unsigned pix0 = (unsigned)(b * HW + oy0 * p.W + ox0);
#pragma unroll
            for (int i = 0; i < DW_PH; ++i)
#pragma unroll
                for (int j = 0; j < DW_PW; ++j) {
                    if (oy0 + i >= p.H || ox0 + j >= p.W) continue;
                    float v0 = o[i][j][0], v1 = o[i][j][1];
                    if (p.actd == ACT_SILU) { v0 = ps_silu(v0); v1 = ps_silu(v1); }
                    const unsigned pix = pix0 + (unsigned)(i * p.W + j);
                    if (p.res) {
                        const unsigned rr = *(const unsigned*)((const __bf16*)p.res + (size_t)(pix * (unsigned)p.res_stride + (unsigned)(p.res_coff + c)));
                        v0 += __uint_as_float(rr << 16); v1 += __uint_as_float(rr & 0xffff0000u);
                    }
                    __attribute__((aligned(4))) __bf16 ob[2] = {(__bf16)v0, (__bf16)v1};
                    *(unsigned*)((__bf16*)p.y2 + (size_t)(pix * (unsigned)p.y2_stride + (unsigned)(p.y2_coff + c))) = *(const unsigned*)ob;
                }
        }
    } else if constexpr (SP == 3) {
        // ---- SPPF: m1 = pool5(y), m2 = pool5(m1), m3 = pool5(m2) -> channel slices 0, 1, 2 of y2 (each Csp wide) ------------------------
        unsigned char* const bufA = img;                               // [HW][64 B]: the pointwise result (raw bf16), then keys
        unsigned char* const bufB = img + (size_t)HW * 64;
        unsigned char* const tmp = img + (size_t)HW * 128;
        const int items = HW * 4, W = p.W, H = p.H;
        for (int i = tid; i < items; i += PS_NT) *(uint4*)(bufA + i * 16) = ps_key4(*(const uint4*)(bufA + i * 16));
        __syncthreads();
        for (int stage = 0; stage < 3; ++stage) {
            const unsigned char* src = (stage & 1) ? bufB : bufA;
            unsigned char* dst = (stage & 1) ? bufA : bufB;
            for (int i = tid; i < items; i += PS_NT) {                   // row pass (clamped neighbours stand for the -inf padding)
                const int pxl = i >> 2, y = pxl / W, x = pxl - y * W;
                const int x0 = max(x - 2, 0), x1 = max(x - 1, 0), x3 = min(x + 1, W - 1), x4 = min(x + 2, W - 1);
                const unsigned char* row = src + (size_t)(y * W) * 64 + (i & 3) * 16;
                uint4 m = *(const uint4*)(row + x * 64);
                m = ps_max4(m, *(const uint4*)(row + x0 * 64));
                m = ps_max4(m, *(const uint4*)(row + x1 * 64));
                m = ps_max4(m, *(const uint4*)(row + x3 * 64));
                m = ps_max4(m, *(const uint4*)(row + x4 * 64));
                *(uint4*)(tmp + i * 16) = m;
            }
            __syncthreads();
            __bf16* yb = (__bf16*)p.y2 + (size_t)b * HW * p.y2_stride + p.y2_coff + stage * p.Csp + cs0;
            for (int i = tid; i < items; i += PS_NT) {                   // column pass + store
                const int pxl = i >> 2, y = pxl / W, x = pxl - y * W;
                const int y0 = max(y - 2, 0), y1 = max(y - 1, 0), y3 = min(y + 1, H - 1), y4 = min(y + 2, H - 1);
                const unsigned char* col = tmp + (size_t)x * 64 + (i & 3) * 16;
                uint4 m = *(const uint4*)(col + (size_t)(y * W) * 64);
                m = ps_max4(m, *(const uint4*)(col + (size_t)(y0 * W) * 64));
                m = ps_max4(m, *(const uint4*)(col + (size_t)(y1 * W) * 64));
                m = ps_max4(m, *(const uint4*)(col + (size_t)(y3 * W) * 64));
                m = ps_max4(m, *(const uint4*)(col + (size_t)(y4 * W) * 64));
                *(uint4*)(dst + i * 16) = m;
                *(uint4*)(yb + (size_t)pxl * p.y2_stride + (i & 3) * 8) = ps_key4(m);
            }
            __syncthreads();
        }
    }
    PS_STAMP(5);
}

// -----------------------------------------------------------------------------------------------------------------------------------
static size_t pwsp_lds_bytes(const PwSpParams& p, int NS) {
    const size_t ring = (size_t)3 * (NS / 8) * 1024 + 1024;
    const size_t px = (size_t)4 * 256 * 128;                       // two A and two B pixel slots of 256 rows x 128 B
    size_t sp = 0;
    if (p.sp == 3) sp = (size_t)p.H * p.W * 64 * 3;
    else if (p.sp != 0) {
        const int pad = p.sp == 1 ? 1 : 3, PH = NS == 64 ? 5 : 3, PW = 3;
        const int IH = (p.H + PH - 1) / PH * PH + 2 * pad, IW = (p.W + PW - 1) / PW * PW + 2 * pad;
        sp = (((size_t)IH * IW * (NS * 2 + 8) + 15) & ~(size_t)15) + (p.sp == 2 ? (size_t)49 * (NS / 2) * 8 : 0);
    }
    return ring + std::max(px, sp);
}

int pwsp_slice(const PwSpParams& p) {
    // 64-channel slices while that still gives every CU a workgroup, else 32 (the pool form is written for 32)
    if (p.sp == 3) return 32;
    static const int force = [] { const char* v = std::getenv("YOLOP_PWSP_NS"); return v ? atoi(v) : 0; }();      // experiment: slice width
    if (force == 32) return 32;
    if (force == 64 && (p.C1 % 64) == 0 && (p.sp == 0 || ((p.sp_c0 % 64) == 0 && (p.Csp % 64) == 0))) return 64;
    // (>= 96 workgroups: the wider slice halves the redundant reads of the image's pixels - every slice's workgroup pulls all of them through its
    // CU - and with it the CU time of the launch: forced 32 / as before / forced 64 on the bench workload 1.569 / 1.446 / 1.435 ms per step with two
    // batches in flight, 1.800 / 1.682 / 1.682 with one; below that the narrow slice's extra workgroups still shorten a lone small launch)
    if ((p.C1 % 64) == 0 && (p.sp == 0 || ((p.sp_c0 % 64) == 0 && (p.Csp % 64) == 0)) && (long)p.B * (p.C1 / 64) >= 96) return 64;
    return 32;
}

bool pwsp_valid(const PwSpParams& p) {
    if (p.sp < 0 || p.sp > 3) return false;
    const int NS = pwsp_slice(p);
    if ((p.C1 % NS) != 0 || (p.K % 32) != 0 || p.Kpad1 != p.K || p.K < 64) return false;
    if (p.H * p.W > 16 * PS_MF * PS_NW || p.H < 1 || p.W < 1) return false;
    if ((p.x_stride & 7) || (p.x_coff & 7) || p.x_bytes >= (1ull << 31) || p.w1_bytes >= (1ull << 31)) return false;
    if (p.y1 && ((p.y1_stride & 3) || (p.y1_coff & 3))) return false;
    if (p.res1 && (p.sp != 0 || (p.res1_stride & 3) || (p.res1_coff & 3))) return false;
    if ((p.C1 + 127) / 128 * 128 < p.C1) return false;
    if (p.sp != 0) {
        if (!p.y2 || (p.sp_c0 % NS) != 0 || (p.Csp % NS) != 0 || p.sp_c0 < 0 || p.sp_c0 + p.Csp > p.C1) return false;
        if ((p.y2_stride & 7) || (p.y2_coff & 7)) return false;
        if (p.res && ((p.res_stride & 1) || (p.res_coff & 1))) return false;
        if (p.sp == 3 && p.res) return false;
    } else if (!p.y1) return false;
    return pwsp_lds_bytes(p, NS) <= 160 * 1024;
}

const char* pwsp_kernel_name(const PwSpParams& p) {
    const int NS = pwsp_slice(p);
    static const char* n[2][4] = {{"pwsp_kernel<32,0>", "pwsp_kernel<32,1>", "pwsp_kernel<32,2>", "pwsp_kernel<32,3>"},
                                  {"pwsp_kernel<64,0>", "pwsp_kernel<64,1>", "pwsp_kernel<64,2>", "pwsp_kernel<64,3>"}};
    return n[NS == 64][p.sp];
}

template <int NS, int SP>
static hipError_t launch_pwsp_t(const PwSpParams& p, hipStream_t st) {
    const size_t sh = pwsp_lds_bytes(p, NS);
    auto kern = pwsp_kernel<NS, SP>;
    static size_t attr = 0;
    if (sh > attr && sh > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr = 160 * 1024;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(p.B * (p.C1 / NS))), dim3(PS_NT), sh, st, p);
    return hipGetLastError();
}

hipError_t launch_pwsp(const PwSpParams& p, hipStream_t st) {
    if (!pwsp_valid(p)) return hipErrorInvalidValue;
    const int NS = pwsp_slice(p);
    if (NS == 64) {
        switch (p.sp) {
            case 0: return launch_pwsp_t<64, 0>(p, st);
            case 1: return launch_pwsp_t<64, 1>(p, st);
            case 2: return launch_pwsp_t<64, 2>(p, st);
            default: return hipErrorInvalidValue;
        }
    }
    switch (p.sp) {
        case 0: return launch_pwsp_t<32, 0>(p, st);
        case 1: return launch_pwsp_t<32, 1>(p, st);
        case 2: return launch_pwsp_t<32, 2>(p, st);
        default: return launch_pwsp_t<32, 3>(p, st);
    }
}

}  // namespace yp
