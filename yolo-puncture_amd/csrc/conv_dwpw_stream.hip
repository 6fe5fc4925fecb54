// Streaming form of the fused depthwise 3x3 (+bias, SiLU) -> pointwise 1x1 (+bias, SiLU) pair for 128-channel inputs (bf16): the class
// branch of the v10Detect head on the 80x80 / 40x40 maps (`one2one_cv3.{l}.{0,1}`; SURVEY.md Appendix A.4 [U], run inside `.predict`,
// reference yolo_seg/app.py:91) - the largest time share of the step in `conv_dwpw_kernel`'s chunked form (4 launches, 150 us).
//
// Why a second form (DESIGN.md "conv_dwpw ablations"): the chunked kernel walks a tile in four 32-channel phases of two barriers each; its
// skeleton alone takes 33-39 us of 52, its stages add up linearly, and co-resident workgroups do not help. Here the shape that brought
// the 1x1 layers to 4 TB/s (conv_wres.hip): a persistent workgroup, an 8x16-pixel tile with its 1-pixel halo as WHOLE pixel rows
// (256 contiguous bytes each, LDS-DMA, double-buffered), TWO barriers per tile:
//   (a) rows of this tile landed / everybody is done with the previous tile -> issue the next tile's rows
//       depthwise stage on the VALU: wave r takes output row r, lane c the channel pair (2c, 2c + 1); a 3x3 window of bf16 pairs slides
//       along the row (three 4-byte LDS reads per output pair, nine `v_pk_fma_f32`), bias + SiLU + bf16 -> the pixel-operand tile
//   (b) pixel-operand tile complete -> pointwise GEMM: wave w owns output channels [16w, 16w + 16) with its weights IN REGISTERS
//       (4 fragments = 16 VGPRs, loaded once per workgroup) and multiplies all 8 pixel fragments of the tile; bias rides in the
//       accumulators; SiLU, bf16, 8 buffer stores per wave (a fixed number: the wait in front of the next tile is a counted vmcnt).
// The depthwise result is rounded to bf16 exactly where the unfused graph stores it; fp32 accumulation in both stages.
#include "common.h"
#include <cstdlib>

namespace yp {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((address_space(3))) void lds_void;

constexpr int DS_TH = 8, DS_TW = 16;                 // output tile (one row per wave, one MFMA pixel fragment per row)
constexpr int DS_C = 128;                            // input channels = 64 pairs = one per lane
constexpr int DS_HP = (DS_TH + 2) * (DS_TW + 2);     // 180 halo pixels
constexpr int DS_PIECES = 48;                        // 1-KiB pieces of a halo slot, 6 per wave (45 in use: 4 pixels of 256 B each)
constexpr int DS_HB = DS_PIECES * 1024;
constexpr int DS_AB = DS_TH * DS_TW * 256;           // pixel-operand tile [128 px][128 ch] bf16

template <int N> __device__ __forceinline__ void ds_wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt range");
    __builtin_amdgcn_s_waitcnt((N & 0xF) | (7 << 4) | (0xF << 8) | (((N >> 4) & 3) << 14));
}
__device__ __forceinline__ unsigned ds_lds_addr(const void* p) { return (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)p; }

// NW = 8: one workgroup per CU, two halo slots (the next tile's rows fly under this tile's stages). NW = 4: TWO co-resident workgroups of
// four waves with ONE halo slot each (80 KB): a wave takes two output rows and 32 output channels; the rows of the next tile are issued
// behind barrier (b), when the slot is free, and what hides them is the other workgroup, which is in another stage - the VALU-bound
// stages (two SiLUs per element: 128 quarter-rate transcendentals per wave and tile) then keep the SIMDs busy across each other's barriers.
template <int NW>
__global__ __launch_bounds__(NW * 64, 2) void conv_dwpw_stream_kernel(const DwPwParams p, const int tiles_h, const int tiles_w, const int num_tiles, const int G) {
    constexpr int RPW = DS_TH / NW;                                // output rows per wave (depthwise stage)
    constexpr int NFW = 8 / NW;                                    // 16-channel output fragments per wave (pointwise stage)
    constexpr int NSLOT = NW == 8 ? 2 : 1;
    constexpr int NST = DS_TH * NFW;                               // stores per wave and tile
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char* const Hs = smem;                                // 2 halo slots: pixel hp at hp * 256, channel c at c * 2
    unsigned char* const As = smem + NSLOT * DS_HB;                    // [128 px][256 B], 16-byte chunk q of pixel px at position q ^ (px & 15)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fc = lane >> 4;

    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w_pw, 0, (int)p.wpw_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, (int)p.y_bytes, 0x00020000);

    // six pieces per wave and tile, whether they exist or not (pieces 45..47 and the tiles behind the end read nothing)
    auto issue_tile = [&](int tile, int slot) {
        int t = tile;
        const int tw = t % tiles_w; t /= tiles_w;
        const int th = t % tiles_h;
        const int b = t / tiles_h;
        const int h0 = th * DS_TH - 1, w0 = tw * DS_TW - 1;
        unsigned char* const dst = Hs + slot * DS_HB;
#pragma unroll
        for (int j = 0; j < DS_PIECES / NW; ++j) {
            const int ii = wave + j * NW;
            const int s = ii * 64 + lane;
            const int hp = s >> 4, c = s & 15;
            const int hy = hp / (DS_TW + 2), hx = hp - hy * (DS_TW + 2);
            const int hi = h0 + hy, wi = w0 + hx;
            const bool ok = tile < num_tiles && hp < DS_HP && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            const unsigned voff = ok ? (unsigned)((((b * p.H + hi) * p.W + wi) * p.x_stride + p.x_coff + c * 8) * 2) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_void*)(dst + ii * 1024), 16, voff, 0, 0, 0);
        }
    };
    int tile = blockIdx.x;
    issue_tile(tile, 0);

    // ---- operands that stay in registers ------------------------------------------------------------------------------------------------
    // depthwise: taps and bias of this lane's channel pair as fp32 pairs
    f32x2 wd[9], bd;
    {
        const unsigned* w32 = (const unsigned*)p.w_dw;             // packed [9][C] bf16: the pair (2 * lane, 2 * lane + 1) is dword `lane` of a tap row
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const unsigned u = w32[t * (DS_C / 2) + lane];
            wd[t] = f32x2{__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)};
        }
        bd = f32x2{p.b_dw[2 * lane], p.b_dw[2 * lane + 1]};
    }
    // pointwise: this wave's 16 output channels x 128 k as four A fragments; k order inside a pair of substeps chosen so that a lane's two
    // loads are 32 contiguous bytes (conv_wrs.hip): lane group fc holds channels [fc * 16, fc * 16 + 16) of the pair's 64, the first 8 in
    // the even substep
    bf16x8 wreg[4][NFW];
    float bpw[NFW][4];
#pragma unroll
    for (int i = 0; i < NFW; ++i) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const unsigned voff = (unsigned)((((wave * NFW + i) * 16 + fr) * p.Kpad + (ks >> 1) * 64 + fc * 16 + (ks & 1) * 8) * 2);
            const __attribute__((ext_vector_type(4))) unsigned v = __builtin_amdgcn_raw_buffer_load_b128(wrs, voff, 0, 0);
            wreg[ks][i] = __builtin_bit_cast(bf16x8, v);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int co = (wave * NFW + i) * 16 + fc * 4 + r; bpw[i][r] = (co < p.Cout) ? p.b_pw[co] : 0.f; }
    }
    // (known complete before the loop, then passed through empty asm statements: see conv_wres.hip)
    ds_wait_vm<0>();
#pragma unroll
    for (int t = 0; t < 9; ++t) asm volatile("" : "+v"(wd[t]));
    asm volatile("" : "+v"(bd));
#pragma unroll
    for (int i = 0; i < NFW; ++i) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(wreg[ks][i]));
#pragma unroll
        for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(bpw[i][r]));
    }

    const unsigned as_l = ds_lds_addr(As);
    for (int it = 0; tile < num_tiles; tile += G, ++it) {
        // (a) this tile's rows have landed (issued in front of the previous tile's stores, which may still fly)
        if (it == 0) ds_wait_vm<0>();
        else ds_wait_vm<NST>();
        __builtin_amdgcn_s_barrier();
        const int slot = NSLOT == 2 ? (it & 1) : 0;
        if (NSLOT == 2) issue_tile(tile + G, slot ^ 1);
        int t = tile;
        const int tw = t % tiles_w; t /= tiles_w;
        const int th = t % tiles_h;
        const int b = t / tiles_h;
        // ---- depthwise stage: output rows wave * RPW ..., channel pair `lane` -------------------------------------------------------------
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            const int row = wave * RPW + rr;
            f32x2 win[3][3];                                       // win[ky][j]: column x + j of halo row `row` + ky
            auto rd = [&](int ky, int col) -> f32x2 {
                const unsigned u = *(const unsigned*)(Hs + slot * DS_HB + ((row + ky) * (DS_TW + 2) + col) * 256 + lane * 4);
                return f32x2{__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)};
            };
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) { win[ky][0] = rd(ky, 0); win[ky][1] = rd(ky, 1); }
#pragma unroll
            for (int x = 0; x < DS_TW; ++x) {
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) win[ky][2] = rd(ky, x + 2);
                f32x2 a = bd;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) a = __builtin_elementwise_fma(win[ky][kx], wd[ky * 3 + kx], a);
                if (p.act_dw == ACT_SILU) {
                    f32x2 e = a * -1.4426950408889634f;
                    e[0] = __builtin_amdgcn_exp2f(e[0]); e[1] = __builtin_amdgcn_exp2f(e[1]);
                    e = e + 1.0f;
                    e[0] = __builtin_amdgcn_rcpf(e[0]); e[1] = __builtin_amdgcn_rcpf(e[1]);
                    a = a * e;
                }
                __attribute__((aligned(4))) __bf16 o[2] = {(__bf16)a[0], (__bf16)a[1]};
                const int px = row * DS_TW + x;
                const unsigned dst = as_l + (unsigned)(px * 256 + (((lane >> 2) ^ (px & 15)) << 4) + (lane & 3) * 4);
                asm volatile("ds_write_b32 %0, %1" : : "v"(dst), "v"(*(const unsigned*)o) : "memory");
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) { win[ky][0] = win[ky][1]; win[ky][1] = win[ky][2]; }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                              // (b) the pixel-operand tile is complete, the halo slot has been read
        if (NSLOT == 1) issue_tile(tile + G, 0);
        // ---- pointwise stage: this wave's 16 * NFW channels x all 8 pixel fragments ------------------------------------------------------
        f32x4 acc[NFW][DS_TH];
#pragma unroll
        for (int i = 0; i < NFW; ++i)
#pragma unroll
            for (int f = 0; f < DS_TH; ++f) acc[i][f] = f32x4{bpw[i][0], bpw[i][1], bpw[i][2], bpw[i][3]};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int ch = (ks >> 1) * 8 + fc * 2 + (ks & 1);      // the chunk that holds this lane's 8 channels of the substep (see the weights)
            bf16x8 xf[DS_TH];
#pragma unroll
            for (int f = 0; f < DS_TH; ++f) xf[f] = *(const bf16x8*)(As + (f * 16 + fr) * 256 + ((ch ^ fr) << 4));
#pragma unroll
            for (int i = 0; i < NFW; ++i)
#pragma unroll
                for (int f = 0; f < DS_TH; ++f) acc[i][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[ks][i], xf[f], acc[i][f], 0, 0, 0);
        }
        const int wo = tw * DS_TW + fr;
#pragma unroll
        for (int i = 0; i < NFW; ++i) {
            const int co = (wave * NFW + i) * 16 + fc * 4;
#pragma unroll
            for (int f = 0; f < DS_TH; ++f) {
                const int ho = th * DS_TH + f;
                const bool ok = ho < p.H && wo < p.W && co < p.Cout;   // (Cout % 4 == 0: a lane's four channels exist together)
                float v[4] = {acc[i][f][0], acc[i][f][1], acc[i][f][2], acc[i][f][3]};
                if (p.act_pw == ACT_SILU) silu4_packed(v);
                __attribute__((aligned(8))) __bf16 o[4] = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                const unsigned off = ok ? (unsigned)((((b * p.H + ho) * p.W + wo) * p.y_stride + p.y_coff + co) * 2) : OOB;
                __builtin_amdgcn_raw_buffer_store_b64(*(const __attribute__((ext_vector_type(2))) unsigned*)o, yrs, off, 0, 0);
            }
        }
    }
    ds_wait_vm<0>();
}

bool dwpw_stream_valid(const DwPwParams& p) {
    static const bool off = [] { const char* v = std::getenv("YOLOP_NO_DWPW_STREAM"); return v && *v == '1'; }();   // A/B switch
    if (off || p.w3 || p.out_f32 || p.clk) return false;
    if (p.C != DS_C || p.Kpad != DS_C || p.Cout != 128) return false;      // (the class branch of the 128-wide heads: v10-S; other widths keep the chunked kernel)
    if ((p.x_stride & 7) || (p.x_coff & 7) || (p.y_stride & 3) || (p.y_coff & 3)) return false;
    if (p.x_bytes >= (1ull << 31) || p.y_bytes >= (1ull << 31) || p.wpw_bytes >= (1ull << 31)) return false;
    if (p.act_dw != ACT_SILU && p.act_dw != ACT_NONE) return false;
    if (p.act_pw != ACT_SILU && p.act_pw != ACT_NONE) return false;
    // partial tiles compute for nothing: the maps this form is for fill them (80x80, 40x40 exactly; others at least two thirds)
    return (long)((p.H + DS_TH - 1) / DS_TH * DS_TH) * ((p.W + DS_TW - 1) / DS_TW * DS_TW) * 2 <= (long)p.H * p.W * 3;
}

template <int NW>
static hipError_t launch_dwpw_stream_t(const DwPwParams& p, hipStream_t st) {
    const size_t sh = (size_t)(NW == 8 ? 2 : 1) * DS_HB + DS_AB;
    auto kern = conv_dwpw_stream_kernel<NW>;
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr = true;
    }
    const int tiles_h = (p.H + DS_TH - 1) / DS_TH, tiles_w = (p.W + DS_TW - 1) / DS_TW;
    const int num_tiles = p.B * tiles_h * tiles_w;
    const int gmax = NW == 8 ? 256 : 512;
    const int G = num_tiles < gmax ? num_tiles : gmax;
    hipLaunchKernelGGL(kern, dim3((unsigned)G), dim3(NW * 64), sh, st, p, tiles_h, tiles_w, num_tiles, G);
    return hipGetLastError();
}

hipError_t launch_dwpw_stream(const DwPwParams& p, hipStream_t st) {
    if (!dwpw_stream_valid(p)) return hipErrorInvalidValue;
    static const bool one = [] { const char* v = std::getenv("YOLOP_DWPW_STREAM_ONE"); return v && *v == '1'; }();   // A/B switch: one workgroup per CU
    return one ? launch_dwpw_stream_t<8>(p, st) : launch_dwpw_stream_t<4>(p, st);
}

}  // namespace yp
