// Streaming form of SCDown (1x1 conv + SiLU -> depthwise 3x3 stride 2, 256 output channels) for 128 or 256 input channels (bf16).
// SURVEY.md A.3 layers 5 / 20 [U] (SCDown = Conv(c1, c2, 1, 1) -> Conv(c2, c2, k=3, s=2, g=c2, act=False)), run inside `.predict`
// (reference yolo_seg/app.py:91).
//
// `scdown_fused_kernel` walks a tile in four 64-channel groups with the group's weights streamed through LDS (66 us for 79 MB on
// `model.5`: a quarter of the HBM roofline). Here the shape of conv_dwpw_stream.hip, stages in the other order: a persistent workgroup,
// a 4x4 tile of the block's OUTPUT = a 9x9 patch of the 1x1's pixels as WHOLE pixel rows by LDS-DMA (double-buffered), TWO barriers per tile:
//   (a) the patch landed / everybody is done with the previous tile -> issue the next patch
//       1x1 GEMM: wave w owns intermediate channels [32w, 32w + 32) with its weights IN REGISTERS (K / 32 x 2 fragments, loaded once per
//       workgroup) and multiplies all 6 pixel fragments of the patch; bias in the accumulators; SiLU, bf16 - zero where the patch pixel
//       lies outside the frame (the depthwise conv's padding) - into the LDS image t [81 px][256 ch]
//   (b) t complete -> depthwise 3x3 s2 on the VALU: a thread takes one output row of the tile (4 pixels) and one channel pair, a window of
//       bf16 pairs slides two columns per output (`v_pk_fma_f32`), + bias (+ activation), bf16, 4-byte stores (a wave = 256 contiguous bytes)
// t is rounded to bf16 exactly where the unfused graph stores it; fp32 accumulation in both stages.
#include "common.h"
#include <cstdlib>

namespace yp {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((address_space(3))) void lds_void;

constexpr int SS_NW = 8;
constexpr int SS_T = 4;                               // 4 x 4 output pixels per tile
constexpr int SS_R = 2 * SS_T + 1;                    // 9 x 9 patch
constexpr int SS_RP = SS_R * SS_R;                    // 81 patch pixels
constexpr int SS_FM = 6;                              // pixel fragments (96 rows; rows 81.. are nothing)
constexpr int SS_C = 256;                             // intermediate / output channels
constexpr int SS_TB = SS_FM * 16 * SS_C * 2;          // t: [96 px][256 ch] bf16 = 48 KB

template <int N> __device__ __forceinline__ void ss_wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt range");
    __builtin_amdgcn_s_waitcnt((N & 0xF) | (7 << 4) | (0xF << 8) | (((N >> 4) & 3) << 14));
}
__device__ __forceinline__ unsigned ss_lds_addr(const void* p) { return (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)p; }

template <int NKS>
__global__ __launch_bounds__(SS_NW * 64, 2) void scdown_stream_kernel(const ScdParams p, const int tiles_h, const int tiles_w, const int num_tiles, const int G) {
    constexpr int K = NKS * 32, RB = K * 2;                        // patch row bytes
    constexpr int XB = SS_FM * 16 * RB;                            // one patch slot
    constexpr int PIECES = XB / 1024;                              // 24 / 48
    constexpr int PPW = PIECES / SS_NW;                            // 3 / 6 per wave
    constexpr int CPR = RB / 16;                                   // 16-byte chunks per patch row
    constexpr int NST = SS_T;                                      // stores per wave and tile
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char* const Xs = smem;                                // 2 patch slots: row px, chunk q at position q ^ (px & 15)
    unsigned char* const Ts = smem + 2 * XB;                       // t: row px (512 B), chunk q at position q ^ (px & 15) (within its half of 16)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fc = lane >> 4;

    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w1, 0, (int)p.w1_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, (int)p.y_bytes, 0x00020000);

    // PPW pieces per wave and tile, whether they exist or not (rows 81.. and the tiles behind the end read nothing)
    auto issue_tile = [&](int tile, int slot) {
        int t = tile;
        const int tw = t % tiles_w; t /= tiles_w;
        const int th = t % tiles_h;
        const int b = t / tiles_h;
        const int h0 = 2 * th * SS_T - 1, w0 = 2 * tw * SS_T - 1;
        unsigned char* const dst = Xs + slot * XB;
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            const int ii = wave + j * SS_NW;
            const int s = ii * 64 + lane;
            const int px = s / CPR, pc = s - px * CPR;
            const int c = pc ^ (px & 15);
            const int ry = px / SS_R, rx = px - ry * SS_R;
            const int hi = h0 + ry, wi = w0 + rx;
            const bool ok = tile < num_tiles && px < SS_RP && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            const unsigned voff = ok ? (unsigned)((((b * p.H + hi) * p.W + wi) * p.x_stride + p.x_coff + c * 8) * 2) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_void*)(dst + ii * 1024), 16, voff, 0, 0, 0);
        }
    };
    int tile = blockIdx.x;
    issue_tile(tile, 0);

    // ---- operands that stay in registers ------------------------------------------------------------------------------------------------
    // 1x1: this wave's 32 intermediate channels x K as NKS x 2 A fragments; k order inside a pair of substeps chosen so that a lane's two
    // loads are 32 contiguous bytes (conv_wrs.hip)
    bf16x8 wreg[NKS][2];
    float b1[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const unsigned voff = (unsigned)((((wave * 2 + i) * 16 + fr) * p.Kpad1 + (ks >> 1) * 64 + fc * 16 + (ks & 1) * 8) * 2);
            const __attribute__((ext_vector_type(4))) unsigned v = __builtin_amdgcn_raw_buffer_load_b128(wrs, voff, 0, 0);
            wreg[ks][i] = __builtin_bit_cast(bf16x8, v);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) b1[i][r] = p.bias1[(wave * 2 + i) * 16 + fc * 4 + r];
    }
    // depthwise: a thread = channel pair cp (2cp, 2cp + 1) x output row oy of the tile
    const int cp = tid & 127, oy = tid >> 7;
    f32x2 wd[9], bd;
    {
        const unsigned* w32 = (const unsigned*)p.wd;               // packed [9][C] bf16: pair cp is dword cp of a tap row
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const unsigned u = w32[t * (SS_C / 2) + cp];
            wd[t] = f32x2{__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)};
        }
        bd = f32x2{p.biasd[2 * cp], p.biasd[2 * cp + 1]};
    }
    // (known complete before the loop, then passed through empty asm statements: see conv_wres.hip)
    ss_wait_vm<0>();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) asm volatile("" : "+v"(wreg[ks][i]));
#pragma unroll
        for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(b1[i][r]));
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) asm volatile("" : "+v"(wd[t]));
    asm volatile("" : "+v"(bd));

    const unsigned ts_l = ss_lds_addr(Ts);
    for (int it = 0; tile < num_tiles; tile += G, ++it) {
        // (a) this tile's patch has landed (issued in front of the previous tile's stores, which may still fly)
        if (it == 0) ss_wait_vm<0>();
        else ss_wait_vm<NST>();
        __builtin_amdgcn_s_barrier();
        const int slot = it & 1;
        issue_tile(tile + G, slot ^ 1);
        int t = tile;
        const int tw = t % tiles_w; t /= tiles_w;
        const int th = t % tiles_h;
        const int b = t / tiles_h;
        const int h0 = 2 * th * SS_T - 1, w0 = 2 * tw * SS_T - 1;
        // ---- 1x1 GEMM on the patch: this wave's 32 channels x 6 pixel fragments ---------------------------------------------------------
        // (one 16-channel fragment at a time: both at once need 48 accumulator registers beside the 64 weight registers of K = 256 and spill)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            f32x4 acc[SS_FM];
#pragma unroll
            for (int f = 0; f < SS_FM; ++f) acc[f] = f32x4{b1[i][0], b1[i][1], b1[i][2], b1[i][3]};
            const unsigned char* const X = Xs + slot * XB;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const int ch = (ks >> 1) * 8 + fc * 2 + (ks & 1);  // the chunk that holds this lane's 8 channels of the substep (see the weights)
                bf16x8 xf[SS_FM];
#pragma unroll
                for (int f = 0; f < SS_FM; ++f) xf[f] = *(const bf16x8*)(X + (f * 16 + fr) * RB + ((ch ^ fr) << 4));
#pragma unroll
                for (int f = 0; f < SS_FM; ++f) acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[ks][i], xf[f], acc[f], 0, 0, 0);
            }
            const int co = (wave * 2 + i) * 16 + fc * 4;           // chunk co >> 3 of the row, bytes (co & 7) * 2 ..
#pragma unroll
            for (int f = 0; f < SS_FM; ++f) {
                const int px = f * 16 + fr;
                const int ry = px / SS_R, rx = px - ry * SS_R;
                const bool in = px < SS_RP && (unsigned)(h0 + ry) < (unsigned)p.H && (unsigned)(w0 + rx) < (unsigned)p.W;
                float v[4] = {acc[f][0], acc[f][1], acc[f][2], acc[f][3]};
                if (p.act1 == ACT_SILU) silu4_packed(v);
                __attribute__((aligned(8))) __bf16 o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (__bf16)(in ? v[r] : 0.f);
                const unsigned dst = ts_l + (unsigned)(px * 512 + ((((co >> 3) & 15) ^ fr) << 4) + ((co >> 7) << 8) + (co & 7) * 2);
                asm volatile("ds_write_b64 %0, %1" : : "v"(dst), "v"(*(const unsigned long long*)o) : "memory");
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                              // (b) t is complete
        // ---- depthwise 3x3 stride 2: output row oy, channel pair cp, four outputs --------------------------------------------------------
        {
            auto rd = [&](int ky, int col) -> f32x2 {
                const int px = (2 * oy + ky) * SS_R + col;
                const unsigned u = *(const unsigned*)(Ts + px * 512 + ((((cp >> 2) & 15) ^ (px & 15)) << 4) + ((cp >> 6) << 8) + (cp & 3) * 4);
                return f32x2{__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)};
            };
            f32x2 win[3][3];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) win[ky][0] = rd(ky, 0);
            const int ho = th * SS_T + oy;
#pragma unroll
            for (int x = 0; x < SS_T; ++x) {
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) { win[ky][1] = rd(ky, 2 * x + 1); win[ky][2] = rd(ky, 2 * x + 2); }
                f32x2 a = bd;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) a = __builtin_elementwise_fma(win[ky][kx], wd[ky * 3 + kx], a);
                if (p.actd == ACT_SILU) {
                    f32x2 e = a * -1.4426950408889634f;
                    e[0] = __builtin_amdgcn_exp2f(e[0]); e[1] = __builtin_amdgcn_exp2f(e[1]);
                    e = e + 1.0f;
                    e[0] = __builtin_amdgcn_rcpf(e[0]); e[1] = __builtin_amdgcn_rcpf(e[1]);
                    a = a * e;
                }
                __attribute__((aligned(4))) __bf16 o[2] = {(__bf16)a[0], (__bf16)a[1]};
                const int wo = tw * SS_T + x;
                const bool ok = ho < p.Ho && wo < p.Wo;
                const unsigned off = ok ? (unsigned)((((b * p.Ho + ho) * p.Wo + wo) * p.y_stride + p.y_coff + 2 * cp) * 2) : OOB;
                __builtin_amdgcn_raw_buffer_store_b32(*(const unsigned*)o, yrs, off, 0, 0);
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) win[ky][0] = win[ky][2];
            }
        }
    }
    ss_wait_vm<0>();
}

bool scdown_stream_valid(const ScdParams& p) {
    static const bool off = [] { const char* v = std::getenv("YOLOP_NO_SCD_STREAM"); return v && *v == '1'; }();   // A/B switch
    if (off || p.clk) return false;
    // K = 128 (`model.5`: 59.5 us stand-alone against 66-67.5, same-box A/B of the step -9 us). K = 256 (`model.20`: 32.4 us against 22.4 + 12.7
    // for the two kernels it replaces) costs a lone batch 7 us per step and GAINS 12 us with two batches in flight (1.4302 against 1.4422 ms,
    // three tunings each: one launch holds the CUs for less time than two) - on, since the ring is how batches are run. YOLOP_SCD_STREAM_K=128 / 256: one width only.
    static const int only_k = [] { const char* v = std::getenv("YOLOP_SCD_STREAM_K"); return v ? atoi(v) : 0; }();
    if (only_k && p.K != only_k) return false;
    if ((p.K != 128 && p.K != 256) || p.Kpad1 != p.K || p.C != SS_C) return false;
    if ((p.x_stride & 7) || (p.x_coff & 7) || (p.y_stride & 1) || (p.y_coff & 1)) return false;
    if (p.x_bytes >= (1ull << 31) || p.y_bytes >= (1ull << 31) || p.w1_bytes >= (1ull << 31)) return false;
    if (p.w1_bytes < (size_t)p.C * p.Kpad1 * 2) return false;
    if (p.act1 != ACT_SILU && p.act1 != ACT_NONE) return false;
    if (p.actd != ACT_SILU && p.actd != ACT_NONE) return false;
    if ((p.H & 1) || (p.W & 1) || p.Ho * 2 != p.H || p.Wo * 2 != p.W) return false;
    const long covered = (long)((p.Ho + SS_T - 1) / SS_T * SS_T) * ((p.Wo + SS_T - 1) / SS_T * SS_T);
    return covered * 2 <= (long)p.Ho * p.Wo * 3;                       // (tiny maps: the separate kernels waste less)
}

const char* scdown_stream_kernel_name(const ScdParams& p) { return p.K == 128 ? "scdown_fused_kernel<stream,4>" : "scdown_fused_kernel<stream,8>"; }

template <int NKS>
static hipError_t launch_scdown_stream_t(const ScdParams& p, hipStream_t st) {
    const size_t sh = (size_t)2 * SS_FM * 16 * NKS * 64 + SS_TB;
    auto kern = scdown_stream_kernel<NKS>;
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr = true;
    }
    const int tiles_h = (p.Ho + SS_T - 1) / SS_T, tiles_w = (p.Wo + SS_T - 1) / SS_T;
    const int num_tiles = p.B * tiles_h * tiles_w;
    const int G = num_tiles < 256 ? num_tiles : 256;
    hipLaunchKernelGGL(kern, dim3((unsigned)G), dim3(SS_NW * 64), sh, st, p, tiles_h, tiles_w, num_tiles, G);
    return hipGetLastError();
}

hipError_t launch_scdown_stream(const ScdParams& p, hipStream_t st) {
    if (!scdown_stream_valid(p)) return hipErrorInvalidValue;
    return p.K == 128 ? launch_scdown_stream_t<4>(p, st) : launch_scdown_stream_t<8>(p, st);
}

}  // namespace yp
