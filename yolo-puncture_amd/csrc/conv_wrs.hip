// Weights-in-registers streaming form of the bf16 1x1 convolution (configuration ids 1200+): y[px][co] = act(W . x[px] + b).
// The 128- / 256- / 512-wide 1x1 layers of C2f and SCDown on the 40x40 and 80x80 maps (SURVEY.md Appendix A.2-A.3 [U]; run inside
// `.predict`, reference yolo_seg/app.py:91).
//
// Why: `conv_wres_kernel` keeps a 128-channel weight block in LDS; a 256-wide layer then needs two
// workgroups per pixel tile and every pixel row is fetched twice (measured: the 256 -> 256 layers at 40x40 stay at 2.3 TB/s algorithmic).
// A CU's register file is 512 KB: the WHOLE weight matrix of these layers (64-196 KB) fits in it, spread over the 8 waves of one
// workgroup - wave w owns output channels [w * 16 * NFW, (w + 1) * 16 * NFW) for all K as NKS x NFW MFMA fragments (<= 128 VGPRs), loaded
// once per workgroup. LDS then holds nothing but pixel tiles ([TP px][K], whole rows by LDS-DMA, double-buffered, ONE barrier per tile);
// every wave multiplies every pixel fragment of the tile by its own channels. No pixel row is read twice, no weight byte moves after
// the prologue, the k loop is fully unrolled (register-indexed weights) and the epilogue of a tile runs under the next tile's loads.
// A folded nearest-x2 upsample is two row segments per pixel, each from its own tensor (as conv_wres.hip).
#include "common.h"
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace yp {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void lds_void;

constexpr int WS_NW = 8;

template <int N> __device__ __forceinline__ void ws_wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt range");
    __builtin_amdgcn_s_waitcnt((N & 0xF) | (7 << 4) | (0xF << 8) | (((N >> 4) & 3) << 14));
}

template <int NKS, int NFW, int TP>
__global__ __launch_bounds__(WS_NW * 64) void conv_wrs_kernel(const ConvParams p, const int G) {
    constexpr int FM = TP / 16;                                    // pixel fragments of a tile: every wave takes all of them
    constexpr int K = NKS * 32;
    constexpr int RBW = K * 2;
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int KA = p.x2_C, KB = K - KA;                            // KA channels from the low-resolution source (0: none), KB from x
    const int RBA = KA * 2, RBB = KB * 2;
    const int ksA = KA >> 5;                                       // substeps of segment A
    // swizzle: position pc of row r holds source chunk pc ^ (r & mask); mask = 15 where a row is a multiple of 256 bytes, else 7
    const int mA = (KA % 128) == 0 ? 15 : 7, mB = (KB % 128) == 0 ? 15 : 7;
    constexpr size_t xtile = (size_t)TP * RBW;                     // one pixel tile: [TP][KA] then [TP][KB]
    unsigned char* const Xs = smem;                                // 2 tiles
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fc = lane >> 4;
    const int ntiles = (p.M + TP - 1) / TP;
    const int HoWo = p.Ho * p.Wo;
    const int n0 = wave * 16 * NFW;

    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t x2rs = __builtin_amdgcn_make_buffer_rsrc((void*)(KA > 0 ? p.x2 : p.x), 0, (int)(KA > 0 ? p.x2_bytes : p.x_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, (int)p.y_bytes, 0x00020000);

    auto issue_tile = [&](int tile, unsigned char* dst) {
        const long row0 = (long)tile * TP;
        if (KA > 0) {                                              // segment A: channels [0, KA) of pixel (b, ho, wo) live at (b, ho >> 1, wo >> 1) of x2
            const int cpr = RBA >> 4, pieces = (TP * cpr) >> 6;
            for (int ii = wave; ii < pieces; ii += WS_NW) {
                const int s = ii * 64 + lane;
                const int r = s / cpr, pc = s - r * cpr;
                const int c = pc ^ (r & mA);
                const long m = row0 + r;
                unsigned voff = OOB;
                if (m < p.M) {
                    const int mi = (int)m;
                    const int b = mi / HoWo, q = mi - b * HoWo;
                    const int ho = q / p.Wo, wo = q - ho * p.Wo;
                    voff = (unsigned)((((b * p.x2_H + (ho >> 1)) * p.x2_W + (wo >> 1)) * p.x2_stride + p.x2_coff + c * 8) * 2);
                }
                __builtin_amdgcn_raw_ptr_buffer_load_lds(x2rs, (lds_void*)(dst + ii * 1024), 16, voff, 0, 0, 0);
            }
        }
        if (KA == 0) {                                             // the usual case: every division below by a constant
            constexpr int cpr = K / 8, pieces = (TP * cpr) >> 6, mK = (K % 128) == 0 ? 15 : 7;
#pragma unroll
            for (int j = 0; j < (pieces + WS_NW - 1) / WS_NW; ++j) {
                const int ii = wave + j * WS_NW;
                if (ii >= pieces) break;
                const int s = ii * 64 + lane;
                const int r = s / cpr, pc = s - r * cpr;
                const int c = pc ^ (r & mK);
                const long m = row0 + r;
                const unsigned voff = (m < p.M) ? (unsigned)((m * p.x_stride + p.x_coff + c * 8) * 2) : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_void*)(dst + ii * 1024), 16, voff, 0, 0, 0);
            }
        } else {
            unsigned char* const d2 = dst + (size_t)TP * RBA;
            const int cpr = RBB >> 4, pieces = (TP * cpr) >> 6;
            for (int ii = wave; ii < pieces; ii += WS_NW) {
                const int s = ii * 64 + lane;
                const int r = s / cpr, pc = s - r * cpr;
                const int c = pc ^ (r & mB);
                const long m = row0 + r;
                const unsigned voff = (m < p.M) ? (unsigned)((m * p.x_stride + p.x_coff + KA + c * 8) * 2) : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_void*)(d2 + ii * 1024), 16, voff, 0, 0, 0);
            }
        }
    };
    unsigned long long clk[5] = {0, 0, 0, 0, 0};                   // debug (YOLOP_WRS_CLOCKS=1): prologue, wait + barrier, issue, k loop, epilogue
    unsigned long long last = p.clk ? __builtin_amdgcn_s_memtime() : 0ull;
#define WS_STAMP(i) if (p.clk) { const unsigned long long now = __builtin_amdgcn_s_memtime(); clk[i] += now - last; last = now; }
    int tile = blockIdx.x;
    if (tile < ntiles) issue_tile(tile, Xs);

    // this wave's weights, in MFMA fragment layout: lane (fr, fc) of fragment (ks, i) holds 8 channels of row W[n0 + i * 16 + fr]
    // (rows beyond Cout: the packed matrix's zero rows / beyond its end: zeros)
    bf16x8 wreg[NKS][NFW];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
        for (int i = 0; i < NFW; ++i) {
            // k order inside a PAIR of substeps: lane group fc takes the 16 channels [fc * 16, fc * 16 + 16) of the pair's 64, the first 8
            // in the even substep, the last 8 in the odd one - its two loads are 32 contiguous bytes and the four groups of a weight row
            // cover a whole 128-byte line (the plain order, 16 bytes of every other 64: half-used lines, a third more prologue)
            const unsigned voff = (unsigned)(((n0 + i * 16 + fr) * p.Kpad + (ks >> 1) * 64 + fc * 16 + (ks & 1) * 8) * 2);
            const __attribute__((ext_vector_type(4))) unsigned v = __builtin_amdgcn_raw_buffer_load_b128(wrs, voff, 0, 0);
            wreg[ks][i] = __builtin_bit_cast(bf16x8, v);
        }
    float bias[NFW][4];
#pragma unroll
    for (int i = 0; i < NFW; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = n0 + i * 16 + fc * 4 + r;
            bias[i][r] = (co < p.Cout) ? p.bias[co] : 0.f;
        }
    // Weights and bias must be KNOWN to be complete before the loop (the compiler cannot count a loop iteration's vector-memory
    // operations and would wait `vmcnt(0)` in front of their first use in every iteration - behind the next tile's loads): waited for
    // here, then passed through empty asm statements (see conv_wres.hip).
    ws_wait_vm<0>();
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
        for (int i = 0; i < NFW; ++i) asm volatile("" : "+v"(wreg[ks][i]));
#pragma unroll
    for (int i = 0; i < NFW; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(bias[i][r]));

    WS_STAMP(0)
    for (int it = 0; tile < ntiles; tile += G, ++it) {
        // this tile's rows have landed; the NFW * FM stores of the previous tile, issued behind them, may still fly
        if (it == 0) ws_wait_vm<0>();
        else ws_wait_vm<NFW * FM>();
        __builtin_amdgcn_s_barrier();
        WS_STAMP(1)
        if (tile + G < ntiles) issue_tile(tile + G, Xs + ((it & 1) ^ 1) * xtile);
        WS_STAMP(2)
        const unsigned char* const XA = Xs + (it & 1) * xtile;
        const unsigned char* const XB = XA + (size_t)TP * RBA;
        f32x4 acc[NFW][FM];
#pragma unroll
        for (int i = 0; i < NFW; ++i)
#pragma unroll
            for (int f = 0; f < FM; ++f) acc[i][f] = f32x4{bias[i][0], bias[i][1], bias[i][2], bias[i][3]};
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const bool inA = ks < ksA;                                             // (uniform)
            const unsigned char* const X = inA ? XA : XB;
            const int rb = inA ? RBA : RBB;
            const int kl = inA ? ks : ks - ksA;                                    // (segments are whole pairs of substeps)
            const int ch = (kl >> 1) * 8 + fc * 2 + (kl & 1);                      // the chunk that holds this lane's 8 channels of the substep (see the weights)
            const int pc = ch ^ (fr & (inA ? mA : mB));                            // (row & 15 = fr)
            bf16x8 xf[FM];
#pragma unroll
            for (int f = 0; f < FM; ++f) xf[f] = *(const bf16x8*)(X + (size_t)(f * 16 + fr) * rb + pc * 16);
#pragma unroll
            for (int i = 0; i < NFW; ++i)
#pragma unroll
                for (int f = 0; f < FM; ++f) acc[i][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[ks][i], xf[f], acc[i][f], 0, 0, 0);
        }
        WS_STAMP(3)
        // ---- activation, bf16 stores (a fixed number per wave and tile: masked ones go out of range) ----------------------------------
#pragma unroll
        for (int f = 0; f < FM; ++f) {
            const long m = (long)tile * TP + f * 16 + fr;
#pragma unroll
            for (int i = 0; i < NFW; ++i) {
                const int co = n0 + i * 16 + fc * 4;
                const bool ok = m < p.M && co < p.Cout;                            // (Cout % 4 == 0: a lane's four channels exist together)
                float v[4] = {acc[i][f][0], acc[i][f][1], acc[i][f][2], acc[i][f][3]};
                if (p.act == ACT_SILU) silu4_packed(v);
                __attribute__((aligned(8))) __bf16 o[4] = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                const unsigned off = ok ? (unsigned)((m * p.y_stride + p.y_coff + co) * 2) : OOB;
                __builtin_amdgcn_raw_buffer_store_b64(*(const __attribute__((ext_vector_type(2))) unsigned*)o, yrs, off, 0, 0);
            }
        }
        WS_STAMP(4)
    }
    ws_wait_vm<0>();
    if (p.clk && lane == 0)
        for (int i = 0; i < 5; ++i) p.clk[((size_t)blockIdx.x * WS_NW + wave) * 5 + i] = clk[i];
#undef WS_STAMP
}

// ---------------------------------------------------------------------------------------------------------------------------------------
struct WrsCfg { int TP; const char* name; };
static const WrsCfg kWrs[] = {
    {64, "conv_wrs_kernel<64>"},
    {32, "conv_wrs_kernel<32>"},
};
static const int kNumWrs = (int)(sizeof(kWrs) / sizeof(kWrs[0]));

int conv_wrs_num_cfgs() { return kNumWrs; }
const char* conv_wrs_kernel_name(int c) { return kWrs[c].name; }

// (K / 32, channel fragments per wave) pairs this build instantiates: NKS * NFW * 4 weight registers per lane, 128 at most
static bool wrs_shape(const ConvParams& p, int& nks, int& nfw) {
    if ((p.Cin % 64) != 0) return false;
    nks = p.Cin / 32;
    nfw = (p.Cout + 127) / 128;                                    // 8 waves x 16 channels x nfw >= Cout
    if (nfw == 3) nfw = 4;
    static const int ok[][2] = {{4, 1}, {6, 1}, {8, 1}, {12, 1}, {8, 2}, {12, 2}, {16, 2}, {8, 4}};
    for (auto& s : ok)
        if (s[0] == nks && s[1] == nfw) return true;
    return false;
}

bool conv_wrs_cfg_valid(const ConvParams& p, int c) {
    if (c < 0 || c >= kNumWrs) return false;
    const WrsCfg& k = kWrs[c];
    int nks, nfw;
    if (p.ks != 1 || p.stride != 1 || p.up != 1 || p.w2 || p.out_f32 || p.pool_in || p.up_bilinear || p.res) return false;
    if (p.act != ACT_SILU && p.act != ACT_NONE) return false;
    if (p.Kpad != p.Cin || !wrs_shape(p, nks, nfw)) return false;
    if (nks * nfw * 4 >= 128 && k.TP > 32) return false;          // (128 weight registers + the 64-pixel tile's accumulators and fragments spill)
    if (p.Cout <= 64 * nfw) return false;                          // (more than half of the waves' channels would be padding)
    if (p.x2_C > 0 && ((p.x2_C % 64) != 0 || p.x2_C >= p.Cin || (p.x2_stride & 7) || (p.x2_coff & 7) || p.x2_bytes >= (1ull << 31))) return false;
    if ((p.x_stride & 7) || (p.x_coff & 7)) return false;
    if ((p.Cout & 3) || (p.y_stride & 3) || (p.y_coff & 3)) return false;
    if (p.x_bytes >= (1ull << 31) || p.w_bytes >= (1ull << 31) || p.y_bytes >= (1ull << 31) || p.M <= 0 || p.Ho <= 0 || p.Wo <= 0) return false;
    if ((((size_t)k.TP * (p.Cin - p.x2_C) * 2) & 1023) || (((size_t)k.TP * p.x2_C * 2) & 1023)) return false;       // whole 1-KiB pieces per segment
    return (size_t)2 * k.TP * p.Cin * 2 <= 156 * 1024;
}

template <int NKS, int NFW, int TP>
static hipError_t launch_wrs_t(const ConvParams& p, hipStream_t st) {
    const size_t sh = (size_t)2 * TP * p.Cin * 2;
    auto kern = conv_wrs_kernel<NKS, NFW, TP>;
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr = true;
    }
    const int ntiles = (p.M + TP - 1) / TP;
    const int G = ntiles < 256 ? ntiles : 256;                     // one workgroup per CU (its waves hold the weights: 2 per SIMD)
    static const bool clocks = [] { const char* v = std::getenv("YOLOP_WRS_CLOCKS"); return v && *v == '1'; }();   // debug: per-phase s_memtime sums
    if (clocks) {
        ConvParams q = p;
        const size_t n = (size_t)G * WS_NW * 5;
        if (hipMalloc((void**)&q.clk, n * 8) != hipSuccess) return hipErrorOutOfMemory;
        hipLaunchKernelGGL(kern, dim3((unsigned)G), dim3(WS_NW * 64), sh, st, q, G);
        (void)hipStreamSynchronize(st);
        std::vector<unsigned long long> h(n);
        (void)hipMemcpy(h.data(), q.clk, n * 8, hipMemcpyDeviceToHost);
        (void)hipFree(q.clk);
        static const char* nm[5] = {"prologue", "wait+barrier", "issue", "k loop", "epilogue"};
        for (int w = 0; w < WS_NW; w += WS_NW - 1) {
            fprintf(stderr, "[wrs clocks] K=%d Cout=%d M=%d TP=%d tiles/wg %.2f wave %d, s_memtime ticks per workgroup:", p.Cin, p.Cout, p.M, TP, (double)ntiles / G, w);
            for (int i = 0; i < 5; ++i) {
                double s = 0;
                for (int g = 0; g < G; ++g) s += (double)h[((size_t)g * WS_NW + w) * 5 + i];
                fprintf(stderr, " %s %.0f", nm[i], s / G);
            }
            fprintf(stderr, "\n");
        }
        return hipGetLastError();
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)G), dim3(WS_NW * 64), sh, st, p, G);
    return hipGetLastError();
}

template <int TP>
static hipError_t launch_wrs_tp(const ConvParams& p, int nks, int nfw, hipStream_t st) {
    if (nfw == 1) switch (nks) {
        case 4: return launch_wrs_t<4, 1, TP>(p, st);
        case 6: return launch_wrs_t<6, 1, TP>(p, st);
        case 8: return launch_wrs_t<8, 1, TP>(p, st);
        case 12: return launch_wrs_t<12, 1, TP>(p, st);
    }
    if (nfw == 2) switch (nks) {
        case 8: return launch_wrs_t<8, 2, TP>(p, st);
        case 12: return launch_wrs_t<12, 2, TP>(p, st);
        case 16: return launch_wrs_t<16, 2, TP>(p, st);
    }
    if (nfw == 4 && nks == 8) return launch_wrs_t<8, 4, TP>(p, st);
    return hipErrorInvalidValue;
}

hipError_t launch_conv_wrs(const ConvParams& p, int c, hipStream_t st) {
    if (!conv_wrs_cfg_valid(p, c)) return hipErrorInvalidValue;
    int nks, nfw;
    wrs_shape(p, nks, nfw);
    return kWrs[c].TP == 64 ? launch_wrs_tp<64>(p, nks, nfw, st) : launch_wrs_tp<32>(p, nks, nfw, st);
}

}  // namespace yp
