// Weights-resident streaming form of the bf16 1x1 convolution (configuration ids 1100+): y[px][co] = act(W . x[px] + b) (+ res).
// The 1x1 layers of C2f / SCDown / PSA / the heads (SURVEY.md Appendix A.2-A.4 [U]; run inside `.predict`, reference yolo_seg/app.py:91).
//
// Why: the 1x1 layers on the 80x80 and 40x40 maps are HBM-bound (AI 64-150 FLOP/B) and the k-stepped GEMM
// kernels run them at 2.5-3.6 TB/s: a k-step is a barrier, a ring slot and a handful of 1-KiB pieces per wave, and the pixel rows arrive
// 128 bytes at a time. `cls_out_kernel` (the logits layer, same idea) reaches 4.1 TB/s = 89 % of the chip's copy rate. Here, generalised:
//   * a persistent workgroup keeps ITS block of output channels' weights [NB][K] in LDS for its whole life (NB <= 128; wider layers are
//     split into blocks whose workgroups sit on the same XCD and walk the same pixel tiles at the same time: the second read of a tile is
//     an L2 hit);
//   * a pixel tile is [TP px][K] - whole pixel rows, K * 2 contiguous bytes each - double-buffered by LDS-DMA: ONE barrier per tile, the
//     next tile's rows in flight under this tile's MFMAs, SiLU and stores;
//   * a folded nearest-x2 upsample (1x1 on a [upsampled | skip] concat) is two row segments per pixel, each from its own tensor.
// 8 waves as WGM (pixels) x WGN (channel fragments); a lane ends with 4 consecutive channels of a pixel: bias rides in the accumulator,
// activation, optional residual, bf16, one 8-byte buffer store. A wave issues the same number of stores for every tile (masked ones go
// to an out-of-range offset), so the wait in front of the next tile is a counted `vmcnt`: the rows must have landed, the stores issued
// behind them may still be in flight (first form: `vmcnt(0)` there and four waves - 77 us on `model.4.cv2`, every store drain exposed).
#include "common.h"

namespace yp {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void lds_void;

constexpr int WR_NW = 8;
constexpr int WR_MAXNF = 8;             // channel fragments of a block (NB <= 128)

template <int N> __device__ __forceinline__ void wr_wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt range");
    __builtin_amdgcn_s_waitcnt((N & 0xF) | (7 << 4) | (0xF << 8) | (((N >> 4) & 3) << 14));
}

// the fragment set (xs, ws) has landed once at most N younger LDS reads are outstanding; its registers pass through empty asm statements
// behind the wait, so no use of them moves above it (volatile asm statements keep their order)
template <int N, int FM, int NFW> __device__ __forceinline__ void wr_wait_lgkm(bf16x8 (&xs)[FM], bf16x8 (&ws)[NFW]) {
    static_assert(N >= 0 && N < 16, "lgkmcnt range");
    asm volatile("s_waitcnt lgkmcnt(%0)" : : "n"(N) : "memory");
#pragma unroll
    for (int f = 0; f < FM; ++f) asm volatile("" : "+v"(xs[f]));
#pragma unroll
    for (int i = 0; i < NFW; ++i) asm volatile("" : "+v"(ws[i]));
}

template <int TP, int WGN, bool HAS_RES>
__global__ __launch_bounds__(WR_NW * 64) void conv_wres_kernel(const ConvParams p, const int NB, const int nblk, const int G, const int TPe) {
    constexpr int WGM = WR_NW / WGN;
    constexpr int FM = TP / (16 * WGM);                            // pixel fragments per wave
    constexpr int NFW = WR_MAXNF / WGN;                            // channel fragments per wave at most
    static_assert(FM >= 1 && FM * 16 * WGM == TP, "pixel tile");
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int K = p.Cin, KA = p.x2_C, KB = K - KA;                 // KA channels from the low-resolution source (0: none), KB from x
    const int RBW = K * 2, RBA = KA * 2, RBB = KB * 2;             // row bytes: weights, segment A, segment B
    // swizzle: position pc of row r holds source chunk pc ^ (r & mask); mask = 15 where a row is a multiple of 256 bytes (a fragment read's 16
    // rows then fall into the 16 different 16-byte bank groups), else 7 (8 groups, two rows each)
    const int mA = (KA % 128) == 0 ? 15 : 7, mB = (KB % 128) == 0 ? 15 : 7, mW = ((K % 128) == 0 && (KA % 128) == 0) ? 15 : 7;
    const int nfb = (NB + 15) >> 4;                                // channel fragments of the block
    const int nfw = nfb / WGN;                                     // ... of a wave (nfb % WGN == 0)
    unsigned char* const Ws = smem;                                // [nfb * 16][K]
    const size_t xtile = (size_t)TP * RBW;                         // one pixel tile: [TP][KA] then [TP][KB]
    unsigned char* const Xs = smem + (size_t)nfb * 16 * RBW;       // 2 tiles
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WGM, wn = wave / WGM;
    const int fr = lane & 15, fc = lane >> 4;
    // TPe <= TP rows of a tile are in use: the host sizes the tiles so that every tile lane gets the same number of pixels (a 40x40 map in
    // 64-pixel tiles is 3.1 tiles per lane = four rounds of which the last is mostly idle; four rounds of 50 pixels are not)
    const int ntiles = (p.M + TPe - 1) / TPe;
    const int HoWo = p.Ho * p.Wo;

    // workgroup -> (channel block, tile lane): the blocks of one tile lane are neighbours on one XCD (G % (8 * nblk) == 0)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int nb = slot % nblk, g = (slot / nblk) * 8 + xcd;
    const int Gt = G / nblk;                                       // tile lanes
    const int n0 = nb * NB;

    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t x2rs = __builtin_amdgcn_make_buffer_rsrc((void*)(KA > 0 ? p.x2 : p.x), 0, (int)(KA > 0 ? p.x2_bytes : p.x_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, (int)p.y_bytes, 0x00020000);

    // rows are written 16 bytes per lane, whole rows per piece, swizzled as above
    auto issue_tile = [&](int tile, unsigned char* dst) {
        const long row0 = (long)tile * TPe;
        if (KA > 0) {                                              // segment A: channels [0, KA) of pixel (b, ho, wo) live at (b, ho >> 1, wo >> 1) of x2
            const int cpr = RBA >> 4, pieces = (TP * cpr) >> 6;
            for (int ii = wave; ii < pieces; ii += WR_NW) {
                const int s = ii * 64 + lane;
                const int r = s / cpr, pc = s - r * cpr;
                const int c = pc ^ (r & mA);
                const long m = row0 + r;
                unsigned voff = OOB;
                if (m < p.M && r < TPe) {
                    const int mi = (int)m;
                    const int b = mi / HoWo, q = mi - b * HoWo;
                    const int ho = q / p.Wo, wo = q - ho * p.Wo;
                    voff = (unsigned)((((b * p.x2_H + (ho >> 1)) * p.x2_W + (wo >> 1)) * p.x2_stride + p.x2_coff + c * 8) * 2);
                }
                __builtin_amdgcn_raw_ptr_buffer_load_lds(x2rs, (lds_void*)(dst + ii * 1024), 16, voff, 0, 0, 0);
            }
        }
        {
            unsigned char* const d2 = dst + (size_t)TP * RBA;
            const int cpr = RBB >> 4, pieces = (TP * cpr) >> 6;
            for (int ii = wave; ii < pieces; ii += WR_NW) {
                const int s = ii * 64 + lane;
                const int r = s / cpr, pc = s - r * cpr;
                const int c = pc ^ (r & mB);
                const long m = row0 + r;
                const unsigned voff = (m < p.M && r < TPe) ? (unsigned)((m * p.x_stride + p.x_coff + KA + c * 8) * 2) : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_void*)(d2 + ii * 1024), 16, voff, 0, 0, 0);
            }
        }
    };
    {                                                              // the block's weights: rows n0 .. n0 + nfb * 16 (beyond Cout: the packed matrix's zero rows / beyond its end: zeros)
        const int cpr = RBW >> 4, pieces = (nfb * 16 * cpr) >> 6;
        for (int ii = wave; ii < pieces; ii += WR_NW) {
            const int s = ii * 64 + lane;
            const int r = s / cpr, pc = s - r * cpr;
            const int c = pc ^ (r & mW);
            const unsigned voff = (unsigned)(((n0 + r) * p.Kpad + c * 8) * 2);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_void*)(Ws + ii * 1024), 16, voff, 0, 0, 0);
        }
    }
    int tile = g;
    if (tile < ntiles) issue_tile(tile, Xs);

    float bias[NFW][4];
#pragma unroll
    for (int i = 0; i < NFW; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = n0 + (wn * nfw + i) * 16 + fc * 4 + r;
            bias[i][r] = (i < nfw && co < p.Cout) ? p.bias[co] : 0.f;
        }
    // The bias loads must be KNOWN to be complete before the loop: the compiler cannot count the vector-memory operations of a loop
    // iteration (runtime piece counts) and otherwise puts `s_waitcnt vmcnt(0)` in front of the accumulators' initialisation - in every
    // iteration, behind the next tile's loads: no prefetch at all. Waited for here, then passed through an empty asm statement (a value
    // that comes out of one is no memory result to wait for).
    wr_wait_vm<0>();
#pragma unroll
    for (int i = 0; i < NFW; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(bias[i][r]));

    for (int it = 0; tile < ntiles; tile += Gt, ++it) {
        // this tile's rows (first time: the weights too) have landed; the FM * NFW stores of the previous tile, issued behind them, may still fly
        if (it == 0) wr_wait_vm<0>();
        else wr_wait_vm<FM * NFW>();
        __builtin_amdgcn_s_barrier();
        if (tile + Gt < ntiles) issue_tile(tile + Gt, Xs + ((it & 1) ^ 1) * xtile);
        const unsigned char* const XA = Xs + (it & 1) * xtile;
        const unsigned char* const XB = XA + (size_t)TP * RBA;
        f32x4 acc[NFW][FM];
#pragma unroll
        for (int i = 0; i < NFW; ++i)
#pragma unroll
            for (int f = 0; f < FM; ++f) acc[i][f] = f32x4{bias[i][0], bias[i][1], bias[i][2], bias[i][3]};
        // One segment's k loop, software-pipelined by hand: the fragment reads of substep ks + 1 are issued in front of the MFMAs of ks
        // (two register sets, inline-asm `ds_read_b128`, counted `lgkmcnt` waits whose statement the set's registers pass through - the
        // compiler's own schedule was read, wait, MFMA, read, wait, MFMA: four exposed LDS round trips per substep, 9.8 k cycles per
        // 64-pixel tile on `model.6.cv1`). nks is even (segments are multiples of 64 channels).
        auto run_segment = [&](const unsigned char* X, int rb, int kchunk0, int nks, int mX) {
            unsigned xrow[FM], wrow[NFW];
            const unsigned xl = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)X;
            const unsigned wl = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)Ws;
#pragma unroll
            for (int f = 0; f < FM; ++f) xrow[f] = xl + (unsigned)(((wm * FM + f) * 16 + fr) * rb);
#pragma unroll
            for (int i = 0; i < NFW; ++i) wrow[i] = wl + (unsigned)(((wn * nfw + min(i, nfw - 1)) * 16 + fr) * RBW + kchunk0 * 16);   // (fewer than NFW fragments: the last one again, multiplied for nothing)
            const unsigned swx = (unsigned)(fr & mX), sww = (unsigned)(fr & mW);      // (row & 15 = fr for both operands' fragments)
            auto rd = [&](bf16x8 (&xd)[FM], bf16x8 (&wd)[NFW], int ks) {
                const unsigned ch = (unsigned)(ks * 4 + fc);
                const unsigned offx = (ch ^ swx) << 4, offw = (ch ^ sww) << 4;
#pragma unroll
                for (int f = 0; f < FM; ++f) asm volatile("ds_read_b128 %0, %1" : "=v"(xd[f]) : "v"(xrow[f] + offx));
#pragma unroll
                for (int i = 0; i < NFW; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(wd[i]) : "v"(wrow[i] + offw));
            };
            auto mma = [&](bf16x8 (&xd)[FM], bf16x8 (&wd)[NFW]) {
#pragma unroll
                for (int i = 0; i < NFW; ++i)
#pragma unroll
                    for (int f = 0; f < FM; ++f) acc[i][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wd[i], xd[f], acc[i][f], 0, 0, 0);
            };
            bf16x8 xa[FM], wa[NFW], xb[FM], wb[NFW];
            rd(xa, wa, 0);
            for (int ks = 0; ks < nks; ks += 2) {
                rd(xb, wb, ks + 1);
                wr_wait_lgkm<FM + NFW>(xa, wa);
                mma(xa, wa);
                if (ks + 2 < nks) {
                    rd(xa, wa, ks + 2);
                    wr_wait_lgkm<FM + NFW>(xb, wb);
                } else {
                    wr_wait_lgkm<0>(xb, wb);
                }
                mma(xb, wb);
            }
        };
        if (KA > 0) run_segment(XA, RBA, 0, KA >> 5, mA);
        run_segment(XB, RBB, KA >> 3, KB >> 5, mB);
        // ---- activation, residual, bf16 stores ----------------------------------------------------------------------------------
#pragma unroll
        for (int f = 0; f < FM; ++f) {
            const int ri = (wm * FM + f) * 16 + fr;
            const long m = (long)tile * TPe + ri;
#pragma unroll
            for (int i = 0; i < NFW; ++i) {
                const int co = n0 + (wn * nfw + i) * 16 + fc * 4;
                const bool ok = i < nfw && ri < TPe && m < p.M && co < p.Cout;                 // (Cout % 4 == 0: a lane's four channels exist together)
                float v[4] = {acc[i][f][0], acc[i][f][1], acc[i][f][2], acc[i][f][3]};
                if (p.act == ACT_SILU && i < nfw) silu4_packed(v);
                if (HAS_RES && ok) {
                    const uint2 rr = *(const uint2*)((const __bf16*)p.res + m * p.res_stride + p.res_coff + co);
                    v[0] += __uint_as_float(rr.x << 16); v[1] += __uint_as_float(rr.x & 0xffff0000u);
                    v[2] += __uint_as_float(rr.y << 16); v[3] += __uint_as_float(rr.y & 0xffff0000u);
                }
                __attribute__((aligned(8))) __bf16 o[4] = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                const unsigned off = ok ? (unsigned)((m * p.y_stride + p.y_coff + co) * 2) : OOB;
                __builtin_amdgcn_raw_buffer_store_b64(*(const __attribute__((ext_vector_type(2))) unsigned*)o, yrs, off, 0, 0);
            }
        }
    }
    wr_wait_vm<0>();
}

// ---------------------------------------------------------------------------------------------------------------------------------------
struct WresCfg { int TP, WGN; const char* name; };
static const WresCfg kWres[] = {
    {64, 2, "conv_wres_kernel<64,2>"},
    {128, 2, "conv_wres_kernel<128,2>"},
    {32, 4, "conv_wres_kernel<32,4>"},
};
static const int kNumWres = (int)(sizeof(kWres) / sizeof(kWres[0]));

int conv_wres_num_cfgs() { return kNumWres; }
const char* conv_wres_kernel_name(int c) { return kWres[c].name; }

// channel block of a workgroup: as few blocks as 128 channels each allow, equal sizes, whole fragments (pairs of them for two channel waves)
static void wres_blocks(const ConvParams& p, const WresCfg& k, int& NB, int& nblk) {
    nblk = (p.Cout + 127) / 128;
    const int q = 16 * k.WGN;
    NB = ((p.Cout + nblk - 1) / nblk + q - 1) / q * q;
}
static size_t wres_lds(const ConvParams& p, const WresCfg& k) {
    int NB, nblk;
    wres_blocks(p, k, NB, nblk);
    return (size_t)NB * p.Cin * 2 + (size_t)2 * k.TP * p.Cin * 2;
}

bool conv_wres_cfg_valid(const ConvParams& p, int c) {
    if (c < 0 || c >= kNumWres) return false;
    const WresCfg& k = kWres[c];
    if (p.ks != 1 || p.stride != 1 || p.up != 1 || p.w2 || p.out_f32 || p.pool_in || p.up_bilinear) return false;
    if (p.act != ACT_SILU && p.act != ACT_NONE) return false;
    if (p.Cin < 64 || (p.Cin % 64) != 0 || p.Kpad != p.Cin || p.Cin > 1024) return false;
    if (p.x2_C > 0 && ((p.x2_C % 64) != 0 || p.x2_C >= p.Cin || (p.x2_stride & 7) || (p.x2_coff & 7) || p.x2_bytes >= (1ull << 31))) return false;
    if ((p.x_stride & 7) || (p.x_coff & 7)) return false;
    if ((p.Cout & 3) || (p.y_stride & 3) || (p.y_coff & 3) || (p.res && ((p.res_stride & 3) || (p.res_coff & 3)))) return false;
    if (p.x_bytes >= (1ull << 31) || p.w_bytes >= (1ull << 31) || p.y_bytes >= (1ull << 31) || p.M <= 0 || p.Ho <= 0 || p.Wo <= 0) return false;
    if ((((size_t)k.TP * (p.Cin - p.x2_C) * 2) & 1023) || (((size_t)k.TP * p.x2_C * 2) & 1023)) return false;       // whole 1-KiB pieces per segment
    return wres_lds(p, k) <= 156 * 1024;
}

template <int TP, int WGN, bool HAS_RES>
static hipError_t launch_wres_t(const ConvParams& p, const WresCfg& k, hipStream_t st) {
    int NB, nblk;
    wres_blocks(p, k, NB, nblk);
    const size_t sh = wres_lds(p, k);
    auto kern = conv_wres_kernel<TP, WGN, HAS_RES>;
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr = true;
    }
    const int ntiles = (p.M + TP - 1) / TP;
    // an HBM-bound kernel: as many workgroups as fit (two per CU when the tiles are small), tile lanes in whole XCD rounds
    const int gmax = sh <= 78 * 1024 ? 512 : 256;
    int lanes8 = gmax / (8 * nblk);                                // tile lanes / 8
    if (lanes8 < 1) lanes8 = 1;
    while (lanes8 > 1 && (lanes8 - 1) * 8 >= ntiles) --lanes8;
    const int G = lanes8 * 8 * nblk;
    // equal pixels per tile lane: the rounds the full tiles need, then the tile height that fills exactly those rounds
    const long lanes = (long)lanes8 * 8;
    const int rounds = (int)((p.M + lanes * TP - 1) / (lanes * TP));
    const int TPe = tile_balance_enabled(2) ? (int)((p.M + lanes * rounds - 1) / (lanes * rounds)) : TP;
    hipLaunchKernelGGL(kern, dim3((unsigned)G), dim3(WR_NW * 64), sh, st, p, NB, nblk, G, TPe);
    return hipGetLastError();
}

hipError_t launch_conv_wres(const ConvParams& p, int c, hipStream_t st) {
    if (!conv_wres_cfg_valid(p, c)) return hipErrorInvalidValue;
    const WresCfg& k = kWres[c];
    const bool r = p.res != nullptr;
    switch (c) {
        case 0: return r ? launch_wres_t<64, 2, true>(p, k, st) : launch_wres_t<64, 2, false>(p, k, st);
        case 1: return r ? launch_wres_t<128, 2, true>(p, k, st) : launch_wres_t<128, 2, false>(p, k, st);
        default: return r ? launch_wres_t<32, 4, true>(p, k, st) : launch_wres_t<32, 4, false>(p, k, st);
    }
}

}  // namespace yp
