// Weights-in-registers form of the bf16 3x3 stride-1 convolution (configuration ids 700+).
//
// Why: PMC and in-kernel stamps of the halo kernels (profiles/r01_*) show the 3x3 layers at 500-650 TFLOP/s with the LDS array,
// not the matrix pipe, as the busy unit: with the weights in LDS a wave of conv_halo_p reads 10 fragments (6 weight + 4 pixel,
// 4 LDS cycles each) per 12 MFMAs (16 cycles each), and two such workgroups share one CU. The weight slab of these layers is
// small and never changes during the launch: 64 couts x 576 k (64->64) or 32 couts x 1152 k (a quarter of 128->128) is 73 KB =
// 288 VGPRs of a 512-register wave. So here every wave keeps ITS output channels' weights in registers for the whole launch
// (loaded once from L2 in MFMA A-fragment layout), one wave per SIMD, and the only LDS traffic of the main loop is the pixel
// operand: FM+2 halo-row fragments per (32-channel chunk, kx) feed 3*FM*FN MFMAs (16x16 tile: 6 reads per 48 MFMAs).
//
// Tiling: a workgroup owns a TH x TW pixel tile and all BN = WGN*FN*16 output channels; WGM waves split the tile's fragment rows,
// WGN waves split the channels (waves of one wm read the same pixel fragments). The input tile + halo is staged per 32-channel
// chunk by LDS-DMA through the 3-slot ring of conv_halo_p (same exact vmcnt accounting, unconditional buffer stores).
// TW = 16: a B fragment is 16 consecutive pixels of one halo row. TW = 8 (for 40-wide maps): a fragment is 2 rows x 8 pixels, the
// second row's columns rotated by 6 so that the 16 lanes of every ds_read_b128 lane group fall on 16 distinct 16-byte LDS slots
// (positions of lanes {0-3,12-15} and of lanes {4-11} must each be distinct mod 8 under the chunk swizzle; halo pitch 10).
#include "common.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <utility>
#include <vector>

namespace yp {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void lds_void;

template <int N> __device__ __forceinline__ void wr_wait_vmc() {
    static_assert(N >= 0 && N < 64, "vmcnt range");
    __builtin_amdgcn_s_waitcnt((N & 0xF) | (7 << 4) | (0xF << 8) | (((N >> 4) & 3) << 14));
}
__device__ __forceinline__ int wr_pswz(int row) { return ((row >> 2) & 1) << 1; }
template <int... Is, typename F> __device__ __forceinline__ void wr_static_for_impl(std::integer_sequence<int, Is...>, F&& f) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F> __device__ __forceinline__ void wr_static_for(F&& f) { wr_static_for_impl(std::make_integer_sequence<int, N>{}, f); }

template <int NCH, int FM, int FN, int WGM, int WGN, int TW, bool HAS_RES, bool OUT_F32>
__global__ __launch_bounds__(WGM * WGN * 64) void conv_wreg_kernel(const ConvParams p, const int tiles_h, const int tiles_w, const int G) {
    constexpr int NW = WGM * WGN;
    constexpr int RPF = 16 / TW;                      // tile rows per fragment
    constexpr int TH = WGM * FM * RPF;
    constexpr int PW = TW + 2;                        // halo pitch (positions)
    constexpr int HP = (TH + 2) * PW;
    constexpr int H_INSTR = (HP * 4 + 63) / 64;
    constexpr int LH = (H_INSTR + NW - 1) / NW;
    constexpr int HB = H_INSTR * 1024;
    constexpr int S = FM * FN;                        // stores per wave per tile
    constexpr int NSH = 3;
    constexpr int NH = RPF * FM + 2 - (RPF - 1);      // distinct halo fragment rows a wave reads per (chunk, kx): FM+2 (TW 16) / 2FM+1 (TW 8)
    constexpr unsigned OOB = 0x80000000u;
    static_assert(TW == 16 || TW == 8, "fragment shapes");
    static_assert(NCH * 9 * FN * 4 <= 320, "the weight slab of a wave must fit its registers");
    static_assert((NSH - 2) * LH + 2 * S < 64, "vmcnt immediate");

    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char* const Hs = smem;                   // NSH halo slots
    unsigned char* const dump = smem + NSH * HB;      // 1 KiB landing zone of padding loads
    unsigned char* const Wst = dump + 1024;           // weight staging image of one 32-channel chunk: [9][BN] rows of 64 B (prologue only)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned long long clk[5] = {0, 0, 0, 0, 0}, last = p.clk ? __builtin_amdgcn_s_memtime() : 0ull;
#define WR_STAMP(i) if (p.clk) { const unsigned long long now = __builtin_amdgcn_s_memtime(); clk[i] += now - last; last = now; }
    const int wm = wave % WGM, wn = wave / WGM;
    const int fr = lane & 15, fc = lane >> 4;
    // this lane's pixel inside a fragment: (row, col) and its position offset in the halo image
    const int prow = (TW == 16) ? 0 : (fr >> 3);
    const int pcol = (TW == 16) ? fr : (((fr & 7) + 6 * (fr >> 3)) & 7);
    const int ppos = prow * PW + pcol;

    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, j = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int j0 = bid;
    const int B = p.M / (p.Ho * p.Wo);
    const int num_tiles = B * tiles_h * tiles_w;

    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, (int)p.y_bytes, 0x00020000);

    float bias[FN][4];
#pragma unroll
    for (int a = 0; a < FN; ++a) {
        const int co = wn * (FN * 16) + a * 16 + fc * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) bias[a][r] = (co + r < p.Cout) ? p.bias[co + r] : 0.f;
    }

    bf16x8 wf[NCH][9][FN];          // this wave's weights for the whole launch (MFMA A fragments), filled below

    // ---- issue side: halo pieces of (tile it_tile, chunk it_c) --------------------------------------------------------
    // per-lane halo piece descriptors, fixed for the launch: position (hy,hx) inside the halo image and the byte offset relative
    // to the image's first pixel; a tile then costs two adds and two range checks per piece (no divisions in the loop)
    int hyx[LH];                      // hy << 16 | hx, or -1 when this lane's slot lies outside the image
    unsigned hrel[LH];
#pragma unroll
    for (int j = 0; j < LH; ++j) {
        const int ii = wave * LH + j;
        const int s = ii * 64 + lane;
        const int hp = s >> 2, pc = s & 3;
        const int c8 = pc ^ wr_pswz(hp);
        const int hy = hp / PW, hx = hp - hy * PW;
        hyx[j] = (ii < H_INSTR && hp < HP) ? ((hy << 16) | hx) : -1;
        hrel[j] = (unsigned)(((hy * p.W + hx) * p.x_stride) * 2 + c8 * 16);
    }
    unsigned hconst[LH];
    auto set_tile = [&](int tile) {
        int t = tile;                                   // (uniform: scalar arithmetic)
        const int tw = t % tiles_w; t /= tiles_w;
        const int th = t % tiles_h;
        const int b = t / tiles_h;
        const int h0 = th * TH - 1, w0 = tw * TW - 1;
        const unsigned base = (unsigned)((((b * p.H + h0) * p.W + w0) * p.x_stride + p.x_coff) * 2);
        const bool live = tile < num_tiles;
#pragma unroll
        for (int j = 0; j < LH; ++j) {
            const int hy = hyx[j] >> 16, hx = hyx[j] & 0xffff;
            const bool ok = live && (hyx[j] >= 0) && ((unsigned)(h0 + hy) < (unsigned)p.H) && ((unsigned)(w0 + hx) < (unsigned)p.W);
            hconst[j] = ok ? base + hrel[j] : OOB;
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
        if (++it_c == NCH) {
            it_c = 0;
            it_tile += G;
            set_tile(it_tile);
        }
    };

#pragma unroll
    for (int s = 0; s < NSH - 1; ++s) issue_next();

    // ---- weights: one 32-channel chunk per round through an LDS staging image ([tap][n] rows of 64 B, chunk swizzle), then into the
    //      registers of the waves that own the rows. Every workgroup of the launch wants the same bytes at the same moment: walked in
    //      the same order, all CUs of an XCD queue on one L2 channel at a time (stamps: 21k cycles for 295 KB per CU = 14 B/clk);
    //      each workgroup therefore starts its walk at a different piece. -------------------------------------------------------
    {
        constexpr int BN = WGN * FN * 16;
        constexpr int W_INSTR = 9 * BN / 16;                  // 1-KiB pieces per chunk
        const int rot = (int)(((unsigned)bid * 37u) % (unsigned)W_INSTR);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            for (int i0 = wave; i0 < W_INSTR; i0 += NW) {
                int ii = i0 + rot;
                ii = ii >= W_INSTR ? ii - W_INSTR : ii;
                const int s = ii * 64 + lane;
                const int rg = s >> 2, pc = s & 3;
                const int c8 = pc ^ wr_pswz(rg);
                const int t = rg / BN, n = rg - t * BN;
                const unsigned voff = (unsigned)((n * p.Kpad + t * p.Cin + c * 32 + c8 * 8) * 2);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_void*)(Wst + ii * 1024), 16, voff, 0, 0, 0);
            }
            wr_wait_vmc<0>();
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int a = 0; a < FN; ++a) {
                    const int rw = t * BN + wn * (FN * 16) + a * 16 + fr;
                    wf[c][t][a] = *(const bf16x8*)(Wst + rw * 64 + ((fc ^ wr_pswz(rw)) * 16));
                }
            if (c + 1 < NCH) {
                __builtin_amdgcn_s_waitcnt(0xc07f);           // lgkmcnt(0): the fragments are in registers before the image is overwritten
                __builtin_amdgcn_s_barrier();
            }
        }
    }
    WR_STAMP(0)

    int rd_slot = 0;
    unsigned epmask = 0;                // bit k: iteration (current-1-k) ended a tile
    bool first_iter = true;
    for (int tile = j0; tile < num_tiles; tile += G) {
        f32x4 acc[FN][FM];
#pragma unroll
        for (int a = 0; a < FN; ++a)
#pragma unroll
            for (int r = 0; r < FM; ++r) acc[a][r] = f32x4{bias[a][0], bias[a][1], bias[a][2], bias[a][3]};   // bias rides in the accumulator

#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (!first_iter) {
                const int k = __builtin_popcount(epmask & ((1u << (NSH - 1)) - 1u));
                if (k == 0) wr_wait_vmc<(NSH - 2) * LH>();
                else if (k == 1) wr_wait_vmc<(NSH - 2) * LH + S>();
                else wr_wait_vmc<(NSH - 2) * LH + 2 * S>();
                __builtin_amdgcn_s_barrier();
            }
            first_iter = false;
            WR_STAMP(1)
            issue_next();
            epmask <<= 1;
            WR_STAMP(2)

            const unsigned char* hsl = Hs + rd_slot * HB;
            // software pipeline over the 3*NH pixel fragments of this chunk: the read of fragment i+1 is issued before the MFMAs of
            // fragment i (one wave per SIMD: nothing else hides the ~100-cycle LDS latency), pinned by sched_group_barriers
            auto read_frag = [&](int i) -> bf16x8 {
                const int kx = i / NH, h = i - kx * NH;
                const int hp = (wm * FM * RPF + h) * PW + kx + ppos;
                return *(const bf16x8*)(hsl + hp * 64 + ((fc ^ wr_pswz(hp)) * 16));
            };
            bf16x8 xf = read_frag(0);
            wr_static_for<3 * NH>([&](auto I) {
                constexpr int i = decltype(I)::value;
                constexpr int kx = i / NH, h = i - kx * NH;
                bf16x8 xn = xf;
                if constexpr (i + 1 < 3 * NH) xn = read_frag(i + 1);
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int d = h - ky;                          // = RPF * r
                    if (d >= 0 && (d % RPF) == 0 && d / RPF < FM) {
                        const int r = d / RPF;
#pragma unroll
                        for (int a = 0; a < FN; ++a)
                            acc[a][r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[c][ky * 3 + kx][a], xf, acc[a][r], 0, 0, 0);
                    }
                }
                xf = xn;
            });
            rd_slot = (rd_slot + 1 == NSH) ? 0 : rd_slot + 1;
            WR_STAMP(3)
        }

        // ---- epilogue of `tile`: exactly S buffer stores per wave ---------------------------------------------------------
        {
            int t = tile;                               // (uniform)
            const int tw = t % tiles_w; t /= tiles_w;
            const int th = t % tiles_h;
            const int b = t / tiles_h;
            const int row0 = th * TH + (wm * FM) * RPF + prow, wo = tw * TW + pcol;
            const bool col_ok = wo < p.Wo;
            const unsigned m0 = (unsigned)((b * p.Ho + row0) * p.Wo + wo);          // pixel index of fragment 0; fragment r is RPF rows further
            const int co0 = wn * (FN * 16) + fc * 4;
            uint2 rres[FM][FN];
            if (HAS_RES) {
#pragma unroll
                for (int r = 0; r < FM; ++r) {
                    const bool pix_ok = col_ok && (row0 + r * RPF < p.Ho);
                    const unsigned m = m0 + (unsigned)(r * RPF * p.Wo);
#pragma unroll
                    for (int a = 0; a < FN; ++a) {
                        const int co = co0 + a * 16;
                        rres[r][a] = (pix_ok && co < p.Cout)
                                         ? *(const uint2*)((const __bf16*)p.res + (size_t)m * p.res_stride + p.res_coff + co)
                                         : make_uint2(0u, 0u);
                    }
                }
            }
            const unsigned es = OUT_F32 ? 4u : 2u;
            const unsigned off0 = (m0 * (unsigned)p.y_stride + (unsigned)(p.y_coff + co0)) * es;
            const unsigned roff = (unsigned)(RPF * p.Wo * p.y_stride) * es;
#pragma unroll
            for (int r = 0; r < FM; ++r) {
                const bool pix_ok = col_ok && (row0 + r * RPF < p.Ho);
#pragma unroll
                for (int a = 0; a < FN; ++a) {
                    const bool ok = pix_ok && (co0 + a * 16 < p.Cout);
                    float v[4] = {acc[a][r][0], acc[a][r][1], acc[a][r][2], acc[a][r][3]};
                    if (p.act == ACT_SILU) silu4_packed(v);
                    if (HAS_RES) {
                        const uint2 rr = rres[r][a];
                        v[0] += __uint_as_float(rr.x << 16); v[1] += __uint_as_float(rr.x & 0xffff0000u);
                        v[2] += __uint_as_float(rr.y << 16); v[3] += __uint_as_float(rr.y & 0xffff0000u);
                    }
                    const unsigned off = ok ? off0 + (unsigned)r * roff + (unsigned)(a * 16) * es : OOB;
                    if (OUT_F32) {
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, make_float4(v[0], v[1], v[2], v[3])), yrs, off, 0, 0);
                    } else {
                        __attribute__((aligned(8))) __bf16 o[4] = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                        __builtin_amdgcn_raw_buffer_store_b64(*(const __attribute__((ext_vector_type(2))) unsigned*)o, yrs, off, 0, 0);
                    }
                }
            }
        }
        epmask |= 1u;
        WR_STAMP(4)
    }
    wr_wait_vmc<0>();
    if (p.clk && lane == 0)
        for (int i = 0; i < 5; ++i) p.clk[((size_t)blockIdx.x * NW + wave) * 5 + i] = clk[i];
#undef WR_STAMP
}

// ---------------------------------------------------------------------------------------------------------------
struct WregCfg { int NCH, FM, FN, WGM, WGN, TW; const char* name; };
static const WregCfg kWreg[] = {
    {2, 4, 4, 4, 1, 16, "conv_wreg_kernel<2,4,4,4,1,16>"},   // 0: 64 -> <=64, 16x16 px tile
    {2, 2, 4, 4, 1, 16, "conv_wreg_kernel<2,2,4,4,1,16>"},   // 1: 64 -> <=64,  8x16 px tile
    {4, 8, 2, 1, 4, 16, "conv_wreg_kernel<4,8,2,1,4,16>"},   // 2: 128 -> <=128, 8x16 px tile
    {4, 4, 2, 1, 4, 16, "conv_wreg_kernel<4,4,2,1,4,16>"},   // 3: 128 -> <=128, 4x16 px tile
    {4, 4, 2, 1, 4, 8, "conv_wreg_kernel<4,4,2,1,4,8>"},     // 4: 128 -> <=128, 8x8 px tile (40-wide maps)
    {2, 2, 4, 4, 1, 8, "conv_wreg_kernel<2,2,4,4,1,8>"},     // 5: 64 -> <=64, 16x8 px tile
    {4, 4, 2, 2, 2, 16, "conv_wreg_kernel<4,4,2,2,2,16>"},   // 6: 128 -> <=64, 8x16 px tile
    // two waves per SIMD (256 registers each, 144 of them weights): the partner's MFMAs cover a wave's LDS latency and epilogue
    {2, 4, 2, 4, 2, 16, "conv_wreg_kernel<2,4,2,4,2,16>"},   // 7: 64 -> <=64, 16x16 px tile, 8 waves
    {2, 2, 2, 4, 2, 16, "conv_wreg_kernel<2,2,2,4,2,16>"},   // 8: 64 -> <=64,  8x16 px tile, 8 waves
    {4, 8, 1, 1, 8, 16, "conv_wreg_kernel<4,8,1,1,8,16>"},   // 9: 128 -> <=128, 8x16 px tile, 8 waves
    {4, 4, 1, 1, 8, 8, "conv_wreg_kernel<4,4,1,1,8,8>"},     // 10: 128 -> <=128, 8x8 px tile, 8 waves
    {4, 8, 1, 1, 8, 8, "conv_wreg_kernel<4,8,1,1,8,8>"},     // 11: 128 -> <=128, 16x8 px tile, 8 waves
    {2, 2, 2, 4, 2, 8, "conv_wreg_kernel<2,2,2,4,2,8>"},     // 12: 64 -> <=64, 16x8 px tile, 8 waves
};
constexpr int kNumWreg = (int)(sizeof(kWreg) / sizeof(kWreg[0]));

int conv_wreg_num_cfgs() { return kNumWreg; }

static size_t wreg_lds(const WregCfg& k) {
    const int RPF = 16 / k.TW, TH = k.WGM * k.FM * RPF;
    const int HP = (TH + 2) * (k.TW + 2), H_INSTR = (HP * 4 + 63) / 64;
    return (size_t)3 * H_INSTR * 1024 + 1024 + (size_t)9 * (k.WGN * k.FN * 16) * 64;
}

bool conv_wreg_cfg_valid(const ConvParams& p, int c) {
    if (c < 0 || c >= kNumWreg) return false;
    const WregCfg& k = kWreg[c];
    if (p.ks != 3 || p.stride != 1 || p.pad != 1 || p.up != 1 || p.Cin != k.NCH * 32 || p.Kpad != 9 * p.Cin || p.x2_C > 0 || p.w2) return false;
    if (p.x_bytes >= (1ull << 31) || p.w_bytes >= (1ull << 31) || p.y_bytes >= (1ull << 31)) return false;
    if ((p.Cout & 3) || (p.y_stride & 3) || (p.y_coff & 3) || (p.res && ((p.res_stride & 3) || (p.res_coff & 3)))) return false;
    if (p.res && p.out_f32) return false;
    const int BN = k.WGN * k.FN * 16;
    if (p.Cout > BN || p.Cout * 2 <= BN) return false;                       // one N tile, at least half used
    if ((size_t)((p.Cout + 127) / 128 * 128) * p.Kpad * 2 > p.w_bytes) return false;
    const int RPF = 16 / k.TW, TH = k.WGM * k.FM * RPF;
    const long covered = (long)((p.Ho + TH - 1) / TH * TH) * ((p.Wo + k.TW - 1) / k.TW * k.TW);
    if (covered * 4 > (long)p.Ho * p.Wo * 5) return false;                  // at most 25 % of the tile area outside the map
    return true;
}

const char* conv_wreg_kernel_name(int c) { return kWreg[c].name; }

template <int NCH, int FM, int FN, int WGM, int WGN, int TW, bool HAS_RES, bool OUT_F32>
static hipError_t launch_wreg_var(const ConvParams& p, const WregCfg& k, hipStream_t st) {
    constexpr int RPF = 16 / TW, TH = WGM * FM * RPF;
    const size_t sh = wreg_lds(k);
    const int B = p.M / (p.Ho * p.Wo);
    const int tiles_h = (p.Ho + TH - 1) / TH, tiles_w = (p.Wo + TW - 1) / TW;
    const int num_tiles = B * tiles_h * tiles_w;
    // one workgroup per CU (512-register waves); use only as many workgroups as keep every one of them on the same number of tiles
    const int rounds = (num_tiles + 255) / 256;
    int G = (num_tiles + rounds - 1) / rounds;
    if (G < 1) G = 1;
    auto kern = conv_wreg_kernel<NCH, FM, FN, WGM, WGN, TW, HAS_RES, OUT_F32>;
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
        if (e != hipSuccess) return e;
        attr = true;
    }
    static const bool clocks = [] { const char* v = std::getenv("YOLOP_WREG_CLOCKS"); return v && *v == '1'; }();   // debug: per-phase s_memtime sums
    if (clocks) {
        ConvParams q = p;
        const size_t n = (size_t)G * WGM * WGN * 5;
        if (hipMalloc((void**)&q.clk, n * 8) != hipSuccess) return hipErrorOutOfMemory;
        (void)hipMemset(q.clk, 0, n * 8);
        hipLaunchKernelGGL(kern, dim3(G), dim3(WGM * WGN * 64), sh, st, q, tiles_h, tiles_w, G);
        (void)hipStreamSynchronize(st);
        std::vector<unsigned long long> h(n);
        (void)hipMemcpy(h.data(), q.clk, n * 8, hipMemcpyDeviceToHost);
        (void)hipFree(q.clk);
        double s5[5] = {0, 0, 0, 0, 0}, mx[5] = {0, 0, 0, 0, 0};
        for (size_t w = 0; w < n / 5; ++w)
            for (int i = 0; i < 5; ++i) { s5[i] += (double)h[w * 5 + i]; mx[i] = std::max(mx[i], (double)h[w * 5 + i]); }
        fprintf(stderr, "[wreg clocks] %s G=%d tiles=%d  per-wave mean cycles: prologue %.0f  wait+barrier %.0f  issue %.0f  compute %.0f  epilogue %.0f   (max %.0f %.0f %.0f %.0f %.0f)\n",
                k.name, G, num_tiles, s5[0] / (n / 5), s5[1] / (n / 5), s5[2] / (n / 5), s5[3] / (n / 5), s5[4] / (n / 5), mx[0], mx[1], mx[2], mx[3], mx[4]);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(kern, dim3(G), dim3(WGM * WGN * 64), sh, st, p, tiles_h, tiles_w, G);
    return hipGetLastError();
}

template <int NCH, int FM, int FN, int WGM, int WGN, int TW>
static hipError_t launch_wreg_one(const ConvParams& p, const WregCfg& k, hipStream_t st) {
    if (p.out_f32) return launch_wreg_var<NCH, FM, FN, WGM, WGN, TW, false, true>(p, k, st);
    if (p.res) return launch_wreg_var<NCH, FM, FN, WGM, WGN, TW, true, false>(p, k, st);
    return launch_wreg_var<NCH, FM, FN, WGM, WGN, TW, false, false>(p, k, st);
}

hipError_t launch_conv_wreg(const ConvParams& p, int c, hipStream_t st) {
    const WregCfg& k = kWreg[c];
    switch (c) {
        case 0: return launch_wreg_one<2, 4, 4, 4, 1, 16>(p, k, st);
        case 1: return launch_wreg_one<2, 2, 4, 4, 1, 16>(p, k, st);
        case 2: return launch_wreg_one<4, 8, 2, 1, 4, 16>(p, k, st);
        case 3: return launch_wreg_one<4, 4, 2, 1, 4, 16>(p, k, st);
        case 4: return launch_wreg_one<4, 4, 2, 1, 4, 8>(p, k, st);
        case 5: return launch_wreg_one<2, 2, 4, 4, 1, 8>(p, k, st);
        case 6: return launch_wreg_one<4, 4, 2, 2, 2, 16>(p, k, st);
        case 7: return launch_wreg_one<2, 4, 2, 4, 2, 16>(p, k, st);
        case 8: return launch_wreg_one<2, 2, 2, 4, 2, 16>(p, k, st);
        case 9: return launch_wreg_one<4, 8, 1, 1, 8, 16>(p, k, st);
        case 10: return launch_wreg_one<4, 4, 1, 1, 8, 8>(p, k, st);
        case 11: return launch_wreg_one<4, 8, 1, 1, 8, 8>(p, k, st);
        default: return launch_wreg_one<2, 2, 2, 4, 2, 8>(p, k, st);
    }
}

}  // namespace yp
