// Stride-2 member of the halo family (see conv_halo_p.hip): 3x3, stride 2, pad 1, bf16, persistent, weights resident in LDS.
//
// The im2col form of a stride-2 3x3 re-reads every input pixel 2.25 times through the 1-KiB LDS-DMA pieces and streams
// it a k-step at a time (model.1 / model.3 / model.17: 100 / 80 / 35 us at 3.1 / 2.0 / 1.9 TB/s). Here the input patch of
// an output tile - (2*TH+1) rows x 33 columns per 32-channel chunk - is staged ONCE. A lane of the MFMA B operand reads its
// own pixel, so the stride-2 gather is free in principle; what has to be arranged is the bank pattern: 16 lanes reading
// every second pixel of a row hit 2 of the 16-byte slots. The LDS image therefore stores the EVEN and the ODD input columns
// of a row as two planes (17 + 16 pixels): tap kx = 0 reads plane E at n, kx = 1 plane O at n, kx = 2 plane E at n+1 -
// always 16 consecutive LDS rows, conflict-free under the usual chunk swizzle. The LDS-DMA fill computes the source pixel of
// every LDS row from that layout (the destination of a piece is lane-linear, the source address is per lane).
// Rows: input row jr = 2r + ky serves output row r; an even input row feeds (r = jr/2, ky = 0) and (r = jr/2 - 1, ky = 2),
// an odd one (r = (jr-1)/2, ky = 1), so each fragment read still feeds up to 2*FN MFMAs.
// vmcnt accounting, persistence, epilogue: identical to conv_halo_p.hip.
#include "common.h"

namespace yp {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ float silu_s2(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
template <int N> __device__ __forceinline__ void wait_vs2() {
    static_assert(N >= 0 && N < 64, "vmcnt range");
    __builtin_amdgcn_s_waitcnt((N & 0xF) | (7 << 4) | (0xF << 8) | (((N >> 4) & 3) << 14));
}
__device__ __forceinline__ int sswz(int row) { return ((row >> 2) & 1) << 1; }

template <int FM, int FN, int WGM, int WGN, int NSH, bool HAS_RES, bool OUT_F32, bool PW2>
__global__ __launch_bounds__(WGM * WGN * 64) void conv_halo_s2_kernel(const ConvParams p, const int tiles_h, const int tiles_w,
                                                                    const int ntiles, const int G) {
    constexpr int NW = WGM * WGN;
    constexpr int TH = WGM * FM, BN = WGN * FN * 16;
    constexpr int HR = 2 * TH + 1;                     // input rows of the patch
    constexpr int HP = HR * 33;                        // LDS rows (pixels) per slot: [jr][E0..E16 | O0..O15]
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
            const int c8 = pc ^ sswz(rg);
            const int n = rg % BN, q = rg / BN;
            const int tap = q % 9, ch = q / 9;
            const unsigned voff = (unsigned)(((n0 + n) * p.Kpad + tap * p.Cin + ch * 32 + c8 * 8) * 2);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_void*)(Wres + ii * 1024), 16, voff, 0, 0, 0);
        }
    }

    // ---- PW2: the trailing 1x1's weights [C2 = BN][BN] (128-B rows, 8-slot swizzle) stay in LDS behind the 3x3 weights ------
    unsigned char* const W2s = Wres + (size_t)9 * nchunk * BN * 64;
    float bias2[FN][4];
    if (PW2) {
        static_assert(!PW2 || BN == 64, "the fused trailing 1x1 is written for a 64-wide intermediate and output");
        const __amdgpu_buffer_rsrc_t w2rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w2, 0, (int)p.w2_bytes, 0x00020000);
        for (int ii = wave; ii < BN * 8 / 64; ii += NW) {          // BN rows x 8 chunks of 16 B
            const int s = ii * 64 + lane;
            const int row = s >> 3, pc = s & 7;
            const int c8 = pc ^ ((row >> 1) & 7);
            const unsigned voff = (unsigned)((row * p.Kpad2 + c8 * 8) * 2);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w2rs, (lds_void*)(W2s + ii * 1024), 16, voff, 0, 0, 0);
        }
#pragma unroll
        for (int a = 0; a < FN; ++a) {
            const int co = wn * (FN * 16) + a * 16 + fc * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) bias2[a][r] = (co + r < p.C2) ? p.bias2[co + r] : 0.f;
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
            const int c8 = pc ^ sswz(hp);
            const int jr = hp / 33, e = hp - jr * 33;
            const int jc = (e < 17) ? 2 * e : 2 * (e - 17) + 1;            // column of the patch this LDS row holds
            const int hi = 2 * h0 - 1 + jr, wi = 2 * w0 - 1 + jc;
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
    wait_vs2<0>();                      // weights + first chunks landed (once per workgroup)
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
                if (k == 0) wait_vs2<(NSH - 2) * LH>();
                else if (k == 1) wait_vs2<(NSH - 2) * LH + S>();
                else wait_vs2<(NSH - 2) * LH + 2 * S>();
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
                const int eoff = (kx == 1) ? 17 + fr : fr + (kx >> 1);      // plane O at n, plane E at n / n+1
#pragma unroll
                for (int jj = 0; jj < 2 * FM + 1; ++jj) {
                    const int hp = (2 * wm * FM + jj) * 33 + eoff;
                    const bf16x8 xf = *(const bf16x8*)(hsl + swz64((unsigned)(hp * 64 + fc * 16)));
                    if (jj & 1) {
#pragma unroll
                        for (int a = 0; a < FN; ++a)
                            acc[a][jj >> 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1][a], xf, acc[a][jj >> 1], 0, 0, 0);
                    } else {
                        if ((jj >> 1) < FM) {
#pragma unroll
                            for (int a = 0; a < FN; ++a)
                                acc[a][jj >> 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][a], xf, acc[a][jj >> 1], 0, 0, 0);
                        }
                        if ((jj >> 1) >= 1) {
#pragma unroll
                            for (int a = 0; a < FN; ++a)
                                acc[a][(jj >> 1) - 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[2][a], xf, acc[a][(jj >> 1) - 1], 0, 0, 0);
                        }
                    }
                }
            }
            rd_slot = (rd_slot + 1 == NSH) ? 0 : rd_slot + 1;
        }

        if (PW2) {
            // the tile's intermediate (bias + SiLU, bf16 - exactly the tensor the unfused graph would store) goes through the
            // halo slot this tile just finished with (free until the ring refills it after the next iteration's barrier)
            // and is multiplied by the resident 64x64 weights; acc is then the trailing 1x1's accumulator
            unsigned char* tb = Hs + ((rd_slot == 0) ? NSH - 1 : rd_slot - 1) * HB;
            __builtin_amdgcn_s_barrier();                     // every wave is done reading that slot
#pragma unroll
            for (int r = 0; r < FM; ++r) {
                const int px = (wm * FM + r) * 16 + fr;
#pragma unroll
                for (int a = 0; a < FN; ++a) {
                    const int co = wn * (FN * 16) + a * 16 + fc * 4;
                    __attribute__((aligned(8))) __bf16 o[4];
                    float sv[4] = {acc[a][r][0], acc[a][r][1], acc[a][r][2], acc[a][r][3]};
                    if (p.act == ACT_SILU) silu4_packed(sv);
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] = (__bf16)sv[i];
                    // (asm: a compiler-visible ds_write would first drain vmcnt to 0, i.e. wait for the next tile's halo DMA)
                    asm volatile("ds_write_b64 %0, %1" ::"v"((unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)(tb + px * 128 + (((co >> 3) ^ ((px >> 1) & 7)) * 16) + (co & 7) * 2)),
                                 "v"(*(const unsigned long long*)o) : "memory");
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int a = 0; a < FN; ++a)
#pragma unroll
                for (int r = 0; r < FM; ++r) acc[a][r] = f32x4{bias2[a][0], bias2[a][1], bias2[a][2], bias2[a][3]};
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) {
                bf16x8 w2f[FN], t2f[FM];
#pragma unroll
                for (int a = 0; a < FN; ++a) {
                    const int rw = wn * (FN * 16) + a * 16 + fr;
                    w2f[a] = *(const bf16x8*)(W2s + rw * 128 + (((ss * 4 + fc) ^ ((rw >> 1) & 7)) * 16));
                }
#pragma unroll
                for (int r = 0; r < FM; ++r) {
                    const int px = (wm * FM + r) * 16 + fr;
                    t2f[r] = *(const bf16x8*)(tb + px * 128 + (((ss * 4 + fc) ^ ((px >> 1) & 7)) * 16));
                }
#pragma unroll
                for (int a = 0; a < FN; ++a)
#pragma unroll
                    for (int r = 0; r < FM; ++r) acc[a][r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2f[a], t2f[r], acc[a][r], 0, 0, 0);
            }
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
                    const bool ok = pix_ok && (co < (PW2 ? p.C2 : p.Cout));
                    float v[4] = {acc[a][r][0], acc[a][r][1], acc[a][r][2], acc[a][r][3]};
                    if ((PW2 ? p.act2 : p.act) == ACT_SILU) silu4_packed(v);
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
    wait_vs2<0>();
}

// ---------------------------------------------------------------------------------------------------------------
struct HaloS2Cfg { int FM, FN, WGM, WGN; const char* name; };
static const HaloS2Cfg kS2[] = {
    {2, 2, 4, 2, "conv_halo_s2_kernel<2,2,4,2,3>"},   // 0: 8x16 out px x 64 ch, 8 waves
    {1, 2, 4, 2, "conv_halo_s2_kernel<1,2,4,2,3>"},   // 1: 4x16 out px x 64 ch, 8 waves
    {2, 2, 4, 1, "conv_halo_s2_kernel<2,2,4,1,3>"},   // 2: 8x16 out px x 32 ch, 4 waves
    {1, 2, 4, 1, "conv_halo_s2_kernel<1,2,4,1,3>"},   // 3: 4x16 out px x 32 ch, 4 waves
    {2, 4, 4, 2, "conv_halo_s2_kernel<2,4,4,2,3>"},   // 4: 8x16 out px x 128 ch, 8 waves (Cin = 32)
};
constexpr int kNumS2 = (int)(sizeof(kS2) / sizeof(kS2[0]));
int conv_halo_s2_num_cfgs() { return kNumS2; }
const char* conv_halo_s2_kernel_name(int c) { return kS2[c].name; }

static size_t halo_s2_lds(const HaloS2Cfg& k, int Cin, bool pw2 = false) {
    const int TH = k.WGM * k.FM, BN = k.WGN * k.FN * 16;
    const int HP = (2 * TH + 1) * 33, H_INSTR = (HP * 4 + 63) / 64;
    return (size_t)3 * H_INSTR * 1024 + 1024 + (size_t)9 * (Cin / 32) * BN * 64 + (pw2 ? (size_t)BN * 128 : 0);
}

bool conv_halo_s2_cfg_valid(const ConvParams& p, int c) {
    if (c < 0 || c >= kNumS2) return false;
    if (p.ks != 3 || p.stride != 2 || p.pad != 1 || p.up != 1 || (p.Cin % 32) != 0 || (p.Kpad != 9 * p.Cin)) return false;
    if ((p.H & 1) || (p.W & 1) || p.Ho * 2 != p.H || p.Wo * 2 != p.W) return false;
    if (p.x_bytes >= (1ull << 31) || p.w_bytes >= (1ull << 31) || p.y_bytes >= (1ull << 31)) return false;
    if ((p.Cout & 3) || (p.y_stride & 3) || (p.y_coff & 3) || (p.res && ((p.res_stride & 3) || (p.res_coff & 3)))) return false;
    if (p.res && p.out_f32) return false;
    const HaloS2Cfg& k = kS2[c];
    const int BN = k.WGN * k.FN * 16, TH = k.WGM * k.FM;
    const int cpad = (p.Cout + 31) / 32 * 32;
    if (BN > cpad) return false;
    if (halo_s2_lds(k, p.Cin) > 160 * 1024) return false;
    const long covered = (long)((p.Ho + TH - 1) / TH * TH) * ((p.Wo + 15) / 16 * 16);
    if (covered * 2 > (long)p.Ho * p.Wo * 3) return false;
    return true;
}

template <int FM, int FN, int WGM, int WGN, bool HAS_RES, bool OUT_F32, bool PW2 = false>
static hipError_t launch_halo_s2_var(const ConvParams& p, const HaloS2Cfg& k, hipStream_t st) {
    constexpr int TH = WGM * FM, BN = WGN * FN * 16;
    const size_t sh = halo_s2_lds(k, p.Cin, PW2);
    const int B = p.M / (p.Ho * p.Wo);
    const int tiles_h = (p.Ho + TH - 1) / TH, tiles_w = (p.Wo + 15) / 16, ntiles = (p.Cout + BN - 1) / BN;
    const int num_tiles = B * tiles_h * tiles_w;
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(2, (160 * 1024) / sh));
    int G = (256 * per_cu) / ntiles;
    if (G < 1) G = 1;
    if (G > num_tiles) G = num_tiles;
    auto kern = conv_halo_s2_kernel<FM, FN, WGM, WGN, 3, HAS_RES, OUT_F32, PW2>;
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
        if (e != hipSuccess) return e;
        attr = true;
    }
    hipLaunchKernelGGL(kern, dim3(G * ntiles), dim3(WGM * WGN * 64), sh, st, p, tiles_h, tiles_w, ntiles, G);
    return hipGetLastError();
}

template <int FM, int FN, int WGM, int WGN>
static hipError_t launch_halo_s2_one(const ConvParams& p, const HaloS2Cfg& k, hipStream_t st) {
    if (p.out_f32) return launch_halo_s2_var<FM, FN, WGM, WGN, false, true>(p, k, st);
    if (p.res) return launch_halo_s2_var<FM, FN, WGM, WGN, true, false>(p, k, st);
    return launch_halo_s2_var<FM, FN, WGM, WGN, false, false>(p, k, st);
}

// ---- fused trailing 1x1 ------------------------------------------------------------------------------------------------
// valid when the 3x3 s2 conv and the 1x1 behind it are both 64 wide (model.1 -> model.2.cv1 of v10-S), nothing carries a residual,
// the intermediate has no other reader (the graph pass checks that), and LDS holds both weight sets beside the halo ring
int conv_halo_s2_pw_cfg(const ConvParams& p) {
    if (p.w2 == nullptr && p.C2 == 0) return -1;
    if (p.Cout != 64 || p.C2 != 64 || p.Kpad2 != 64 || p.res || p.out_f32) return -1;
    for (int c : {0, 1}) {
        ConvParams q = p;
        q.Cout = p.C2;                                          // the store-side checks of the plain form apply to the final view
        if (!conv_halo_s2_cfg_valid(q, c)) continue;
        if (halo_s2_lds(kS2[c], p.Cin, true) > 160 * 1024) continue;
        return c;
    }
    return -1;
}
const char* conv_halo_s2_pw_kernel_name(int c) { return c == 0 ? "conv_halo_s2_kernel<2,2,4,2,3,false,false,true>" : "conv_halo_s2_kernel<1,2,4,2,3,false,false,true>"; }

hipError_t launch_conv_halo_s2(const ConvParams& p, int c, hipStream_t st) {
    const HaloS2Cfg& k = kS2[c];
    if (p.C2 > 0) {
        if (c == 0) return launch_halo_s2_var<2, 2, 4, 2, false, false, true>(p, k, st);
        if (c == 1) return launch_halo_s2_var<1, 2, 4, 2, false, false, true>(p, k, st);
        return hipErrorInvalidValue;
    }
    switch (c) {
        case 0: return launch_halo_s2_one<2, 2, 4, 2>(p, k, st);
        case 1: return launch_halo_s2_one<1, 2, 4, 2>(p, k, st);
        case 2: return launch_halo_s2_one<2, 2, 4, 1>(p, k, st);
        case 3: return launch_halo_s2_one<1, 2, 4, 1>(p, k, st);
        default: return launch_halo_s2_one<2, 4, 4, 2>(p, k, st);
    }
}

}  // namespace yp
