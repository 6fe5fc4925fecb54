// Fused front end of the network: stem (3x3 s2, BGR u8 -> 32 ch) -> model.1 (3x3 s2, 32 -> 64) -> model.2.cv1 (1x1, 64 -> 64)
// in ONE persistent kernel (bf16). SURVEY.md A.3 layers 0-2 [U]; run inside `.predict` (reference yolo_seg/app.py:91).
//
// Unfused, the stem writes 210 MB and model.1 reads them back (0.42 GB of the step's HBM traffic, two launches of 68 + 93 us).
// Here a workgroup owns an 8x16 tile of model.1's output. Per tile:
//   A  every thread builds the im2col row (27 taps: u8 -> x/255 -> bf16, K padded to 32) of one or two of the 561 stem
//      pixels the tile's 17x33 input patch needs; the u8 dwords of the NEXT tile are already in flight in registers
//   B  stem GEMM on MFMA (36 pixel fragments x 2 channel halves), bias + SiLU + bf16, written straight into the halo image
//      of conv_halo_s2.hip (even / odd column planes) - zero for pixels outside the stem's output (model.1's padding)
//   C  model.1: the stride-2 halo loop of conv_halo_s2.hip over that image, weights resident
//   D  model.2.cv1: the tile's bias + SiLU + bf16 intermediate through LDS, times the resident 64x64 weights (PW2 form)
//   E  bias + SiLU, bf16 stores
// Rounding points are those of the three separate kernels (each intermediate is rounded to bf16 exactly where the unfused
// graph stores it), so the result differs from the unfused engine only by fp32 summation order inside a stage.
#include "common.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <type_traits>

namespace yp {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ float silu_fe(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ void silu4(float* v) { silu4_packed(v); }
__device__ __forceinline__ int fswz(int row) { return ((row >> 2) & 1) << 1; }

constexpr int FE_TH = 8;                         // output rows per tile (4 waves in M x 2 rows)
constexpr int FE_HR = 2 * FE_TH + 1;             // 17 stem rows
constexpr int FE_HP = FE_HR * 33;                // 561 stem pixels per tile
constexpr int FE_HPAD = 576;                     // 36 pixel fragments
constexpr int FE_HB = 36 * 1024;                 // halo image bytes
constexpr int FE_NW = 8;

// the three aligned dwords that cover the 9 bytes (3 px x BGR) of one tap row of stem pixel (sy, sx); zero outside the image
struct FeRows { unsigned d[3][3]; };

__device__ __forceinline__ void fe_load(const FrontParams& p, const __amdgpu_buffer_rsrc_t irs, int b, int sy, int sx, bool live, FeRows& r) {
    // buffer loads: an out-of-range offset (dead pixel, image row above / below the frame, or the dword in front of the very
    // first byte) reads as zero. The dword left of column 0 of any other row holds the previous row's tail: those bytes are
    // the ones fe_row masks for sx == 0. Right of the last pixel nothing beyond the 9 needed bytes is ever used.
    const int rowbytes = p.imgW * 3;
    const int ab = ((sx * 2 - 1) * 3) & ~3;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int hi = sy * 2 - 1 + ky;
        const bool rok = live && (unsigned)hi < (unsigned)p.imgH;
        const int base = (b * p.imgH + hi) * rowbytes + ab;        // -4 for the first pixel of the first row of the first frame
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int off = base + 4 * k;                          // (signed: no 32-bit wrap inside the address unit)
            r.d[ky][k] = __builtin_amdgcn_raw_buffer_load_b32(irs, (rok && off >= 0) ? (unsigned)off : 0x80000000u, 0, 0);
        }
    }
}

__device__ __forceinline__ void fe_row(const FeRows& r, int sx, unsigned char* dst, int hp) {
    __attribute__((aligned(16))) __bf16 row[32];
#pragma unroll
    for (int i = 27; i < 32; ++i) row[i] = (__bf16)0.f;
    const int sb = (sx * 2 - 1) * 3;
    const unsigned sbytes = (unsigned)(sb - (sb & ~3));
    const float k = 1.0f / 255.0f;                // bf16(x * fl(1/255)) == bf16(x / 255) for all 256 byte values
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        unsigned w0 = __builtin_amdgcn_alignbyte(r.d[ky][1], r.d[ky][0], sbytes);
        const unsigned w1 = __builtin_amdgcn_alignbyte(r.d[ky][2], r.d[ky][1], sbytes);
        const unsigned w2 = __builtin_amdgcn_alignbyte(0u, r.d[ky][2], sbytes);
        if (sx == 0) w0 &= 0xff000000u;
        __bf16* rr = row + ky * 9;
        rr[0] = (__bf16)((float)(w0 & 0xffu) * k);          rr[1] = (__bf16)((float)((w0 >> 8) & 0xffu) * k);
        rr[2] = (__bf16)((float)((w0 >> 16) & 0xffu) * k);  rr[3] = (__bf16)((float)(w0 >> 24) * k);
        rr[4] = (__bf16)((float)(w1 & 0xffu) * k);          rr[5] = (__bf16)((float)((w1 >> 8) & 0xffu) * k);
        rr[6] = (__bf16)((float)((w1 >> 16) & 0xffu) * k);  rr[7] = (__bf16)((float)(w1 >> 24) * k);
        rr[8] = (__bf16)((float)(w2 & 0xffu) * k);
    }
    const uint4* r4 = (const uint4*)row;
#pragma unroll
    for (int c = 0; c < 4; ++c) *(uint4*)(dst + hp * 64 + ((c ^ fswz(hp)) * 16)) = r4[c];
}

__global__ __launch_bounds__(FE_NW * 64) void frontend_kernel(const FrontParams p, const int tiles_h, const int tiles_w, const int G) {
    constexpr int FM = 2, FN = 2, WGM = 4;
    constexpr int BN = 64;
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char* const Hs = smem;                          // halo image of the stem output  [561 px][32 ch]
    unsigned char* const Ib = Hs + FE_HB;                    // im2col rows                    [576 px][32 k]
    unsigned char* const W1s = Ib + FE_HB;                   // model.1 weights                [9 taps][64 co][32 ci]
    unsigned char* const W2s = W1s + 9 * BN * 64;            // model.2.cv1 weights            [64 co][64 ci] (128-B rows)
    unsigned char* const W0s = W2s + BN * 128;               // stem weights                   [32 co][32 k]
    unsigned char* const Vs = W0s + 32 * 64;                 // per stem pixel of the tile: inside the stem's output grid?  [576]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WGM, wn = wave / WGM;
    const int fr = lane & 15, fc = lane >> 4;
    const int num_tiles = p.B * tiles_h * tiles_w;

    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, j = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }

    const __amdgpu_buffer_rsrc_t w1rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w1, 0, (int)p.w1_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w2rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w2, 0, (int)p.w2_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w0rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w0, 0, 32 * 32 * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, (int)p.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t irs = __builtin_amdgcn_make_buffer_rsrc((void*)p.img, 0, (int)((long)p.B * p.imgH * p.imgW * 3), 0x00020000);

    float bias1[FN][4], bias2[FN][4];
#pragma unroll
    for (int a = 0; a < FN; ++a) {
        const int co = wn * (FN * 16) + a * 16 + fc * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) { bias1[a][r] = p.bias1[co + r]; bias2[a][r] = p.bias2[co + r]; }
    }
    float4 bias0[2];
    bias0[0] = *(const float4*)(p.bias0 + fc * 4);
    bias0[1] = *(const float4*)(p.bias0 + 16 + fc * 4);

    // ---- resident weights --------------------------------------------------------------------------------------------------
    for (int ii = wave; ii < 9 * BN * 64 / 1024; ii += FE_NW) {          // model.1: row rg = tap*BN + n
        const int s = ii * 64 + lane;
        const int rg = s >> 2, pc = s & 3;
        const int c8 = pc ^ fswz(rg);
        const int n = rg % BN, tap = rg / BN;
        const unsigned voff = (unsigned)((n * p.Kpad1 + tap * 32 + c8 * 8) * 2);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w1rs, (lds_void*)(W1s + ii * 1024), 16, voff, 0, 0, 0);
    }
    for (int ii = wave; ii < BN * 128 / 1024; ii += FE_NW) {             // model.2.cv1: 128-B rows
        const int s = ii * 64 + lane;
        const int row = s >> 3, pc = s & 7;
        const int c8 = pc ^ ((row >> 1) & 7);
        const unsigned voff = (unsigned)((row * p.Kpad2 + c8 * 8) * 2);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w2rs, (lds_void*)(W2s + ii * 1024), 16, voff, 0, 0, 0);
    }
    for (int ii = wave; ii < 2; ii += FE_NW) {                           // stem: [32 co][32 k]
        const int s = ii * 64 + lane;
        const int row = s >> 2, pc = s & 3;
        const int c8 = pc ^ fswz(row);
        const unsigned voff = (unsigned)((row * 32 + c8 * 8) * 2);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w0rs, (lds_void*)(W0s + ii * 1024), 16, voff, 0, 0, 0);
    }

    // ---- per-thread stem pixels of a tile: hp = tid and (tid < 49) hp = tid + 512 ------------------------------------------
    auto patch_px = [&](int tile, int hp, int& b, int& sy, int& sx) -> bool {
        int t = tile;
        const int tw = t % tiles_w; t /= tiles_w;
        const int th = t % tiles_h;
        b = t / tiles_h;
        const int jr = hp / 33, e = hp - jr * 33;
        const int jc = (e < 17) ? 2 * e : 2 * (e - 17) + 1;
        sy = 2 * th * FE_TH - 1 + jr;
        sx = 2 * tw * 16 - 1 + jc;
        return tile < num_tiles && hp < FE_HP && (unsigned)sy < (unsigned)p.H1 && (unsigned)sx < (unsigned)p.W1;
    };
    const bool two = tid < FE_HP - 512;
    FeRows ra, rb;
    int ab_, asy, asx, bb_, bsy, bsx;
    bool alive, blive;
    int tile = bid;
    alive = patch_px(tile, tid, ab_, asy, asx);
    fe_load(p, irs, ab_, asy, asx, alive, ra);
    blive = two && patch_px(tile, tid + 512, bb_, bsy, bsx);
    if (two) fe_load(p, irs, bb_, bsy, bsx, blive, rb);

    __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (0xF << 8));               // weights landed
    __builtin_amdgcn_s_barrier();

    bf16x8 w0f[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) { const int r = h * 16 + fr; w0f[h] = *(const bf16x8*)(W0s + swz64((unsigned)(r * 64 + fc * 16))); }

    unsigned long long clk[6] = {0, 0, 0, 0, 0, 0};
#define FE_STAMP(i) if (p.clk) { const unsigned long long now = __builtin_amdgcn_s_memtime(); clk[i] += now - last; last = now; }
    unsigned long long last = p.clk ? __builtin_amdgcn_s_memtime() : 0ull;
    for (; tile < num_tiles; tile += G) {
        int t = tile;
        const int tw = t % tiles_w; t /= tiles_w;
        const int th = t % tiles_h;
        const int b = t / tiles_h;
        // ---- A: im2col rows (from the registers loaded one tile ago), then prefetch the next tile's bytes --------------------
        fe_row(ra, asx, Ib, tid);
        Vs[tid] = alive ? 1 : 0;
        if (two) { fe_row(rb, bsx, Ib, tid + 512); Vs[tid + 512] = blive ? 1 : 0; }
        else if (tid < FE_HPAD - 512) Vs[tid + 512] = 0;
        {
            const int nt = tile + G;
            alive = patch_px(nt, tid, ab_, asy, asx);
            fe_load(p, irs, ab_, asy, asx, alive, ra);
            if (two) { blive = patch_px(nt, tid + 512, bb_, bsy, bsx); fe_load(p, irs, bb_, bsy, bsx, blive, rb); }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        FE_STAMP(0)
        __builtin_amdgcn_s_barrier();
        FE_STAMP(5)
        // ---- B: stem GEMM + SiLU into the halo image ------------------------------------------------------------------------
        // (36 pixel fragments over 8 waves: four each as one batch - all operand reads, then the 8 MFMAs, then the epilogues, so the
        //  LDS / MFMA / transcendental latencies overlap across fragments instead of adding up per fragment - and a fifth for waves 0-3)
        auto stem_frags = [&](auto NFC, int f0) {
            constexpr int NF = decltype(NFC)::value;
            bf16x8 xf[NF];
            bool in[NF];
#pragma unroll
            for (int j = 0; j < NF; ++j) {
                const int hp = (f0 + j * FE_NW) * 16 + fr;
                xf[j] = *(const bf16x8*)(Ib + swz64((unsigned)(hp * 64 + fc * 16)));
                in[j] = Vs[hp] != 0;                      // else model.1's zero padding
            }
            f32x4 acc[NF][2];
#pragma unroll
            for (int j = 0; j < NF; ++j)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    acc[j][h] = f32x4{bias0[h].x, bias0[h].y, bias0[h].z, bias0[h].w};
                    acc[j][h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0f[h], xf[j], acc[j][h], 0, 0, 0);
                }
#pragma unroll
            for (int j = 0; j < NF; ++j) {
                const int hp = (f0 + j * FE_NW) * 16 + fr;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    __attribute__((aligned(8))) __bf16 o[4];
                    float sv[4] = {acc[j][h][0], acc[j][h][1], acc[j][h][2], acc[j][h][3]};
                    if (p.act0 == ACT_SILU) silu4(sv);
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] = (__bf16)(in[j] ? sv[i] : 0.f);
                    if (hp < FE_HP) *(uint2*)(Hs + hp * 64 + (((2 * h + (fc >> 1)) ^ fswz(hp)) * 16) + (fc & 1) * 8) = *(const uint2*)o;
                }
            }
        };
        stem_frags(std::integral_constant<int, 4>{}, wave);
        if (wave < FE_HPAD / 16 - 4 * FE_NW) stem_frags(std::integral_constant<int, 1>{}, wave + 4 * FE_NW);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        FE_STAMP(1)
        __builtin_amdgcn_s_barrier();
        FE_STAMP(5)
        // ---- C: model.1 (3x3 s2 over the E/O-plane image) ---------------------------------------------------------------------
        f32x4 acc[FN][FM];
#pragma unroll
        for (int a = 0; a < FN; ++a)
#pragma unroll
            for (int r = 0; r < FM; ++r) acc[a][r] = f32x4{bias1[a][0], bias1[a][1], bias1[a][2], bias1[a][3]};
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            bf16x8 wf[3][FN];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int a = 0; a < FN; ++a) {
                    const int rw = (ky * 3 + kx) * BN + wn * (FN * 16) + a * 16 + fr;
                    wf[ky][a] = *(const bf16x8*)(W1s + swz64((unsigned)(rw * 64 + fc * 16)));
                }
            const int eoff = (kx == 1) ? 17 + fr : fr + (kx >> 1);
#pragma unroll
            for (int jj = 0; jj < 2 * FM + 1; ++jj) {
                const int hp = (2 * wm * FM + jj) * 33 + eoff;
                const bf16x8 xf = *(const bf16x8*)(Hs + swz64((unsigned)(hp * 64 + fc * 16)));
                if (jj & 1) {
#pragma unroll
                    for (int a = 0; a < FN; ++a) acc[a][jj >> 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1][a], xf, acc[a][jj >> 1], 0, 0, 0);
                } else {
                    if ((jj >> 1) < FM) {
#pragma unroll
                        for (int a = 0; a < FN; ++a) acc[a][jj >> 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][a], xf, acc[a][jj >> 1], 0, 0, 0);
                    }
                    if ((jj >> 1) >= 1) {
#pragma unroll
                        for (int a = 0; a < FN; ++a)
                            acc[a][(jj >> 1) - 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[2][a], xf, acc[a][(jj >> 1) - 1], 0, 0, 0);
                    }
                }
            }
        }
        // ---- D: model.2.cv1 on the tile (intermediate through the halo image's memory) ------------------------------------------
        FE_STAMP(2)
        __builtin_amdgcn_s_barrier();                     // every wave is done reading the halo image
        FE_STAMP(5)
#pragma unroll
        for (int r = 0; r < FM; ++r) {
            const int px = (wm * FM + r) * 16 + fr;
#pragma unroll
            for (int a = 0; a < FN; ++a) {
                const int co = wn * (FN * 16) + a * 16 + fc * 4;
                __attribute__((aligned(8))) __bf16 o[4];
                float sv[4] = {acc[a][r][0], acc[a][r][1], acc[a][r][2], acc[a][r][3]};
                if (p.act1 == ACT_SILU) silu4(sv);
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = (__bf16)sv[i];
                *(uint2*)(Hs + px * 128 + (((co >> 3) ^ ((px >> 1) & 7)) * 16) + (co & 7) * 2) = *(const uint2*)o;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        FE_STAMP(3)
        __builtin_amdgcn_s_barrier();
        FE_STAMP(5)
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
                t2f[r] = *(const bf16x8*)(Hs + px * 128 + (((ss * 4 + fc) ^ ((px >> 1) & 7)) * 16));
            }
#pragma unroll
            for (int a = 0; a < FN; ++a)
#pragma unroll
                for (int r = 0; r < FM; ++r) acc[a][r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2f[a], t2f[r], acc[a][r], 0, 0, 0);
        }
        // ---- E: stores ---------------------------------------------------------------------------------------------------------
        const int wo = tw * 16 + fr;
#pragma unroll
        for (int r = 0; r < FM; ++r) {
            const int ho = th * FE_TH + wm * FM + r;
            const bool pix_ok = (ho < p.Ho) && (wo < p.Wo);
            const unsigned m = (unsigned)((b * p.Ho + ho) * p.Wo + wo);
#pragma unroll
            for (int a = 0; a < FN; ++a) {
                const int co = wn * (FN * 16) + a * 16 + fc * 4;
                float v[4] = {acc[a][r][0], acc[a][r][1], acc[a][r][2], acc[a][r][3]};
                if (p.act2 == ACT_SILU) silu4(v);
                const unsigned off = pix_ok ? (m * (unsigned)p.y_stride + (unsigned)(p.y_coff + co)) * 2u : OOB;
                __attribute__((aligned(8))) __bf16 o[4] = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                __builtin_amdgcn_raw_buffer_store_b64(*(const __attribute__((ext_vector_type(2))) unsigned*)o, yrs, off, 0, 0);
            }
        }
        FE_STAMP(4)
        // (the next tile's stage A writes Ib, which stage B of this tile finished with two barriers ago; its stage B writes the
        //  halo image only after the barrier that follows stage A, which every wave reaches after its stage-D reads)
    }
    if (p.clk && lane == 0)
        for (int i = 0; i < 6; ++i) p.clk[((size_t)blockIdx.x * FE_NW + wave) * 6 + i] = clk[i];
}

static size_t frontend_lds() { return (size_t)2 * FE_HB + 9 * 64 * 64 + 64 * 128 + 32 * 64 + 1024; }

bool frontend_valid(const FrontParams& p) {
    if (p.C0 != 32 || p.C1 != 64 || p.C2 != 64 || p.Kpad1 != 9 * 32 || p.Kpad2 != 64) return false;
    if ((p.imgH & 3) || (p.imgW & 3) || p.H1 * 2 != p.imgH || p.W1 * 2 != p.imgW || p.Ho * 2 != p.H1 || p.Wo * 2 != p.W1) return false;
    if ((p.y_stride & 3) || (p.y_coff & 3) || p.y_bytes >= (1ull << 31)) return false;
    if ((long)p.B * p.imgH * p.imgW * 3 >= (1l << 31)) return false;
    const long covered = (long)((p.Ho + FE_TH - 1) / FE_TH * FE_TH) * ((p.Wo + 15) / 16 * 16);
    if (covered * 2 > (long)p.Ho * p.Wo * 3) return false;
    return true;
}

hipError_t launch_frontend(const FrontParams& p, hipStream_t st) {
    const size_t sh = frontend_lds();
    const int tiles_h = (p.Ho + FE_TH - 1) / FE_TH, tiles_w = (p.Wo + 15) / 16;
    const int num_tiles = p.B * tiles_h * tiles_w;
    int G = 256;
    if (G > num_tiles) G = num_tiles;
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)frontend_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
        if (e != hipSuccess) return e;
        attr = true;
    }
    static const bool clocks = [] { const char* v = std::getenv("YOLOP_FRONT_CLOCKS"); return v && *v == '1'; }();   // debug: per-stage s_memtime sums
    if (clocks) {
        FrontParams q = p;
        const size_t n = (size_t)G * FE_NW * 6;
        if (hipMalloc((void**)&q.clk, n * 8) != hipSuccess) return hipErrorOutOfMemory;
        hipLaunchKernelGGL(frontend_kernel, dim3(G), dim3(FE_NW * 64), sh, st, q, tiles_h, tiles_w, G);
        (void)hipStreamSynchronize(st);
        std::vector<unsigned long long> h(n);
        (void)hipMemcpy(h.data(), q.clk, n * 8, hipMemcpyDeviceToHost);
        (void)hipFree(q.clk);
        static const char* nm[6] = {"A im2col", "B stem", "C 3x3s2", "D silu->lds", "E 1x1+store", "barriers"};
        const double tiles_per = (double)num_tiles / G;
        for (int w = 0; w < FE_NW; w += FE_NW - 1) {
            fprintf(stderr, "[frontend clocks] wave %d, s_memtime ticks per tile:", w);
            for (int i = 0; i < 6; ++i) {
                double s = 0;
                for (int g = 0; g < G; ++g) s += (double)h[((size_t)g * FE_NW + w) * 6 + i];
                fprintf(stderr, " %s %.0f", nm[i], s / G / tiles_per);
            }
            fprintf(stderr, "\n");
        }
        return hipGetLastError();
    }
    hipLaunchKernelGGL(frontend_kernel, dim3(G), dim3(FE_NW * 64), sh, st, p, tiles_h, tiles_w, G);
    return hipGetLastError();
}

}  // namespace yp
