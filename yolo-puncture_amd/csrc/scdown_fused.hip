// SCDown (1x1 conv + SiLU -> depthwise 3x3 stride 2) as ONE persistent kernel for K = 128 input channels (bf16).
// SURVEY.md A.3 layer 5 [U] (SCDown = Conv(c1, c2, 1, 1) -> Conv(c2, c2, k=3, s=2, g=c2, act=False)), run inside `.predict`
// (reference yolo_seg/app.py:91).
//
// Unfused, the 1x1 writes B x 80 x 80 x 256 bf16 (105 MB at batch 32) that the depthwise conv reads straight back: 157 + 131 MB
// in two launches of 52 + 30 us. Here a workgroup owns a 4x8 tile of the block's OUTPUT and all its channels:
//   DMA  the 9x17 input patch (128 ch = 256-B rows, zero outside the frame), one tile ahead; the 1x1 weights of the next
//        64-channel group, one group ahead
//   per group of 64 output channels:
//     G   t = SiLU(W1g . x + b1) on the patch (MFMA; a wave = one 16-channel fragment x five 16-pixel fragments), zeroed where
//         the patch pixel lies outside the frame (= the depthwise conv's padding), bf16 -> LDS
//     D   depthwise 3x3 s2 (+ bias) on fp32 VALU, a thread = one output pixel x 4 channels, bf16 stores
//   (D of group g runs one barrier later, next to G of group g+1, on the other of two t buffers)
// Rounding points are those of the two separate kernels (t is rounded to bf16 exactly where the unfused graph stores it).
#include "common.h"
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace yp {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void lds_void;

constexpr int SD_NW = 8;
constexpr int SD_TH = 4, SD_TW = 8;                     // output tile
constexpr int SD_RH = 2 * SD_TH + 1, SD_RW = 2 * SD_TW + 1;   // 9 x 17 patch of the 1x1's output / input
constexpr int SD_RP = SD_RH * SD_RW;                    // 153 pixels
constexpr int SD_XP = 39;                               // patch pieces (4 pixels x 256 B each): 156 pixel rows
constexpr int SD_XB = SD_XP * 1024;
constexpr int SD_TB = 160 * 128;                        // t: [160 px][64 ch]
constexpr int SD_WB = 64 * 256;                         // one group's 1x1 weights [64 co][128 k]
constexpr int SD_LDS = 2 * SD_XB + 2 * SD_TB + 2 * SD_WB + 5120 + 2048;   // 160768 B

// (LDS accesses next to in-flight LDS-DMA go through inline asm: see conv_dwpw.hip)
__device__ __forceinline__ void sd_write8(unsigned char* dst, unsigned long long v) {
    asm volatile("ds_write_b64 %0, %1" ::"v"((unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)dst), "v"(v) : "memory");
}

__global__ __launch_bounds__(SD_NW * 64) void scdown_fused_kernel(const ScdParams p, const int tiles_h, const int tiles_w, const int G) {
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char* const Xs = smem;                           // 2 x [156 px][128 ch]
    unsigned char* const Ts = Xs + 2 * SD_XB;                 // 2 x [160 px][64 ch]
    unsigned char* const Ws = Ts + 2 * SD_TB;                 // 2 x [64 co][128 k]
    unsigned char* const Wd = Ws + 2 * SD_WB;                 // depthwise weights [9][C] bf16 (C <= 256)
    float* const Bs = (float*)(Wd + 5120);                    // bias1[C] | biasd[C]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fc = lane >> 4;
    const int num_tiles = p.B * tiles_h * tiles_w;
    const int ngroups = p.C >> 6;

    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, j = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }

    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w1, 0, (int)p.w1_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, (int)p.y_bytes, 0x00020000);

    auto issue_x = [&](int tile, unsigned char* dst) {            // piece ii = 4 patch pixels x 256 B
        int t = tile;
        const int tw = t % tiles_w; t /= tiles_w;
        const int th = t % tiles_h;
        const int b = t / tiles_h;
        const bool tv = tile < num_tiles;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int ii = wave + k * SD_NW;
            if (ii < SD_XP) {
                const int s = ii * 64 + lane;
                const int q = s >> 4, pc = s & 15;
                const int c = pc ^ (q & 15);
                const int py = q / SD_RW, px = q - py * SD_RW;
                const int iy = 2 * th * SD_TH - 1 + py, ix = 2 * tw * SD_TW - 1 + px;
                const bool ok = tv && q < SD_RP && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                const unsigned voff = ok ? (unsigned)((((b * p.H + iy) * p.W + ix) * p.x_stride + p.x_coff + c * 8) * 2) : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_void*)(dst + ii * 1024), 16, voff, 0, 0, 0);
            }
        }
    };
    auto issue_w = [&](int g, unsigned char* dst) {               // 16 pieces: rows g*64 .. g*64+63 of the packed 1x1 weights
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int ii = wave + k * SD_NW;
            const int s = ii * 64 + lane;
            const int row = s >> 4, pc = s & 15;
            const int c = pc ^ (row & 15);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_void*)(dst + ii * 1024), 16, (unsigned)(((g * 64 + row) * p.Kpad1 + c * 8) * 2), 0, 0, 0);
        }
    };

    // ---- small resident operands (ordinary loads / stores, before any DMA is in flight) -----------------------------------------
    for (int i = tid; i < 9 * p.C / 2; i += SD_NW * 64) ((unsigned*)Wd)[i] = ((const unsigned*)p.wd)[i];
    for (int i = tid; i < p.C; i += SD_NW * 64) { Bs[i] = p.bias1[i]; Bs[256 + i] = p.biasd[i]; }

    int tile = bid;
    issue_x(tile, Xs);
    issue_w(0, Ws);
    __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (0xF << 8));               // vmcnt(0)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    const int cf = wave & 3, ps = wave >> 2;          // G stage: channel fragment of the group, pixel-fragment parity
    const int o = tid >> 4, cg = tid & 15;            // D stage: output pixel of the tile, 4-channel group
    const int oy = o >> 3, ox = o & 7;

    unsigned long long clk[4] = {0, 0, 0, 0};
#define SD_STAMP(i) if (p.clk) { const unsigned long long now = __builtin_amdgcn_s_memtime(); clk[i] += now - last; last = now; }
    unsigned long long last = p.clk ? __builtin_amdgcn_s_memtime() : 0ull;
    // D of one channel group (reads the t buffer `T`, stores to the output tile (pb, pth, ptw))
    auto dw_stage = [&](const unsigned char* T, int g, int pb, int pth, int ptw) {
        const int ch = cg * 4;                   // channel of the group
        const f32x4 bd = *(const f32x4*)(Bs + 256 + g * 64 + ch);
        float a0 = bd[0], a1 = bd[1], a2 = bd[2], a3 = bd[3];
        unsigned long long tv[9], wv[9];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int qq = (2 * oy + tap / 3) * SD_RW + 2 * ox + tap % 3;
            asm volatile("ds_read_b64 %0, %1" : "=v"(tv[tap]) : "v"((unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)(T + qq * 128 + (((ch >> 3) ^ ((qq >> 1) & 7)) * 16) + (ch & 7) * 2)) : "memory");
            asm volatile("ds_read_b64 %0, %1" : "=v"(wv[tap]) : "v"((unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)(Wd + (tap * p.C + g * 64 + ch) * 2)) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(tv[0]), "+v"(tv[1]), "+v"(tv[2]), "+v"(tv[3]), "+v"(tv[4]), "+v"(tv[5]), "+v"(tv[6]), "+v"(tv[7]), "+v"(tv[8]),
                     "+v"(wv[0]), "+v"(wv[1]), "+v"(wv[2]), "+v"(wv[3]), "+v"(wv[4]), "+v"(wv[5]), "+v"(wv[6]), "+v"(wv[7]), "+v"(wv[8])::"memory");
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const unsigned tl = (unsigned)tv[tap], thh = (unsigned)(tv[tap] >> 32), wl = (unsigned)wv[tap], wh = (unsigned)(wv[tap] >> 32);
            a0 = fmaf(__uint_as_float(tl << 16), __uint_as_float(wl << 16), a0);
            a1 = fmaf(__uint_as_float(tl & 0xffff0000u), __uint_as_float(wl & 0xffff0000u), a1);
            a2 = fmaf(__uint_as_float(thh << 16), __uint_as_float(wh << 16), a2);
            a3 = fmaf(__uint_as_float(thh & 0xffff0000u), __uint_as_float(wh & 0xffff0000u), a3);
        }
        float v[4] = {a0, a1, a2, a3};
        if (p.actd == ACT_SILU) silu4_packed(v);
        const int ho = pth * SD_TH + oy, wo = ptw * SD_TW + ox;
        const bool ok = ho < p.Ho && wo < p.Wo;
        const unsigned off = ok ? ((unsigned)((pb * p.Ho + ho) * p.Wo + wo) * (unsigned)p.y_stride + (unsigned)(p.y_coff + g * 64 + ch)) * 2u : OOB;
        __attribute__((aligned(8))) __bf16 ob[4] = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
        __builtin_amdgcn_raw_buffer_store_b64(*(const __attribute__((ext_vector_type(2))) unsigned*)ob, yrs, off, 0, 0);
    };

    // Skewed by one group: between two barriers a wave runs D of the previous group (VALU) and G of this one (MFMA + SiLU) back to
    // back on the two t buffers - one barrier per group, and the two stage kinds of different waves overlap on a SIMD.
    int gi = 0;                                   // running group index: selects the W and t buffers
    int pg = -1, pb = 0, pth = 0, ptw = 0;        // the group whose D stage is pending
    for (int it = 0; tile < num_tiles; tile += G, ++it) {
        int t = tile;
        const int tw = t % tiles_w; t /= tiles_w;
        const int th = t % tiles_h;
        const int b = t / tiles_h;
        const unsigned char* const X = Xs + (it & 1) * SD_XB;
        for (int g = 0; g < ngroups; ++g, ++gi) {
            // W(g) (and at g == 0 the patch) were issued one group ago; only that iteration's single store is younger
            if (gi) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            SD_STAMP(0)
            __builtin_amdgcn_s_barrier();                 // operands and t(gi-1) complete; every wave is past D(gi-2) and G(gi-1)
            SD_STAMP(1)
            const int wsel = gi & 1;
            if (g + 1 < ngroups) issue_w(g + 1, Ws + (wsel ^ 1) * SD_WB);
            else { issue_w(0, Ws + (wsel ^ 1) * SD_WB); issue_x(tile + G, Xs + ((it & 1) ^ 1) * SD_XB); }
            const unsigned char* const Wg = Ws + wsel * SD_WB;
            unsigned char* const Tg = Ts + wsel * SD_TB;
            if (pg >= 0) dw_stage(Ts + (wsel ^ 1) * SD_TB, pg, pb, pth, ptw);
            else { const unsigned zero[2] = {0u, 0u}; __builtin_amdgcn_raw_buffer_store_b64(*(const __attribute__((ext_vector_type(2))) unsigned*)zero, yrs, OOB, 0, 0); }   // (keeps the store count)
            SD_STAMP(3)

            // ---- G: t = act(W1g . x) on the patch ----------------------------------------------------------------------------------
            {
                bf16x8 wA[4], xB[2][5];
                int q[5];
#pragma unroll
                for (int kc = 0; kc < 4; ++kc) { const int row = cf * 16 + fr; wA[kc] = *(const bf16x8*)(Wg + row * 256 + (((kc * 4 + fc) ^ (row & 15)) * 16)); }
#pragma unroll
                for (int j = 0; j < 5; ++j) { const int qq = (ps + 2 * j) * 16 + fr; q[j] = qq < SD_XP * 4 ? qq : SD_XP * 4 - 1; }
                auto rd = [&](int kc) {
#pragma unroll
                    for (int j = 0; j < 5; ++j) xB[kc & 1][j] = *(const bf16x8*)(X + q[j] * 256 + (((kc * 4 + fc) ^ (q[j] & 15)) * 16));
                };
                f32x4 acc[5];
                const f32x4 b1 = *(const f32x4*)(Bs + g * 64 + cf * 16 + fc * 4);
#pragma unroll
                for (int j = 0; j < 5; ++j) acc[j] = b1;
                rd(0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kc = 0; kc < 4; ++kc) {
                    if (kc < 3) rd(kc + 1);
#pragma unroll
                    for (int j = 0; j < 5; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wA[kc], xB[kc & 1][j], acc[j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    const int qq = (ps + 2 * j) * 16 + fr;
                    const int py = q[j] / SD_RW, px = q[j] - py * SD_RW;
                    const int iy = 2 * th * SD_TH - 1 + py, ix = 2 * tw * SD_TW - 1 + px;
                    const bool in = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;      // else the depthwise conv's zero padding
                    float v[4] = {acc[j][0], acc[j][1], acc[j][2], acc[j][3]};
                    if (p.act1 == ACT_SILU) silu4_packed(v);
                    __attribute__((aligned(8))) __bf16 ob[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) ob[i] = (__bf16)(in ? v[i] : 0.f);
                    const int ch = cf * 16 + fc * 4;
                    sd_write8(Tg + qq * 128 + (((ch >> 3) ^ ((qq >> 1) & 7)) * 16) + (ch & 7) * 2, *(const unsigned long long*)ob);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            SD_STAMP(2)
            pg = g; pb = b; pth = th; ptw = tw;
        }
    }
    __builtin_amdgcn_s_barrier();                         // t of the last group is complete
    if (pg >= 0) dw_stage(Ts + ((gi - 1) & 1) * SD_TB, pg, pb, pth, ptw);
    __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (0xF << 8));               // the look-ahead DMA (out of range past the last tile) and the stores
    if (p.clk && lane == 0)
        for (int i = 0; i < 4; ++i) p.clk[((size_t)blockIdx.x * SD_NW + wave) * 4 + i] = clk[i];
}

const char* scdown_fused_kernel_name(const ScdParams& p) { return scdown_stream_valid(p) ? scdown_stream_kernel_name(p) : "scdown_fused_kernel"; }

bool scdown_fused_valid(const ScdParams& p) {
    if (scdown_stream_valid(p)) return true;
    if (p.K != 128 || p.Kpad1 != 128 || (p.C & 63) || p.C > 256 || p.C < 64) return false;
    if ((p.x_stride & 7) || (p.x_coff & 7) || (p.y_stride & 3) || (p.y_coff & 3)) return false;
    if (p.x_bytes >= (1ull << 31) || p.y_bytes >= (1ull << 31)) return false;
    if (p.w1_bytes < (size_t)p.C * p.Kpad1 * 2) return false;
    if ((p.H & 1) || (p.W & 1) || p.Ho * 2 != p.H || p.Wo * 2 != p.W) return false;
    const long covered = (long)((p.Ho + SD_TH - 1) / SD_TH * SD_TH) * ((p.Wo + SD_TW - 1) / SD_TW * SD_TW);
    if (covered * 2 > (long)p.Ho * p.Wo * 3) return false;               // (tiny maps: the separate kernels waste less)
    return true;
}

hipError_t launch_scdown_fused(const ScdParams& p, hipStream_t st) {
    if (scdown_stream_valid(p)) return launch_scdown_stream(p, st);
    const int tiles_h = (p.Ho + SD_TH - 1) / SD_TH, tiles_w = (p.Wo + SD_TW - 1) / SD_TW;
    const int num_tiles = p.B * tiles_h * tiles_w;
    int G = 256;
    if (G > num_tiles) G = num_tiles;
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)scdown_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
        if (e != hipSuccess) return e;
        attr = true;
    }
    static const bool clocks = [] { const char* v = std::getenv("YOLOP_SCD_CLOCKS"); return v && *v == '1'; }();   // debug: per-stage s_memtime sums
    if (clocks) {
        ScdParams q = p;
        const size_t n = (size_t)G * SD_NW * 4;
        if (hipMalloc((void**)&q.clk, n * 8) != hipSuccess) return hipErrorOutOfMemory;
        hipLaunchKernelGGL(scdown_fused_kernel, dim3(G), dim3(SD_NW * 64), (size_t)SD_LDS, st, q, tiles_h, tiles_w, G);
        (void)hipStreamSynchronize(st);
        std::vector<unsigned long long> h(n);
        (void)hipMemcpy(h.data(), q.clk, n * 8, hipMemcpyDeviceToHost);
        (void)hipFree(q.clk);
        static const char* nm[4] = {"wait", "barrier", "G (1x1+SiLU)", "issue+D (dw)"};
        const double steps = (double)num_tiles / G * (p.C / 64);
        for (int w = 0; w < SD_NW; w += SD_NW - 1) {
            fprintf(stderr, "[scdown clocks] wave %d, s_memtime ticks per channel group:", w);
            for (int i = 0; i < 4; ++i) {
                double s = 0;
                for (int g = 0; g < G; ++g) s += (double)h[((size_t)g * SD_NW + w) * 4 + i];
                fprintf(stderr, " %s %.0f", nm[i], s / G / steps);
            }
            fprintf(stderr, "\n");
        }
        return hipGetLastError();
    }
    hipLaunchKernelGGL(scdown_fused_kernel, dim3(G), dim3(SD_NW * 64), (size_t)SD_LDS, st, p, tiles_h, tiles_w, G);
    return hipGetLastError();
}

}  // namespace yp
