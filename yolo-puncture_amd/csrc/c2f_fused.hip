// One C2f block (n = 1, 32-channel bottleneck) behind its first 1x1 as ONE persistent kernel (bf16):
//   [a | b] (64 ch)  ->  t = SiLU(conv3x3(b))  ->  c = SiLU(conv3x3(t)) (+ b)  ->  y = SiLU(conv1x1([a | b | c]))
// SURVEY.md A.3 layer 2 [U] (C2f = cv1 -> Bottleneck(3x3, 3x3, shortcut) -> cv2 over the concat), run inside `.predict`
// (reference yolo_seg/app.py:91).
//
// Unfused, the two 3x3 convs and the trailing 1x1 move 0.52 GB at 160x160x32 frames (t and c are written and read back, the
// concat is re-read) in three launches of 41 + 43 + 68 us. Here a workgroup owns an 8x16 tile of the block's output:
//   DMA  the 12x20 pixel patch of [a | b] (128-B rows; zero outside the frame = the first conv's padding), one tile ahead
//   S2   t on the 10x18 ring the second conv needs (MFMA, weights resident), zero outside the frame (= its padding), -> LDS
//   S3   c on the 8x16 tile (+ b from the patch), -> LDS
//   S4   y = W3 . [a | b | c], bias + SiLU, bf16 stores
// Rounding points are those of the three separate kernels: t, c and y are each rounded to bf16 exactly where the unfused graph
// stores them, so the result differs from the unfused engine only by fp32 summation order inside a stage.
#include "common.h"
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace yp {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void lds_void;

// LDS stores behind the compiler's back: it cannot tell a ds_write from the in-flight LDS-DMA of the next patch apart and would
// drain vmcnt to 0 in front of every one of them (the patch buffers and Ts / Cs never overlap)
__device__ __forceinline__ void lds_write8(unsigned char* dst, unsigned long long v) {
    asm volatile("ds_write_b64 %0, %1" ::"v"((unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)dst), "v"(v) : "memory");
}
// (same for the 8-byte residual read of the current patch; the caller waits on lgkmcnt before the first use)
__device__ __forceinline__ unsigned long long lds_read8_async(const unsigned char* src) {
    unsigned long long v;
    asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"((unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)src) : "memory");
    return v;
}
__device__ __forceinline__ int cswz(int row) { return ((row >> 2) & 1) << 1; }     // 64-B rows
__device__ __forceinline__ int cswz128(int row) { return (row >> 1) & 7; }         // 128-B rows

constexpr int CF_NW = 8;
constexpr int CF_TH = 8;                          // output rows per tile (x 16 columns)
constexpr int CF_PW = 20, CF_PP = 12 * 20;        // [a | b] patch: 12 x 20 pixels
constexpr int CF_TW = 18, CF_TP = 10 * 18;        // t ring: 10 x 18 pixels
constexpr int CF_AB = CF_PP * 128;                // 30 pieces of 1 KB
constexpr int CF_TS = 192 * 64, CF_CS = 128 * 64;
constexpr int CF_W33 = 9 * 32 * 64, CF_W3 = 3 * 64 * 64;
constexpr int CF_LDS = 2 * CF_AB + CF_TS + CF_CS + 2 * CF_W33 + CF_W3 + 512;    // 131584 (+ the three bias vectors)

__global__ __launch_bounds__(CF_NW * 64) void c2f_fused_kernel(const C2fParams p, const int tiles_h, const int tiles_w, const int G) {
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char* const ABs = smem;                          // 2 x [240 px][64 ch]   (a = ch 0..31, b = ch 32..63)
    unsigned char* const Ts = ABs + 2 * CF_AB;                // [180 px][32 ch]
    unsigned char* const Cs = Ts + CF_TS;                     // [128 px][32 ch]
    unsigned char* const W1s = Cs + CF_CS;                    // [9 taps][32 co][32 ci]
    unsigned char* const W2s = W1s + CF_W33;                  // [9 taps][32 co][32 ci]
    unsigned char* const W3s = W2s + CF_W33;                  // [3 chunks][64 co][32 ci]
    float* const Bs = (float*)(W3s + CF_W3);                  // bias1[32] | bias2[32] | bias3[64]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fc = lane >> 4;
    const int num_tiles = p.B * tiles_h * tiles_w;

    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, j = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }

    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w1rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w1, 0, (int)p.w1_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w2rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w2, 0, (int)p.w2_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w3rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w3, 0, (int)p.w3_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, (int)p.y_bytes, 0x00020000);

    // ---- the patch of one tile: piece ii = 8 pixels x 128 B; waves 6 and 7 carry three pieces, the others four --------------
    auto issue_ab = [&](int tile, unsigned char* dst) {
        int t = tile;
        const int tw = t % tiles_w; t /= tiles_w;
        const int th = t % tiles_h;
        const int b = t / tiles_h;
        const bool tv = tile < num_tiles;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ii = wave + k * CF_NW;
            if (ii < CF_AB / 1024) {
                const int s = ii * 64 + lane;
                const int row = s >> 3, pc = s & 7;
                const int c8 = pc ^ cswz128(row);
                const int py = row / CF_PW, px = row - py * CF_PW;
                const int iy = th * CF_TH - 2 + py, ix = tw * 16 - 2 + px;
                const bool ok = tv && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                const unsigned voff = ok ? (unsigned)((((b * p.H + iy) * p.W + ix) * p.x_stride + p.x_coff + c8 * 8) * 2) : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_void*)(dst + ii * 1024), 16, voff, 0, 0, 0);
            }
        }
    };

    // ---- resident weights ----------------------------------------------------------------------------------------------------
    for (int ii = wave; ii < CF_W33 / 1024; ii += CF_NW) {               // row rg = tap*32 + co
        const int s = ii * 64 + lane;
        const int rg = s >> 2, pc = s & 3;
        const int c8 = pc ^ cswz(rg);
        const int n = rg & 31, tap = rg >> 5;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w1rs, (lds_void*)(W1s + ii * 1024), 16, (unsigned)((n * p.Kpad1 + tap * 32 + c8 * 8) * 2), 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w2rs, (lds_void*)(W2s + ii * 1024), 16, (unsigned)((n * p.Kpad2 + tap * 32 + c8 * 8) * 2), 0, 0, 0);
    }
    for (int ii = wave; ii < CF_W3 / 1024; ii += CF_NW) {                // row rg = chunk*64 + co
        const int s = ii * 64 + lane;
        const int rg = s >> 2, pc = s & 3;
        const int c8 = pc ^ cswz(rg);
        const int n = rg & 63, ch = rg >> 6;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w3rs, (lds_void*)(W3s + ii * 1024), 16, (unsigned)((n * p.Kpad3 + ch * 32 + c8 * 8) * 2), 0, 0, 0);
    }

    const int cf = wave & 1, wq = wave >> 1;          // stages 2 and 3: channel half, pixel-fragment group
    if (tid < 128) Bs[tid] = tid < 32 ? p.bias1[tid] : tid < 64 ? p.bias2[tid - 32] : p.bias3[tid - 64];
    const float* const bias1 = Bs + cf * 16 + fc * 4;          // (read per stage: the registers are needed for the weight fragments)
    const float* const bias2 = Bs + 32 + cf * 16 + fc * 4;
    const float* const bias3 = Bs + 64 + cf * 32 + fc * 4;

    int tile = bid;
    issue_ab(tile, ABs);
    __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (0xF << 8));               // vmcnt(0): weights, biases and the first patch

    // ---- this wave's weight fragments stay in registers for every tile: S2 / S3 one channel half of both 3x3 convs, (S4's six come from LDS per tile)
    //      (LDS bandwidth bounds the kernel: 1 KB per fragment read, 128 B/clk per CU) ---------------
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                        // (every wave's weight pieces and the bias vectors have landed)
    bf16x8 wf1[9], wf2[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int rw = tap * 32 + cf * 16 + fr;
        wf1[tap] = *(const bf16x8*)(W1s + swz64((unsigned)(rw * 64 + fc * 16)));
        wf2[tap] = *(const bf16x8*)(W2s + swz64((unsigned)(rw * 64 + fc * 16)));
    }

    unsigned long long clk[7] = {0, 0, 0, 0, 0, 0, 0};
#define CF_STAMP(i) if (p.clk) { const unsigned long long now = __builtin_amdgcn_s_memtime(); clk[i] += now - last; last = now; }
    unsigned long long last = p.clk ? __builtin_amdgcn_s_memtime() : 0ull;
    for (int it = 0; tile < num_tiles; tile += G, ++it) {
        int t = tile;
        const int tw = t % tiles_w; t /= tiles_w;
        const int th = t % tiles_h;
        const int b = t / tiles_h;
        unsigned char* const AB = ABs + (it & 1) * CF_AB;
        if (it) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");         // this tile's patch (only the previous tile's 4 stores are younger)
        CF_STAMP(0)
        __builtin_amdgcn_s_barrier();                                    // patch complete; every wave is past the previous tile's stage 4
        CF_STAMP(1)
        issue_ab(tile + G, ABs + ((it & 1) ^ 1) * CF_AB);

        // ---- S2: t = act(W1 * b) on the 10x18 ring ------------------------------------------------------------------------------
        // (tap by tap over independent accumulators, the reads three taps ahead of their MFMAs; left alone, the scheduler sinks every
        //  read next to its MFMA: one LDS latency per MFMA)
        {
            constexpr int D = 3;                          // read-ahead in taps
            bf16x8 xf[D + 1][3];
            int ty[3], tx[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int q = (wq + 4 * j) * 16 + fr;
                const int qc = q < CF_TP ? q : CF_TP - 1;
                ty[j] = qc / CF_TW; tx[j] = qc - ty[j] * CF_TW;
            }
            auto rd = [&](int tap) {
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int pp = (ty[j] + tap / 3) * CF_PW + tx[j] + tap % 3;
                    xf[tap % (D + 1)][j] = *(const bf16x8*)(AB + pp * 128 + (((4 + fc) ^ cswz128(pp)) * 16));
                }
            };
            f32x4 acc[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[j] = *(const f32x4*)bias1;
#pragma unroll
            for (int tap = 0; tap < D; ++tap) rd(tap);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                if (tap + D < 9) rd(tap + D);
#pragma unroll
                for (int j = 0; j < 3; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf1[tap], xf[tap % (D + 1)][j], acc[j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int q = (wq + 4 * j) * 16 + fr;
                const int iy = th * CF_TH - 1 + ty[j], ix = tw * 16 - 1 + tx[j];
                const bool in = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;    // else the second conv's zero padding
                float v[4] = {acc[j][0], acc[j][1], acc[j][2], acc[j][3]};
                if (p.act1 == ACT_SILU) silu4_packed(v);
                __attribute__((aligned(8))) __bf16 o[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = (__bf16)(in ? v[i] : 0.f);
                if (q < CF_TP) lds_write8(Ts + q * 64 + (((cf * 2 + (fc >> 1)) ^ cswz(q)) * 16) + (fc & 1) * 8, *(const unsigned long long*)o);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        CF_STAMP(2)
        __builtin_amdgcn_s_barrier();
        CF_STAMP(3)
        // ---- S3: c = act(W2 * t) (+ b) on the tile -------------------------------------------------------------------------------
        {
            constexpr int D = 4;
            bf16x8 xf[D + 1][2];
            unsigned long long rr[2] = {0ull, 0ull};
            auto rd = [&](int tap) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int tp = (wq + 4 * j + tap / 3) * CF_TW + fr + tap % 3;
                    xf[tap % (D + 1)][j] = *(const bf16x8*)(Ts + swz64((unsigned)(tp * 64 + fc * 16)));
                }
            };
            f32x4 acc[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[j] = *(const f32x4*)bias2;
#pragma unroll
            for (int tap = 0; tap < D; ++tap) rd(tap);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int pp = (wq + 4 * j + 2) * CF_PW + fr + 2;
                if (p.shortcut) rr[j] = lds_read8_async(AB + pp * 128 + (((4 + cf * 2 + (fc >> 1)) ^ cswz128(pp)) * 16) + (fc & 1) * 8);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                if (tap + D < 9) rd(tap + D);
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf2[tap], xf[tap % (D + 1)][j], acc[j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rr[0]), "+v"(rr[1])::"memory");      // the two asynchronous residual reads
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float v[4] = {acc[j][0], acc[j][1], acc[j][2], acc[j][3]};
                if (p.act2 == ACT_SILU) silu4_packed(v);
                if (p.shortcut) {
                    const unsigned lo = (unsigned)rr[j], hi = (unsigned)(rr[j] >> 32);
                    v[0] += __uint_as_float(lo << 16); v[1] += __uint_as_float(lo & 0xffff0000u);
                    v[2] += __uint_as_float(hi << 16); v[3] += __uint_as_float(hi & 0xffff0000u);
                }
                const int q = (wq + 4 * j) * 16 + fr;
                __attribute__((aligned(8))) __bf16 o[4] = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                lds_write8(Cs + q * 64 + (((cf * 2 + (fc >> 1)) ^ cswz(q)) * 16) + (fc & 1) * 8, *(const unsigned long long*)o);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        CF_STAMP(4)
        __builtin_amdgcn_s_barrier();
        CF_STAMP(5)
        // ---- S4: y = act(W3 * [a | b | c]); wave = two tile rows x two channel fragments -------------------------------------------
        {
            bf16x8 xf[2][3], wf3[3][2];
#pragma unroll
            for (int ch = 0; ch < 3; ++ch)
#pragma unroll
                for (int a = 0; a < 2; ++a) { const int rw = ch * 64 + (cf * 2 + a) * 16 + fr; wf3[ch][a] = *(const bf16x8*)(W3s + swz64((unsigned)(rw * 64 + fc * 16))); }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int r = wq + 4 * j;
                const int pp = (r + 2) * CF_PW + fr + 2, q = r * 16 + fr;
                xf[j][0] = *(const bf16x8*)(AB + pp * 128 + ((fc ^ cswz128(pp)) * 16));
                xf[j][1] = *(const bf16x8*)(AB + pp * 128 + (((4 + fc) ^ cswz128(pp)) * 16));
                xf[j][2] = *(const bf16x8*)(Cs + swz64((unsigned)(q * 64 + fc * 16)));
            }
            f32x4 acc[2][2];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int a = 0; a < 2; ++a) acc[j][a] = *(const f32x4*)(bias3 + a * 16);
#pragma unroll
            for (int ch = 0; ch < 3; ++ch)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int a = 0; a < 2; ++a) acc[j][a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf3[ch][a], xf[j][ch], acc[j][a], 0, 0, 0);
            const int wo = tw * 16 + fr;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int ho = th * CF_TH + wq + 4 * j;
                const bool pix_ok = (ho < p.H) && (wo < p.W);
                const unsigned m = (unsigned)((b * p.H + ho) * p.W + wo);
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    float v[4] = {acc[j][a][0], acc[j][a][1], acc[j][a][2], acc[j][a][3]};
                    if (p.act3 == ACT_SILU) silu4_packed(v);
                    const int co = (cf * 2 + a) * 16 + fc * 4;
                    const unsigned off = pix_ok ? (m * (unsigned)p.y_stride + (unsigned)(p.y_coff + co)) * 2u : OOB;
                    __attribute__((aligned(8))) __bf16 o[4] = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                    __builtin_amdgcn_raw_buffer_store_b64(*(const __attribute__((ext_vector_type(2))) unsigned*)o, yrs, off, 0, 0);
                }
            }
        }
        CF_STAMP(6)
        // (the next tile's S2 writes Ts only after its top barrier, which every wave reaches after its S3 reads; S3 of the next tile
        //  writes Cs two barriers after this S4's reads; the patch buffer read here is refilled one tile later, behind that barrier)
    }
    __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (0xF << 8));               // the look-ahead patch (all out-of-range past the last tile) and the stores
    if (p.clk && lane == 0)
        for (int i = 0; i < 7; ++i) p.clk[((size_t)blockIdx.x * CF_NW + wave) * 7 + i] = clk[i];
}

bool c2f_fused_valid(const C2fParams& p) {
    if (p.C != 32 || p.Cout != 64 || p.Kpad1 != 9 * 32 || p.Kpad2 != 9 * 32 || p.Kpad3 != 96) return false;
    if ((p.x_stride & 7) || (p.x_coff & 7) || (p.y_stride & 3) || (p.y_coff & 3)) return false;
    if (p.x_bytes >= (1ull << 31) || p.y_bytes >= (1ull << 31)) return false;
    if (p.w1_bytes < (size_t)32 * p.Kpad1 * 2 || p.w2_bytes < (size_t)32 * p.Kpad2 * 2 || p.w3_bytes < (size_t)64 * p.Kpad3 * 2) return false;
    const long covered = (long)((p.H + CF_TH - 1) / CF_TH * CF_TH) * ((p.W + 15) / 16 * 16);
    if (covered * 2 > (long)p.H * p.W * 3) return false;                 // (tiny maps: the separate kernels waste less)
    return true;
}

hipError_t launch_c2f_fused(const C2fParams& p, hipStream_t st) {
    const int tiles_h = (p.H + CF_TH - 1) / CF_TH, tiles_w = (p.W + 15) / 16;
    const int num_tiles = p.B * tiles_h * tiles_w;
    int G = 256;
    if (G > num_tiles) G = num_tiles;
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)c2f_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
        if (e != hipSuccess) return e;
        attr = true;
    }
    static const bool clocks = [] { const char* v = std::getenv("YOLOP_C2F_CLOCKS"); return v && *v == '1'; }();   // debug: per-stage s_memtime sums
    if (clocks) {
        C2fParams q = p;
        const size_t n = (size_t)G * CF_NW * 7;
        if (hipMalloc((void**)&q.clk, n * 8) != hipSuccess) return hipErrorOutOfMemory;
        hipLaunchKernelGGL(c2f_fused_kernel, dim3(G), dim3(CF_NW * 64), (size_t)CF_LDS, st, q, tiles_h, tiles_w, G);
        (void)hipStreamSynchronize(st);
        std::vector<unsigned long long> h(n);
        (void)hipMemcpy(h.data(), q.clk, n * 8, hipMemcpyDeviceToHost);
        (void)hipFree(q.clk);
        static const char* nm[7] = {"wait-dma", "barrier1", "S2", "barrier2", "S3", "barrier3", "S4"};
        const double tiles_per = (double)num_tiles / G;
        for (int w = 0; w < CF_NW; w += 7) {
            fprintf(stderr, "[c2f clocks] wave %d, s_memtime ticks per tile:", w);
            for (int i = 0; i < 7; ++i) {
                double s = 0;
                for (int g = 0; g < G; ++g) s += (double)h[((size_t)g * CF_NW + w) * 7 + i];
                fprintf(stderr, " %s %.0f", nm[i], s / G / tiles_per);
            }
            fprintf(stderr, "\n");
        }
        return hipGetLastError();
    }
    hipLaunchKernelGGL(c2f_fused_kernel, dim3(G), dim3(CF_NW * 64), (size_t)CF_LDS, st, p, tiles_h, tiles_w, G);
    return hipGetLastError();
}

}  // namespace yp
