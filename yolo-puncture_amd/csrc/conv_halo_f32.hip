// 3x3 (dilated) fp32 convolution for LARGE maps - the full-resolution layers of U^2-Net's RSU blocks in the engine's parity mode
// (3 / 64 / 128 -> 16 / 64 channels at 190^2 .. 380^2; reference yolo_seg/tasks/models/U2Net.py:11-26 REBNCONV, called per frame by
// yolo_seg/app.py:184 through tasks/unet_segment.py:53-73).
//
// conv_igemm.hip gathers an im2col tile through registers into LDS every 32 k: 31 .. 49 % of the fp32 MFMA rate on these layers.
// Here a workgroup owns an 8 x 32 tile of output pixels and ALL output channels and walks the input channels in chunks of 16:
//   DMA  the (8 + 2d) x (32 + 2d) halo patch of the chunk (64-B pixel rows, zero outside the frame = the padding) and the chunk's
//        weights [tap][co][16] straight into LDS (`buffer_load ... lds`, 1 KB per instruction, 16-B pieces XOR-swizzled by (row >> 2) & 3 so
//        that every ds_read_b128 of 16 consecutive rows is conflict-free), two buffers: chunk c + 1 lands while chunk c is multiplied
//   MMA  a wave owns two tile rows = four 16-pixel fragments and all FN output-channel fragments: per tap FN + 4 ds_read_b128 feed
//        16 x FN `v_mfma_f32_16x16x4_f32` (a lane's float4 = k 4g .. 4g+3 of its pixel / output channel, k order permuted identically
//        on both operands as in conv_small.hip)
// Same arithmetic as conv_igemm's fp32 path (fp32 products, fp32 sums), another summation order.
#include "common.h"
#include <cstdlib>

namespace yp {

typedef __attribute__((ext_vector_type(4))) float hf_f32x4;
typedef __attribute__((address_space(3))) void hf_lds_void;

constexpr int HF_TH = 8, HF_TW = 32;

// LDS reads behind the compiler's back: it cannot tell them from the in-flight LDS-DMA of the next chunk apart and would drain vmcnt to 0
// in front of the first one (= no prefetch at all). Ordered by explicit lgkmcnt waits below.
__device__ __forceinline__ float4 hf_read16(const unsigned char* src) {
    float4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"((unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)src) : "memory");
    return v;
}

__device__ __forceinline__ float hf_act(float v, int act) {
    if (act == ACT_RELU) return fmaxf(v, 0.f);
    if (act == ACT_SILU) return v / (1.0f + __expf(-v));
    return v;
}

// LDS per buffer: X patch (PP pixels rounded up to 16, 64 B each) then W (9 * FN * 16 rows of 64 B)
// Two LDS buffers (chunk c + 1 lands under the MFMAs of chunk c). With 64 output channels that is 116 KB = one workgroup per CU; a
// single-buffer form with two workgroups per CU measured 7 .. 11 % slower (128 vs 120 us for 32 -> 64 at 380^2) and was dropped.
template <int FN>
__global__ __launch_bounds__(256) void conv_halo_f32_kernel(const ConvParams p, const int tiles_w, const int tiles_h, const int xinstr) {
    constexpr unsigned OOB = 0x80000000u;
    constexpr int WROWS = 9 * FN * 16, WINSTR = WROWS / 16;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int d = p.dil > 0 ? p.dil : 1;
    const int PW = HF_TW + 2 * d;                                  // patch width (pixels)
    const int bufbytes = (xinstr + WINSTR) * 1024;
    int t = blockIdx.x;
    const int tx = t % tiles_w; t /= tiles_w;
    const int ty = t % tiles_h; const int b = t / tiles_h;
    const int y0 = ty * HF_TH, x0 = tx * HF_TW;

    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)p.w_bytes, 0x00020000);
    const int nchunks = p.Cin >> 4;

    auto issue = [&](int cc, int buf) {
        if (p.dbg == 2) return;                                    // (timing ablation: no operand traffic)
        unsigned char* Xs = smem + buf * bufbytes;
        unsigned char* Ws = Xs + xinstr * 1024;
        for (int ii = wave; ii < xinstr; ii += 4) {                // 16 patch pixels x 64 B
            const int hp = ii * 16 + (lane >> 2);
            const int c = (lane & 3) ^ ((hp >> 2) & 3);
            const int hy = hp / PW, hx = hp - hy * PW;
            const int hi = y0 - d + hy, wi = x0 - d + hx;
            const bool ok = hy < HF_TH + 2 * d && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            const unsigned voff = ok ? (unsigned)((((size_t)(b * p.H + hi) * p.W + wi) * p.x_stride + p.x_coff + cc * 16 + c * 4) * 4) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (hf_lds_void*)(Xs + ii * 1024), 16, voff, 0, 0, 0);
        }
        for (int ii = wave; ii < WINSTR; ii += 4) {                // 16 weight rows (tap, co) x 64 B
            const int row = ii * 16 + (lane >> 2);                 // = tap * (FN * 16) + co
            const int c = (lane & 3) ^ ((row >> 2) & 3);
            const int tap = row / (FN * 16), co = row - tap * (FN * 16);
            const unsigned voff = (unsigned)(((size_t)co * p.Kpad + tap * p.Cin + cc * 16 + c * 4) * 4);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (hf_lds_void*)(Ws + ii * 1024), 16, voff, 0, 0, 0);
        }
    };

    hf_f32x4 acc[FN][4];
#pragma unroll
    for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[a][q] = hf_f32x4{0.f, 0.f, 0.f, 0.f};

    // this wave's four pixel fragments: tile rows 2 * wave and 2 * wave + 1, columns 0..15 / 16..31; patch pixel of (row, col) at tap
    // (ky, kx) = (row + ky * d) * PW + col + kx * d
    int pbase[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) pbase[q] = (2 * wave + (q >> 1)) * PW + (q & 1) * 16 + fr;

    issue(0, 0);
    for (int cc = 0; cc < nchunks; ++cc) {
        const int buf = cc & 1;
        __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (0xF << 8));     // vmcnt(0): this wave's pieces of chunk cc have landed
        __syncthreads();                                           // ... everyone's have, and everyone is done with the other buffer
        if (cc + 1 < nchunks) issue(cc + 1, buf ^ 1);
        const unsigned char* Xs = smem + buf * bufbytes;
        const unsigned char* Ws = Xs + xinstr * 1024;
        // fragments of tap t + 1 are read under the MFMAs of tap t (two register sets)
        float4 wv[2][FN], xv[2][4];
        auto read_tap = [&](int tap, int set) {
            const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
            for (int a = 0; a < FN; ++a) {
                const int row = tap * (FN * 16) + a * 16 + fr;
                wv[set][a] = hf_read16(Ws + row * 64 + ((g ^ ((row >> 2) & 3)) * 16));
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int hp = pbase[q] + (ky * PW + kx) * d;
                xv[set][q] = hf_read16(Xs + hp * 64 + ((g ^ ((hp >> 2) & 3)) * 16));
            }
        };
        if (p.dbg == 3) continue;                                  // (timing ablation: no fragment reads, no MFMAs)
        read_tap(0, 0);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int set = tap & 1;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);                    // nothing (the MFMAs below in particular) is scheduled across the wait
            if (tap + 1 < 9) read_tap(tap + 1, set ^ 1);
            // k slice outermost: consecutive MFMAs go to different accumulators (4 * FN of them between two on the same one)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int a = 0; a < FN; ++a)
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[a][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[set][a][j], xv[set][q][j], acc[a][q], 0, 0, 0);
        }
    }

    // ---- epilogue ---------------------------------------------------------------------------------------------------------------
    // An accumulator lane holds 4 channels of ONE pixel: stored directly, a wave's instruction writes 16 pixels x 64 B at the tensor's
    // pixel pitch (half-used lines; four instructions later complete them) - measured 60 us of a 120-us launch for 32 -> 64 at 380^2
    // (37 MB out + 37 MB residual in). With FN >= 2 the tile goes through LDS once (bias + activation applied, 16-B pieces XOR-swizzled by
    // the pixel) and comes back with a pixel's channels across consecutive lanes: whole 128- / 256-B rows per pixel, for the residual
    // read and the store alike.
    const bool vec = ((p.y_stride | p.y_coff) & 3) == 0;
    if (FN >= 2 && vec && (p.Cout & 3) == 0 && (!p.res || ((p.res_stride | p.res_coff) & 3) == 0)) {
        constexpr int CH = FN * 4;                                 // 16-B pieces per pixel row
        constexpr int PPI = 64 / CH;                               // pixels per store instruction
        __syncthreads();                                           // every wave is done with the operand buffers
        unsigned char* const T = smem + wave * (64 * CH * 16);    // this wave's [64 px][CH pieces]
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int a = 0; a < FN; ++a) {
                const int co = a * 16 + 4 * g;
                float4 v;
                v.x = hf_act(acc[a][q][0] + (co + 0 < p.Cout ? p.bias[co + 0] : 0.f), p.act);
                v.y = hf_act(acc[a][q][1] + (co + 1 < p.Cout ? p.bias[co + 1] : 0.f), p.act);
                v.z = hf_act(acc[a][q][2] + (co + 2 < p.Cout ? p.bias[co + 2] : 0.f), p.act);
                v.w = hf_act(acc[a][q][3] + (co + 3 < p.Cout ? p.bias[co + 3] : 0.f), p.act);
                const int px = q * 16 + fr;
                *(float4*)(T + (px * CH + ((a * 4 + g) ^ (px & (CH - 1)))) * 16) = v;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // this wave's tile is written (it alone reads it)
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 64 / PPI; ++i) {
            const int px = i * PPI + lane / CH, c = lane % CH;     // pixel of the wave's 64, piece of its row
            const int q = px >> 4, f = px & 15;
            const int yy = y0 + 2 * wave + (q >> 1), xx = x0 + (q & 1) * 16 + f;
            if (yy >= p.Ho || xx >= p.Wo || 4 * c >= p.Cout) continue;
            float4 v = *(const float4*)(T + (px * CH + (c ^ (px & (CH - 1)))) * 16);
            const size_t m = ((size_t)b * p.Ho + yy) * p.Wo + xx;
            if (p.res) {
                const float4 r = *(const float4*)((const float*)p.res + m * p.res_stride + p.res_coff + 4 * c);
                v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
            }
            *(float4*)((float*)p.y + m * p.y_stride + p.y_coff + 4 * c) = v;
        }
        return;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int yy = y0 + 2 * wave + (q >> 1), xx = x0 + (q & 1) * 16 + fr;
        if (yy >= p.Ho || xx >= p.Wo) continue;
        const size_t m = ((size_t)b * p.Ho + yy) * p.Wo + xx;
#pragma unroll
        for (int a = 0; a < FN; ++a) {
            const int co = a * 16 + 4 * g;
            if (co >= p.Cout) continue;
            float v[4] = {acc[a][q][0], acc[a][q][1], acc[a][q][2], acc[a][q][3]};
            const float* rp = p.res ? (const float*)p.res + m * p.res_stride + p.res_coff + co : nullptr;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (co + j < p.Cout) {
                    v[j] = hf_act(v[j] + p.bias[co + j], p.act);
                    if (rp) v[j] += rp[j];
                }
            }
            float* yo = (float*)p.y + m * p.y_stride + p.y_coff + co;
            if (co + 4 <= p.Cout && vec) *(float4*)yo = make_float4(v[0], v[1], v[2], v[3]);
            else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (co + j < p.Cout) yo[j] = v[j];
            }
        }
    }
}

static int hf_xinstr(int d) { return ((HF_TH + 2 * d) * (HF_TW + 2 * d) + 15) / 16; }
static size_t hf_lds(int d, int fn) { return (size_t)2 * (hf_xinstr(d) + 9 * fn) * 1024; }

bool conv_halo_f32_valid(const ConvParams& p, int dtype) {
    if (dtype != DT_F32 || p.out_f32 || p.up != 1 || p.x2_C > 0 || p.pool_in) return false;
    if ((p.Cin & 15) || p.Cout > 64 || p.ks != 3 || p.stride != 1) return false;
    const int d = p.dil > 0 ? p.dil : 1;
    if (p.pad != d || p.Ho != p.H || p.Wo != p.W) return false;
    if ((p.x_stride & 3) || (p.x_coff & 3) || (p.Kpad & 3)) return false;
    if (p.x_bytes >= (1ull << 31) || p.w_bytes >= (1ull << 31)) return false;
    // the weight matrix must hold FN * 16 rows (it is packed with its rows rounded up to 128)
    const int fn = p.Cout <= 16 ? 1 : p.Cout <= 32 ? 2 : 4;
    return hf_lds(d, fn) <= 160 * 1024;
}

hipError_t launch_conv_halo_f32(const ConvParams& p_in, int dtype, hipStream_t st) {
    if (!conv_halo_f32_valid(p_in, dtype)) return hipErrorInvalidValue;
    ConvParams p = p_in;
    static const int dbg = [] { const char* s = getenv("YOLOP_HF_DBG"); return s ? atoi(s) : 0; }();      // 2 / 3: timing ablations (wrong results)
    if (dbg) p.dbg = dbg;
    const int d = p.dil > 0 ? p.dil : 1;
    const int fn = p.Cout <= 16 ? 1 : p.Cout <= 32 ? 2 : 4;
    const int tiles_w = (p.Wo + HF_TW - 1) / HF_TW, tiles_h = (p.Ho + HF_TH - 1) / HF_TH;
    const int B = p.M / (p.Ho * p.Wo);
    const size_t sh = hf_lds(d, fn);
    const dim3 grid((unsigned)(B * tiles_h * tiles_w)), blk(256);
    static bool attr[3] = {false, false, false};
    const int ai = fn == 1 ? 0 : fn == 2 ? 1 : 2;
    const void* fptr = fn == 1 ? (const void*)conv_halo_f32_kernel<1> : fn == 2 ? (const void*)conv_halo_f32_kernel<2> : (const void*)conv_halo_f32_kernel<4>;
    if (!attr[ai]) {
        hipError_t e = hipFuncSetAttribute(fptr, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr[ai] = true;
    }
    const int xi = hf_xinstr(d);
    if (fn == 1) hipLaunchKernelGGL(conv_halo_f32_kernel<1>, grid, blk, sh, st, p, tiles_w, tiles_h, xi);
    else if (fn == 2) hipLaunchKernelGGL(conv_halo_f32_kernel<2>, grid, blk, sh, st, p, tiles_w, tiles_h, xi);
    else hipLaunchKernelGGL(conv_halo_f32_kernel<4>, grid, blk, sh, st, p, tiles_w, tiles_h, xi);
    return hipGetLastError();
}

}  // namespace yp
