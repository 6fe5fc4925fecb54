// Last layer of a class branch + the class-max pass in one kernel: fp32 logits y[px][nc] = W . x[px] + b (the 1x1 `one2one_cv3.{l}.2` /
// `cv3.{l}.2`, no activation) AND the per-anchor key bits(sigmoid(max_c y[px][c])) that the top-k's first stage selects on (OP_AMAX).
// SURVEY.md A.4 / A.6 [U] (v10Detect's class logits; v10postprocess takes the max over classes first), run inside `.predict`
// (reference yolo_seg/app.py:91).
//
// Why: the logits of a level are 204 800 x 80 fp32 = 65.5 MB at P3 / batch 32; the class-max pass read them straight back (14 + 7 + 6 us per
// step over the three levels, three launches - 21 us of a 480-us forward at one frame per call). Here the maximum is taken from the MFMA
// accumulators before the logits leave the registers: a lane holds 4 channels x NF fragments of one pixel, the four lanes of a pixel meet
// through two cross-lane exchanges.
//
// The GEMM is tiny (K = 128 ... 256) and the kernel HBM-bound: persistent workgroups, the [TP px][K] pixel tile double-buffered in LDS
// (LDS-DMA, whole pixel rows per piece), the weights resident, 4 waves x (TP / 64) pixel fragments x all channel fragments.
#include "common.h"

namespace yp {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ float co_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }     // the form of head.hip's sigmoidf_: the same bits

constexpr int CO_NW = 4;
constexpr int CO_MAXNF = 8;             // up to 128 classes

template <int TP, int NF>
__global__ __launch_bounds__(CO_NW * 64) void cls_out_kernel(const ClsOutParams p, const int G, const int TPe) {
    constexpr int FM = TP / (16 * CO_NW);                          // pixel fragments per wave
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int RB = p.K * 2;                                        // bytes per row (a multiple of 128)
    const int CPR = RB >> 4;                                       // 16-byte chunks per row
    constexpr int nf = NF;                                         // channel fragments (NF * 16 >= nc; rows beyond nc are zero weights)
    unsigned char* const Ws = smem;                                // [nf * 16][RB]
    unsigned char* const Xs = smem + (size_t)nf * 16 * RB;         // 2 x [TP][RB]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fc = lane >> 4;
    const int ntiles = (p.M + TPe - 1) / TPe;                     // TPe <= TP rows of a tile in use: equal pixels per workgroup (conv_wres.hip)
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)p.w_bytes, 0x00020000);

    // rows are written 16 bytes per lane; position pc of row r holds source chunk pc ^ (r & 7) (K % 64 == 0: the chunks of a row come in
    // eights; a fragment read's 8 rows per cycle then fall into 8 different 16-byte bank groups)
    auto issue_rows = [&](const __amdgpu_buffer_rsrc_t rs, unsigned char* dst, int nrows, long row0, int stride_el, int coff, long row_lim) {
        const int pieces = (nrows * CPR + 63) >> 6;
        for (int ii = wave; ii < pieces; ii += CO_NW) {
            const int s = ii * 64 + lane;
            const int r = s / CPR, pc = s - r * CPR;
            const int c = pc ^ (r & 7);
            const bool ok = r < nrows && row0 + r < row_lim;
            const unsigned voff = ok ? (unsigned)(((row0 + r) * stride_el + coff + c * 8) * 2) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(dst + ii * 1024), 16, voff, 0, 0, 0);
        }
    };
    issue_rows(wrs, Ws, nf * 16, 0, p.Kpad, 0, (long)(p.nc + 127) / 128 * 128);      // (rows beyond nc are the packed matrix's zero padding)
    int tile = blockIdx.x;
    if (tile < ntiles) issue_rows(xrs, Xs, TPe, (long)tile * TPe, p.x_stride, p.x_coff, p.M);

    float bias[NF][4];
#pragma unroll
    for (int a = 0; a < NF; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int co = a * 16 + fc * 4 + r; bias[a][r] = (co < p.nc) ? p.bias[co] : 0.f; }
    // (known complete before the loop and passed through an empty asm statement: otherwise the compiler, which cannot count a loop
    // iteration's vector-memory operations, waits `vmcnt(0)` in front of the accumulators' initialisation in every iteration - behind the
    // next tile's loads, i.e. no prefetch; see conv_wres.hip)
    __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (0xF << 8));
#pragma unroll
    for (int a = 0; a < NF; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(bias[a][r]));

    for (int it = 0; tile < ntiles; tile += G, ++it) {
        __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (0xF << 8));     // vmcnt(0): this tile's rows (and the weights) have landed; the previous tile's stores are out
        __builtin_amdgcn_s_barrier();
        if (tile + G < ntiles) issue_rows(xrs, Xs + ((it & 1) ^ 1) * (size_t)TP * RB, TPe, (long)(tile + G) * TPe, p.x_stride, p.x_coff, p.M);
        const unsigned char* X = Xs + (it & 1) * (size_t)TP * RB;
        f32x4 acc[NF][FM];
#pragma unroll
        for (int a = 0; a < NF; ++a)
#pragma unroll
            for (int f = 0; f < FM; ++f) acc[a][f] = f32x4{bias[a][0], bias[a][1], bias[a][2], bias[a][3]};
        const int nks = p.K >> 5;
#pragma unroll 2
        for (int ks = 0; ks < nks; ++ks) {
            const int ch = ks * 4 + fc;
            const int pc = ch ^ (fr & 7);                                          // (row & 7 = fr & 7 for both operands' fragments)
            bf16x8 xf[FM], wf[NF];                                                 // (all reads of the substep, then its MFMAs: no branch in between)
#pragma unroll
            for (int f = 0; f < FM; ++f) xf[f] = *(const bf16x8*)(X + (size_t)((wave * FM + f) * 16 + fr) * RB + pc * 16);
#pragma unroll
            for (int a = 0; a < NF; ++a) wf[a] = *(const bf16x8*)(Ws + (size_t)(a * 16 + fr) * RB + pc * 16);
#pragma unroll
            for (int a = 0; a < NF; ++a)
#pragma unroll
                for (int f = 0; f < FM; ++f) acc[a][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[a], xf[f], acc[a][f], 0, 0, 0);
        }
        // ---- logits out, class maximum -> key ---------------------------------------------------------------------------------
#pragma unroll
        for (int f = 0; f < FM; ++f) {
            const int ri = (wave * FM + f) * 16 + fr;
            const long m = (long)tile * TPe + ri;
            const bool ok = ri < TPe && m < p.M;
            float mx = -INFINITY;
#pragma unroll
            for (int a = 0; a < NF; ++a) {
                const int co = a * 16 + fc * 4;
                if (co < p.nc) {                                                   // (nc % 4 == 0: a lane's four channels exist together)
                    mx = fmaxf(mx, fmaxf(fmaxf(acc[a][f][0], acc[a][f][1]), fmaxf(acc[a][f][2], acc[a][f][3])));
                    if (ok) *(float4*)(p.y + m * p.y_stride + p.y_coff + co) = make_float4(acc[a][f][0], acc[a][f][1], acc[a][f][2], acc[a][f][3]);
                }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            if (ok && fc == 0) p.keys[m] = __float_as_uint(co_sigmoid(mx));
        }
    }
    __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (0xF << 8));
}

static int cls_out_nf(const ClsOutParams& p) { const int nf = (p.nc + 15) / 16; return nf <= 1 ? 1 : nf <= 5 ? 5 : CO_MAXNF; }   // instantiated fragment counts
static size_t cls_out_lds(const ClsOutParams& p, int TP) { return (size_t)cls_out_nf(p) * 16 * p.K * 2 + (size_t)2 * TP * p.K * 2; }
static int cls_out_tile(const ClsOutParams& p) { return cls_out_lds(p, 128) <= 78 * 1024 ? 128 : 64; }      // (small enough for two workgroups per CU, else the 64-pixel tile)

bool cls_out_valid(const ClsOutParams& p) {
    if (p.K < 64 || (p.K % 64) != 0 || p.Kpad != p.K || p.K > 512) return false;
    if (p.nc < 4 || (p.nc & 3) || p.nc > 16 * CO_MAXNF) return false;
    if ((p.x_stride & 7) || (p.x_coff & 7) || (p.y_stride & 3) || (p.y_coff & 3)) return false;
    if (p.x_bytes >= (1ull << 31) || p.w_bytes >= (1ull << 31) || p.M <= 0) return false;
    return cls_out_lds(p, cls_out_tile(p)) <= 150 * 1024;
}

const char* cls_out_kernel_name(const ClsOutParams& p) {
    static const char* nm[2][3] = {{"cls_out_kernel<64,1>", "cls_out_kernel<64,5>", "cls_out_kernel<64,8>"}, {"cls_out_kernel<128,1>", "cls_out_kernel<128,5>", "cls_out_kernel<128,8>"}};
    const int nf = cls_out_nf(p);
    return nm[cls_out_tile(p) == 128][nf == 1 ? 0 : nf == 5 ? 1 : 2];
}

template <int TP, int NF>
static hipError_t launch_cls_out_t(const ClsOutParams& p, hipStream_t st) {
    const size_t sh = cls_out_lds(p, TP);
    auto kern = cls_out_kernel<TP, NF>;
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr = true;
    }
    const int ntiles = (p.M + TP - 1) / TP;
    // (an HBM-bound kernel: as many workgroups as fit - two per CU when the tiles are small - each walking every G-th tile)
    int G = sh <= 78 * 1024 ? 512 : 256;
    if (G > ntiles) G = ntiles;
    const int rounds = (int)((p.M + (long)G * TP - 1) / ((long)G * TP));
    const int TPe = tile_balance_enabled(4) ? (int)((p.M + (long)G * rounds - 1) / ((long)G * rounds)) : TP;       // equal pixels per workgroup
    hipLaunchKernelGGL(kern, dim3((unsigned)G), dim3(CO_NW * 64), sh, st, p, G, TPe);
    return hipGetLastError();
}

hipError_t launch_cls_out(const ClsOutParams& p, hipStream_t st) {
    if (!cls_out_valid(p)) return hipErrorInvalidValue;
    const int nf = cls_out_nf(p);
    if (cls_out_tile(p) == 128) return nf == 1 ? launch_cls_out_t<128, 1>(p, st) : nf == 5 ? launch_cls_out_t<128, 5>(p, st) : launch_cls_out_t<128, 8>(p, st);
    return nf == 1 ? launch_cls_out_t<64, 1>(p, st) : nf == 5 ? launch_cls_out_t<64, 5>(p, st) : launch_cls_out_t<64, 8>(p, st);
}

}  // namespace yp
