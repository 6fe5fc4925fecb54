// Depthwise k x k (stride 1) on the matrix cores, for small feature maps (the 7x7 RepVGGDW of CIB at 20x20, SURVEY.md A.2 [U];
// run inside `.predict`, reference yolo_seg/app.py:91).
//
// A depthwise tap is an element-wise scale, which the VALU form pays for with a bf16 unpack + FMA per element and tap
// (49 taps -> VALU-bound at ~35 us per layer). Here a tap is one MFMA with a DIAGONAL weight fragment:
//     out[ch][px] += Wdiag_tap[ch][k] * x[k][px + tap],   Wdiag_tap[ch][k] = (k == ch) ? w[tap][ch] : 0
// Products with the zeros are exact zeros and the accumulation is fp32, so the numbers are those of the VALU form. Only
// 1/16 of the MACs are useful, but the matrix pipe is >16x the VALU rate for bf16 and needs no unpacking.
//
// One workgroup = one image x 32 channels: the whole map plus its zero halo sits in LDS (LDS-DMA, out-of-range offset =
// zero fill), pixels are taken in 4x4 patches (a lane reads its own pixel's 16-B channel slice, so any patch shape is a
// legal B operand; 4x4 tiles a 20x20 map exactly, 16x1 would waste 37 %). The LDS row pitch is = 4 (mod 8) pixels so that the
// two patch rows read by one 8-lane group fall on different halves of the chunk swizzle: no bank conflicts for any tap.
// A wave owns up to MAXP patches and walks ky outermost, building the 2*KS diagonal fragments of that tap row once
// (LDS dword & per-lane mask, see conv_dwpw.hip) and reusing them for all its patches.
#include "common.h"

namespace yp {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ float silu_m(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ int mswz(int row) { return ((row >> 2) & 1) << 1; }

constexpr int DWM_MAXP = 7;      // patches per wave (4 waves -> maps up to 28 patches = 448 pixels)

static inline int dwm_pitch(int W, int KS) {
    int P = W + KS - 1;
    while ((P & 7) != 4) ++P;
    return P;
}

template <int KS, bool SPLIT>
__global__ __launch_bounds__(256) void dwconv_mfma_kernel(const DwParams p, const int P, const int x_instr) {
    constexpr unsigned OOB = 0x80000000u;
    constexpr int W_INSTR = (KS * KS + 15) / 16;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char* const Xs = smem;                                   // [HP*P px][32 ch] bf16, chunk-swizzled
    unsigned char* const Wd = smem + (size_t)x_instr * 1024;         // [KS*KS][32] bf16

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fc = lane >> 4;
    const int nchunk = p.C >> 5;
    const int c = blockIdx.x % nchunk, b = blockIdx.x / nchunk;
    const int psplit = (int)gridDim.y, ps = (int)blockIdx.y;      // the map's patches are dealt out over gridDim.y workgroups (one frame: 16 would idle 240 CUs)
    const int pad = KS / 2, HP = p.H + KS - 1;

    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)(KS * KS * p.C * 2), 0x00020000);

    for (int ii = wave; ii < x_instr; ii += 4) {
        const int s = ii * 64 + lane;
        const int hp = s >> 2, pc = s & 3;
        const int c8 = pc ^ mswz(hp);
        const int hy = hp / P, hx = hp - hy * P;
        const int hi = hy - pad, wi = hx - pad;
        const bool ok = (hy < HP) && ((unsigned)hi < (unsigned)p.H) && ((unsigned)wi < (unsigned)p.W);
        const unsigned voff = ok ? (unsigned)((((b * p.H + hi) * p.W + wi) * p.x_stride + p.x_coff + c * 32 + c8 * 8) * 2) : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_void*)(Xs + ii * 1024), 16, voff, 0, 0, 0);
    }
    for (int ii = wave; ii < W_INSTR; ii += 4) {
        const int s = ii * 64 + lane;
        const int tap = s >> 2, c8 = s & 3;
        const unsigned voff = (tap < KS * KS) ? (unsigned)((tap * p.C + c * 32 + c8 * 8) * 2) : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_void*)(Wd + ii * 1024), 16, voff, 0, 0, 0);
    }

    // per-lane constants of the diagonal fragments (same construction as conv_dwpw.hip, round 3: TWO taps per MFMA - k < 16 carries 16
    // channels of tap A, k >= 16 the same 16 channels of tap B; lane group fc reads its 16 bytes from the halo pixel of tap (fc >> 1))
    unsigned dmask[4];
    {
        const bool valid = (fc & 1) == (fr >> 3);
        const unsigned hw = (fr & 1) ? 0xffff0000u : 0x0000ffffu;
#pragma unroll
        for (int q = 0; q < 4; ++q) dmask[q] = (valid && q == ((fr & 7) >> 1)) ? hw : 0u;
    }
    const int tapsel = fc >> 1;
    const int PW = p.W >> 2, NP = PW * (p.H >> 2);
    const int nslot = ((NP + psplit - 1) / psplit + 3) >> 2;      // patch slots a wave of this workgroup can have
    f32x4 acc[DWM_MAXP][2];
    {
        const float4 b0 = *(const float4*)(p.bias + c * 32 + fc * 4);
        const float4 b1 = *(const float4*)(p.bias + c * 32 + 16 + fc * 4);
#pragma unroll
        for (int i = 0; i < DWM_MAXP; ++i) {
            acc[i][0] = f32x4{b0.x, b0.y, b0.z, b0.w};
            acc[i][1] = f32x4{b1.x, b1.y, b1.z, b1.w};
        }
    }
    int pbase[DWM_MAXP];                                // halo pixel of this lane's patch pixel at tap (0,0)
#pragma unroll
    for (int i = 0; i < DWM_MAXP; ++i) {
        const int pj = (wave + 4 * i) * psplit + ps;
        const int pi = (pj < NP) ? pj : ((wave * psplit + ps < NP) ? wave * psplit + ps : 0);     // a wave's surplus slots redo its first patch (discarded):
        const int pr = pi / PW, pcn = pi - pr * PW;                    // the tap loop stays branch-free and the scheduler can
        pbase[i] = (pr * 4 + (fr >> 2)) * P + pcn * 4 + (fr & 3);      // keep the LDS reads of the next patch in flight
    }

    __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (0xF << 8));           // vmcnt(0)
    __builtin_amdgcn_s_barrier();

    // the KS*KS taps in pairs (2j, 2j+1) (the last one alone: its B half gets zero weights): per pair two fragments (channel halves), per
    // patch two fragment reads + two MFMAs - 25 x 2 MFMAs per patch instead of 49 x 2
    constexpr int NT = KS * KS, NPR = (NT + 1) / 2;
    for (int j = 0; j < NPR; ++j) {
        const int tA = 2 * j, tB = (2 * j + 1 < NT) ? 2 * j + 1 : tA;
        const int t = tapsel ? tB : tA;
        const int ky = t / KS, kx = t - ky * KS;
        const int toff = ky * P + kx;                               // this lane's tap offset in halo pixels
        bf16x8 wd[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            unsigned wbits = *(const unsigned*)(Wd + t * 64 + ((fr + 16 * h) & ~1) * 2);
            if (tapsel && 2 * j + 1 >= NT) wbits = 0u;
            const uint4 v = make_uint4(wbits & dmask[0], wbits & dmask[1], wbits & dmask[2], wbits & dmask[3]);
            wd[h] = *(const bf16x8*)&v;
        }
#pragma unroll
        for (int i = 0; i < DWM_MAXP; ++i) {
            if (SPLIT && i >= nslot) break;                     // (workgroup-uniform: slots beyond this workgroup's share of the patches)
            const int hp = pbase[i] + toff;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const bf16x8 xf = *(const bf16x8*)(Xs + swz64((unsigned)(hp * 64 + (2 * h + (fc & 1)) * 16)));
                acc[i][h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wd[h], xf, acc[i][h], 0, 0, 0);
            }
        }
    }

    __bf16* yb = (__bf16*)p.y + (size_t)b * p.H * p.W * p.y_stride + p.y_coff + c * 32 + fc * 4;
#pragma unroll
    for (int i = 0; i < DWM_MAXP; ++i) {
        const int pi = (wave + 4 * i) * psplit + ps;
        if (pi < NP) {
            const int pr = pi / PW, pcn = pi - pr * PW;
            const int yy = pr * 4 + (fr >> 2), xx = pcn * 4 + (fr & 3);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                __attribute__((aligned(8))) __bf16 o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (__bf16)(p.act == ACT_SILU ? silu_m(acc[i][h][j]) : acc[i][h][j]);
                *(uint2*)(yb + (size_t)(yy * p.W + xx) * p.y_stride + 16 * h) = *(const uint2*)o;
            }
        }
    }
}

static size_t dwm_lds(const DwParams& p, int* x_instr) {
    const int P = dwm_pitch(p.W, p.ks), HP = p.H + p.ks - 1;
    const int xi = (HP * P + 15) / 16;
    if (x_instr) *x_instr = xi;
    return (size_t)xi * 1024 + (size_t)((p.ks * p.ks + 15) / 16) * 1024;
}

bool dwconv_mfma_valid(const DwParams& p, int dtype) {
    if (dtype != DT_BF16 || p.stride != 1 || p.ks != 7 || p.gs != 0 || p.res != nullptr) return false;
    if ((p.C & 31) || (p.H & 3) || (p.W & 3) || p.Ho != p.H || p.Wo != p.W) return false;
    if ((p.x_stride & 7) || (p.x_coff & 7) || (p.y_stride & 3) || (p.y_coff & 3)) return false;
    if (p.x_bytes >= (1ull << 31)) return false;
    if ((p.H >> 2) * (p.W >> 2) > 4 * DWM_MAXP) return false;
    return dwm_lds(p, nullptr) <= 72 * 1024;            // two workgroups per CU
}

hipError_t launch_dwconv_mfma(const DwParams& p, hipStream_t st) {
    int x_instr = 0;
    const size_t sh = dwm_lds(p, &x_instr);
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)dwconv_mfma_kernel<7, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)dwconv_mfma_kernel<7, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
        if (e != hipSuccess) return e;
        attr = true;
    }
    // few (image, channel block) pairs: split each map's patches over up to 4 workgroups (every one stages the whole map - it is L2-resident)
    const int wgs = p.B * (p.C / 32);
    const int psplit = wgs >= 256 ? 1 : wgs >= 128 ? 2 : 4;
    if (psplit == 1) hipLaunchKernelGGL((dwconv_mfma_kernel<7, false>), dim3((unsigned)wgs, 1), dim3(256), sh, st, p, dwm_pitch(p.W, p.ks), x_instr);
    else hipLaunchKernelGGL((dwconv_mfma_kernel<7, true>), dim3((unsigned)wgs, (unsigned)psplit), dim3(256), sh, st, p, dwm_pitch(p.W, p.ks), x_instr);
    return hipGetLastError();
}

}  // namespace yp
