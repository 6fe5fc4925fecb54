// "K split" form of the bf16 1x1 / 3x3 convolution for SMALL pixel counts (configuration ids 900+): the reference's real call shape
// is ONE frame per predict (yolo_seg/app.py:85-91), where the 20x20-level layers are 240-400 pixels x 256-4608 k x 64-512 output channels.
// The tiled kernels give such a layer 2-8 workgroups that walk K serially (32 us for the 3x3 512 -> 64 of the P5 box branch at
// one 640x640 frame, 10 us for every 1x1); here the parallelism comes from the other two axes:
//   workgroup = PXF x 16 pixels x FN x 16 output channels (grid = pixel tiles x channel tiles), its four waves SPLIT K:
//   a unit = one tap x 32 input channels = one `v_mfma_f32_16x16x32_bf16` per (pixel fragment, channel fragment); a lane loads 16 bytes
//   of its pixel (channels 8g..8g+7 at the tap) and 16 bytes of each weight row it owns straight from global memory into registers
//   (L2-resident at these sizes), UB units in flight; no LDS and no barrier in the K loop. The four partial tiles meet in LDS once,
//   then bias, SiLU, residual and the store.
// Same structure as conv_small.hip (the U^2-Net path), with the channel-tile grid axis, stride 2 and the engine's epilogue flags.
#include "common.h"

namespace yp {

typedef __attribute__((ext_vector_type(8))) __bf16 ks_bf16x8;
typedef __attribute__((ext_vector_type(4))) float ks_f32x4;

__device__ __forceinline__ float ks_silu(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

template <int FN, int PXF, int UB, bool HAS_RES, bool OUT_F32>
__global__ __launch_bounds__(256) void conv_ks_kernel(const ConvParams p) {
    __shared__ float4 part[4][FN * PXF][64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int HoWo = p.Ho * p.Wo;
    const int m0 = blockIdx.x * (16 * PXF), n0 = blockIdx.y * (16 * FN);
    int hi0[PXF], wi0[PXF], pbase[PXF];
#pragma unroll
    for (int b = 0; b < PXF; ++b) {
        const int m = m0 + b * 16 + fr;
        if (m < p.M) {
            const int bi = m / HoWo, r = m - bi * HoWo;
            const int ho = r / p.Wo, wo = r - ho * p.Wo;
            hi0[b] = ho * p.stride - p.pad;
            wi0[b] = wo * p.stride - p.pad;
            pbase[b] = bi * p.H * p.W;
        } else {
            hi0[b] = -(1 << 20); wi0[b] = 0; pbase[b] = 0;
        }
    }
    const int nch = p.Cin >> 5, units = p.ks * p.ks * nch;
    const __bf16* xb = (const __bf16*)p.x + p.x_coff + 8 * g;
    const __bf16* wb = (const __bf16*)p.w + (size_t)(n0 + fr) * p.Kpad + 8 * g;

    ks_f32x4 acc[FN][PXF];
#pragma unroll
    for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int b = 0; b < PXF; ++b) acc[a][b] = ks_f32x4{0.f, 0.f, 0.f, 0.f};

    for (int u0 = wave; u0 < units; u0 += 4 * UB) {
        uint4 xv[UB][PXF], wv[UB][FN];
#pragma unroll
        for (int i = 0; i < UB; ++i) {
            const int u = u0 + 4 * i;                          // (wave-uniform)
#pragma unroll
            for (int b = 0; b < PXF; ++b) xv[i][b] = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
            for (int a = 0; a < FN; ++a) wv[i][a] = make_uint4(0u, 0u, 0u, 0u);
            if (u < units) {
                const int tap = u / nch, cc = u - tap * nch;
                const int ky = tap / p.ks, kx = tap - ky * p.ks;
#pragma unroll
                for (int b = 0; b < PXF; ++b) {
                    const int hi = hi0[b] + ky, wi = wi0[b] + kx;
                    if ((unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W)
                        xv[i][b] = *(const uint4*)(xb + (size_t)(pbase[b] + hi * p.W + wi) * p.x_stride + cc * 32);
                }
#pragma unroll
                for (int a = 0; a < FN; ++a) wv[i][a] = *(const uint4*)(wb + (size_t)(a * 16) * p.Kpad + tap * p.Cin + cc * 32);
            }
        }
#pragma unroll
        for (int i = 0; i < UB; ++i)
#pragma unroll
            for (int a = 0; a < FN; ++a)
#pragma unroll
                for (int b = 0; b < PXF; ++b) {
                    union { uint4 u; ks_bf16x8 v; } w, x;
                    w.u = wv[i][a]; x.u = xv[i][b];
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.v, x.v, acc[a][b], 0, 0, 0);
                }
    }

#pragma unroll
    for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int b = 0; b < PXF; ++b) part[wave][a * PXF + b][lane] = make_float4(acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]);
    __syncthreads();

    // a wave finishes every fourth fragment: lane = (pixel fr of pixel fragment b, output channels n0 + 16a + 4g .. +3)
    for (int f = wave; f < FN * PXF; f += 4) {
        const int a = f / PXF, b = f - a * PXF;
        const int m = m0 + b * 16 + fr;
        const int co = n0 + a * 16 + 4 * g;
        if (m >= p.M || co >= p.Cout) continue;
        float4 s = part[0][f][lane];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float4 t = part[w][f][lane];
            s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
        float v[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j] += (co + j < p.Cout) ? p.bias[co + j] : 0.f;
            if (p.act == ACT_SILU) v[j] = ks_silu(v[j]);
            else if (p.act == ACT_RELU) v[j] = fmaxf(v[j], 0.f);
        }
        if (HAS_RES) {
            const uint2 r = *(const uint2*)((const __bf16*)p.res + (size_t)m * p.res_stride + p.res_coff + co);
            v[0] += __uint_as_float(r.x << 16); v[1] += __uint_as_float(r.x & 0xffff0000u);
            v[2] += __uint_as_float(r.y << 16); v[3] += __uint_as_float(r.y & 0xffff0000u);
        }
        if (OUT_F32) {
            *(float4*)((float*)p.y + (size_t)m * p.y_stride + p.y_coff + co) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
            __attribute__((aligned(8))) __bf16 o[4] = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
            *(uint2*)((__bf16*)p.y + (size_t)m * p.y_stride + p.y_coff + co) = *(const uint2*)o;
        }
    }
}

struct KsCfg { int FN, PXF; const char* name; };
static const KsCfg kKs[] = {
    {4, 1, "conv_ks_kernel<4,1,4>"},      // 16 px x 64 couts
    {4, 2, "conv_ks_kernel<4,2,3>"},      // 32 px x 64 couts
    {8, 1, "conv_ks_kernel<8,1,2>"},      // 16 px x 128 couts
    {2, 2, "conv_ks_kernel<2,2,4>"},      // 32 px x 32 couts
};
constexpr int kNumKs = (int)(sizeof(kKs) / sizeof(kKs[0]));

int conv_ks_num_cfgs() { return kNumKs; }

bool conv_ks_cfg_valid(const ConvParams& p, int c) {
    if (c < 0 || c >= kNumKs) return false;
    const KsCfg& k = kKs[c];
    if ((p.ks != 1 && p.ks != 3) || p.up != 1 || p.w2 || p.x2_C > 0 || p.pool_in) return false;
    if ((p.Cin % 32) != 0 || p.pad != p.ks / 2 || (p.dil > 1)) return false;
    if (p.M > 16384) return false;                                                            // a small-problem kernel: one frame, or the deepest level of a few
    if ((p.x_stride & 7) || (p.x_coff & 7) || (p.Kpad & 7)) return false;                     // 16-byte fragment loads
    if ((p.Cout & 3) || (p.y_stride & 3) || (p.y_coff & 3) || (p.res && ((p.res_stride & 3) || (p.res_coff & 3)))) return false;
    if (p.res && p.out_f32) return false;
    const int BN = k.FN * 16;
    const int ntiles = (p.Cout + BN - 1) / BN;
    if ((p.Cout + 127) / 128 * 128 < ntiles * BN) return false;                               // (the packed matrix has rows up to the next multiple of 128)
    if (BN >= 2 * ((p.Cout + 31) / 32 * 32) && BN > 32) return false;                         // mostly padding
    return true;
}

const char* conv_ks_kernel_name(int c) { return kKs[c].name; }

template <int FN, int PXF, int UB>
static hipError_t launch_ks_one(const ConvParams& p, hipStream_t st) {
    const dim3 grid((unsigned)((p.M + 16 * PXF - 1) / (16 * PXF)), (unsigned)((p.Cout + 16 * FN - 1) / (16 * FN))), blk(256);
    if (p.out_f32) hipLaunchKernelGGL((conv_ks_kernel<FN, PXF, UB, false, true>), grid, blk, 0, st, p);
    else if (p.res) hipLaunchKernelGGL((conv_ks_kernel<FN, PXF, UB, true, false>), grid, blk, 0, st, p);
    else hipLaunchKernelGGL((conv_ks_kernel<FN, PXF, UB, false, false>), grid, blk, 0, st, p);
    return hipGetLastError();
}

hipError_t launch_conv_ks(const ConvParams& p, int c, hipStream_t st) {
    if (!conv_ks_cfg_valid(p, c)) return hipErrorInvalidValue;
    switch (c) {
        case 0: return launch_ks_one<4, 1, 4>(p, st);
        case 1: return launch_ks_one<4, 2, 3>(p, st);
        case 2: return launch_ks_one<8, 1, 2>(p, st);
        default: return launch_ks_one<2, 2, 4>(p, st);
    }
}

}  // namespace yp
