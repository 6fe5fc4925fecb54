// Dense convolution (k in {1,3}, stride in {1,2}) as an implicit GEMM on the CDNA4 matrix cores, NHWC.
//
//   D[cout][pixel] = sum_k W[cout][k] * X[pixel][k],   k = (ky,kx,ci)
//
// The weights are the MFMA A operand and the activations the B operand, so a lane of the 16x16 accumulator
// holds 4 CONSECUTIVE output channels of ONE pixel (C/D map: col=lane&15 -> pixel, row=(lane>>4)*4+r -> cout):
// the NHWC store is an 8-byte (bf16) / 16-byte (f32) vector per lane with no LDS round trip.
// Both operands are "row x 8 contiguous k" fragments (16 B for bf16), read from LDS tiles of 32-deep k with a
// per-row chunk swizzle that makes the ds_read_b128 conflict-free (see swz()).
//
// Replaces every `Conv`/1x1 of the ultralytics graph run by `.predict` (reference yolo_seg/app.py:91);
// block definitions: SURVEY.md Appendix A.2 [U]. Fused epilogue: +bias, SiLU, +residual (after the
// activation: Bottleneck / PSA adds), optional fp32 output (head logits), optional strided output placement
// (ConvTranspose2d k2 s2 of Proto = 4 such GEMMs).
#include "common.h"

namespace yp {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <typename T> struct TT;
template <> struct TT<__bf16> {
    static constexpr int ES = 2;
    struct Frag { bf16x8 v; };
    __device__ static inline void mma(f32x4& acc, const Frag& a, const Frag& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, acc, 0, 0, 0);
    }
};
template <> struct TT<float> {
    static constexpr int ES = 4;
    struct __attribute__((aligned(16))) Frag { float v[8]; };
    // lane-group g=(lane>>4) owns k = 8g..8g+7; MFMA j sums element j of all four groups: any k bijection is
    // valid as long as A and B use the same one.
    // Blocked summation: the 32 products of a k-tile are summed on their own and the block sum is added to the running total, as a
    // vectorised CPU kernel (the reference's oneDNN path) keeps partial sums per lane - one K-long sequential fp32 chain carried about
    // twice that path's rounding noise through the network (measured against the fp64 oracle).
    __device__ static inline void mma(f32x4& acc, const Frag& a, const Frag& b) {
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 8; ++j) t = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[j], b.v[j], t, 0, 0, 0);
        acc += t;
    }
};

// Physical 8-element chunk of logical chunk `c` (0..3) in LDS row `row`. For 64-B rows (bf16) and the
// ds_read_b128 lane groups of gfx950 this visits 16 distinct 16-B slots per group (conflict-free);
// derivation in DESIGN.md "conv_igemm LDS image".
__device__ __forceinline__ int swz(int row, int c) { return c ^ ((4 - ((row >> 2) & 3)) & 3); }

// (fp32 parity mode and the register-staged bf16 fallback: the library expf and an IEEE division - the LDS-DMA kernels of the bf16 hot path
// have their own packed form, common.h silu4_packed)
__device__ __forceinline__ float silu(float x) { return x / (1.0f + expf(-x)); }

template <typename T, int BM, int BN, int WGM, int WGN>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvParams p) {
    constexpr int ES = TT<T>::ES;
    constexpr int RB = 32 * ES;  // bytes per LDS row (32 k)
    constexpr int CB = 8 * ES;   // bytes per chunk (8 k)
    constexpr int FM = BM / WGM / 16, FN = BN / WGN / 16;
    constexpr int XL = BM * 4 / 256;
    constexpr int WL = (BN * 4 + 255) / 256;
    static_assert(WGM * WGN == 4, "4 waves");
    static_assert(XL >= 1, "BM >= 64");
    using Frag = typename TT<T>::Frag;

    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (BM + BN) * RB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WGM, wn = wave / WGM;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int chunk = tid & 3;
    const int HoWo = p.Ho * p.Wo;
    const int dil = p.dil > 0 ? p.dil : 1;

    // ---- per-thread X rows (fixed over the k loop) ---------------------------------------------------
    int hi0[XL], wi0[XL], pbase[XL];
#pragma unroll
    for (int i = 0; i < XL; ++i) {
        const int m = m0 + (tid >> 2) + 64 * i;
        if (m < p.M) {
            const int b = m / HoWo, r = m - b * HoWo;
            const int ho = r / p.Wo, wo = r - ho * p.Wo;
            hi0[i] = ho * p.stride - p.pad;
            wi0[i] = wo * p.stride - p.pad;
            pbase[i] = b * p.H * p.W;
        } else {
            hi0[i] = -(1 << 16);  // never in bounds
            wi0[i] = 0;
            pbase[i] = 0;
        }
    }
    // k state of this thread's chunk: (ky,kx,c)
    int ky, kx, kc;
    {
        const int kg = chunk * 8;
        const int tap = kg / p.Cin;
        kc = kg - tap * p.Cin;
        ky = tap / p.ks;
        kx = tap - ky * p.ks;
    }
    const char* xb = (const char*)p.x;
    const char* wb = (const char*)p.w + ((size_t)(n0 + (tid >> 2)) * p.Kpad + chunk * 8) * ES;

    uint4 xr[XL][ES / 2], wr[WL][ES / 2];

    auto load_global = [&](int kt) {
#pragma unroll
        for (int i = 0; i < XL; ++i) {
            const int hi = hi0[i] + ky * dil, wi = wi0[i] + kx * dil;
            const bool ok = (ky < p.ks) && ((unsigned)hi < (unsigned)p.H) && ((unsigned)wi < (unsigned)p.W);
            const size_t off = ((size_t)(pbase[i] + hi * p.W + wi) * p.x_stride + p.x_coff + kc) * ES;
#pragma unroll
            for (int q = 0; q < ES / 2; ++q)
                xr[i][q] = ok ? *(const uint4*)(xb + off + 16 * q) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < WL; ++i) {
            if (BN * 4 >= 256 || tid < BN * 4) {
                const char* a = wb + ((size_t)(64 * i) * p.Kpad + (size_t)kt * 32) * ES;
#pragma unroll
                for (int q = 0; q < ES / 2; ++q) wr[i][q] = *(const uint4*)(a + 16 * q);
            }
        }
        // advance (ky,kx,kc) by 32 k
        kc += 32;
        while (kc >= p.Cin) {
            kc -= p.Cin;
            if (++kx == p.ks) { kx = 0; ++ky; }
        }
    };
    auto store_lds = [&](int buf) {
        unsigned char* Xs = smem + buf * (BM + BN) * RB;
        unsigned char* Ws = Xs + BM * RB;
#pragma unroll
        for (int i = 0; i < XL; ++i) {
            const int row = (tid >> 2) + 64 * i;
            unsigned char* d = Xs + row * RB + swz(row, chunk) * CB;
#pragma unroll
            for (int q = 0; q < ES / 2; ++q) *(uint4*)(d + 16 * q) = xr[i][q];
        }
#pragma unroll
        for (int i = 0; i < WL; ++i) {
            if (BN * 4 >= 256 || tid < BN * 4) {
                const int row = (tid >> 2) + 64 * i;
                unsigned char* d = Ws + row * RB + swz(row, chunk) * CB;
#pragma unroll
                for (int q = 0; q < ES / 2; ++q) *(uint4*)(d + 16 * q) = wr[i][q];
            }
        }
    };

    f32x4 acc[FN][FM];
#pragma unroll
    for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int b = 0; b < FM; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.Kpad / 32;
    load_global(0);
    store_lds(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) load_global(kt + 1);
        {
            const unsigned char* Xs = smem + (kt & 1) * (BM + BN) * RB;
            const unsigned char* Ws = Xs + BM * RB;
            Frag wf[FN], xf[FM];
            const int fr = lane & 15, fc = lane >> 4;
#pragma unroll
            for (int a = 0; a < FN; ++a) {
                const int row = wn * (BN / WGN) + a * 16 + fr;
                wf[a] = *(const Frag*)(Ws + row * RB + swz(row, fc) * CB);
            }
#pragma unroll
            for (int b = 0; b < FM; ++b) {
                const int row = wm * (BM / WGM) + b * 16 + fr;
                xf[b] = *(const Frag*)(Xs + row * RB + swz(row, fc) * CB);
            }
#pragma unroll
            for (int a = 0; a < FN; ++a)
#pragma unroll
                for (int b = 0; b < FM; ++b) TT<T>::mma(acc[a][b], wf[a], xf[b]);
        }
        if (more) store_lds((kt + 1) & 1);
        __syncthreads();
    }

    // ---- epilogue: bias, SiLU, residual, store 4 consecutive couts of one pixel --------------------
    const bool vec_ok = ((p.Cout & 3) == 0) && ((p.y_stride & 3) == 0) && ((p.y_coff & 3) == 0) &&
                        (p.res == nullptr || (((p.res_stride & 3) == 0) && ((p.res_coff & 3) == 0)));
#pragma unroll
    for (int b = 0; b < FM; ++b) {
        const int m = m0 + wm * (BM / WGM) + b * 16 + (lane & 15);
        if (m >= p.M) continue;
        size_t opix = (size_t)m;
        if (p.up != 1) {
            const int bb = m / HoWo, r = m - bb * HoWo;
            const int ho = r / p.Wo, wo = r - ho * p.Wo;
            opix = ((size_t)bb * (p.Ho * p.up) + ho * p.up + p.oy) * (size_t)(p.Wo * p.up) + wo * p.up + p.ox;
        }
#pragma unroll
        for (int a = 0; a < FN; ++a) {
            const int co = n0 + wn * (BN / WGN) + a * 16 + (lane >> 4) * 4;
            if (co >= p.Cout) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = co + r;
                float t = acc[a][b][r] + ((c < p.Cout) ? p.bias[c] : 0.f);
                if (p.act == ACT_SILU) t = silu(t);
                else if (p.act == ACT_RELU) t = fmaxf(t, 0.f);
                v[r] = t;
            }
            if (p.res) {
                const T* rp = (const T*)p.res + (size_t)m * p.res_stride + p.res_coff + co;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (co + r < p.Cout) v[r] += (float)rp[r];
            }
            if (p.out_f32) {
                float* yp = (float*)p.y + opix * p.y_stride + p.y_coff + co;
                if (vec_ok) *(float4*)yp = make_float4(v[0], v[1], v[2], v[3]);
                else
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (co + r < p.Cout) yp[r] = v[r];
            } else {
                T* yp = (T*)p.y + opix * p.y_stride + p.y_coff + co;
                if (vec_ok) {
                    __attribute__((aligned(16))) T o[4] = {(T)v[0], (T)v[1], (T)v[2], (T)v[3]};
                    if (ES == 2) *(uint2*)yp = *(const uint2*)o;
                    else *(uint4*)yp = *(const uint4*)o;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (co + r < p.Cout) yp[r] = (T)v[r];
                }
            }
        }
    }
}

// which instantiation launch_conv picks (also reported to the profiler/bench through yp_op_kernel)
static int conv_tile_choice(const ConvParams& p) {
    if (p.Cout <= 16) return 3;          // U^2-Net's 16-channel (and 1-channel side) layers: a 32-wide tile wastes half its MFMAs
    if (p.Cout <= 32) return 0;
    if ((p.Cout % 128) != 0 || p.M < 128 * 256) return 1;
    return 2;
}
// halo configuration selected for p (explicit p.cfg in [100,200) / [200,300), or the test override), or -1.
// returns 100+c (conv_halo) or 200+c (conv_halo_p)
static int halo_choice(const ConvParams& p, int dtype) {
    if (dtype != DT_BF16) return -1;
    if (p.x2_C > 0) {        // folded upsample: only the persistent LDS-DMA families implement the two-source gather
        auto ok = [&](int c) { return c >= 1200 ? conv_wrs_cfg_valid(p, c - 1200) : c >= 1100 ? conv_wres_cfg_valid(p, c - 1100) : c >= 900 ? false : c >= 800 ? conv_pxd_cfg_valid(p, c - 800) : (c >= 400 && c < 500 ? conv_dma_lc_cfg_valid(p, c - 400) : (c >= 300 && c < 400 && conv_dma_p_cfg_valid(p, c - 300))); };
        const int f = conv_dma_forced_cfg();
        if (ok(f)) return f;
        if (ok(p.cfg)) return p.cfg;
        for (int c = 0; c < conv_dma_p_num_cfgs(); ++c)
            if (conv_dma_p_cfg_valid(p, c)) return 300 + c;
        return -2;
    }
    auto valid = [&](int c) {
        if (c >= 1200) return conv_wrs_cfg_valid(p, c - 1200);
        if (c >= 1100) return conv_wres_cfg_valid(p, c - 1100);
        if (c >= 1000) return false;                                 // (1000 = pwsp_kernel: the engine launches it itself)
        if (c >= 900) return conv_ks_cfg_valid(p, c - 900);
        if (c >= 800) return conv_pxd_cfg_valid(p, c - 800);
        if (c >= 700) return conv_wreg_cfg_valid(p, c - 700);
        if (c >= 600) return conv_tile1_cfg_valid(p, c - 600);
        if (c >= 500) return conv_halo_s2_cfg_valid(p, c - 500);
        if (c >= 400) return conv_dma_lc_cfg_valid(p, c - 400);
        if (c >= 300) return conv_dma_p_cfg_valid(p, c - 300);
        if (c >= 200) return conv_halo_p_cfg_valid(p, c - 200);
        if (c >= 100) return conv_halo_cfg_valid(p, c - 100);
        return false;
    };
    const int f = conv_dma_forced_cfg();
    if (f >= 100 && valid(f)) return f;
    if (f >= 0 && f < 100) return -1;
    if (p.cfg >= 100 && valid(p.cfg)) return p.cfg;
    return -1;
}

// Is tile configuration id `cfg` (the autotuner's numbering: < 100 conv_dma, 100+ conv_halo, 200+ conv_halo_p, 300+ conv_dma_p, 400+ conv_dma_lc,
// 500+ conv_halo_s2, 600+ conv_tile1, 700+ conv_wreg, 800+ conv_pxd, 900+ conv_ks, 1100+ conv_wres, 1200+ conv_wrs; -1 = heuristic) one this build can launch for p? Used for
// configurations that come from outside the tuner (tune cache files, yp_tuning_import).
bool conv_cfg_usable(const ConvParams& p, int dtype, int cfg) {
    if (cfg == -1) return true;
    if (dtype != DT_BF16 || cfg < -1) return false;
    if (cfg >= 1200) return cfg - 1200 < conv_wrs_num_cfgs() && conv_wrs_cfg_valid(p, cfg - 1200);
    if (cfg >= 1100) return cfg - 1100 < conv_wres_num_cfgs() && conv_wres_cfg_valid(p, cfg - 1100);
    if (cfg >= 1000) return false;
    if (cfg >= 900) return cfg - 900 < conv_ks_num_cfgs() && p.x2_C == 0 && conv_ks_cfg_valid(p, cfg - 900);
    if (cfg >= 800) return cfg - 800 < conv_pxd_num_cfgs() && conv_pxd_cfg_valid(p, cfg - 800);
    if (cfg >= 700) return cfg - 700 < conv_wreg_num_cfgs() && p.x2_C == 0 && conv_wreg_cfg_valid(p, cfg - 700);
    if (cfg >= 600) return cfg - 600 < conv_tile1_num_cfgs() && p.x2_C == 0 && conv_tile1_cfg_valid(p, cfg - 600);
    if (cfg >= 500) return cfg - 500 < conv_halo_s2_num_cfgs() && p.x2_C == 0 && conv_halo_s2_cfg_valid(p, cfg - 500);
    if (cfg >= 400) return cfg - 400 < conv_dma_lc_num_cfgs() && conv_dma_lc_cfg_valid(p, cfg - 400);
    if (cfg >= 300) return cfg - 300 < conv_dma_p_num_cfgs() && conv_dma_p_cfg_valid(p, cfg - 300);
    if (cfg >= 200) return cfg - 200 < conv_halo_p_num_cfgs() && p.x2_C == 0 && conv_halo_p_cfg_valid(p, cfg - 200);
    if (cfg >= 100) return cfg - 100 < conv_halo_num_cfgs() && p.x2_C == 0 && conv_halo_cfg_valid(p, cfg - 100);
    return cfg < conv_dma_num_cfgs() && p.x2_C == 0 && conv_dma_supported(p) && conv_dma_cfg_valid(p, cfg);
}

const char* conv_kernel_name(const ConvParams& p, int dtype) {
    const int h = halo_choice(p, dtype);
    if (h >= 1200) return conv_wrs_kernel_name(h - 1200);
    if (h >= 1100) return conv_wres_kernel_name(h - 1100);
    if (h >= 900) return conv_ks_kernel_name(h - 900);
    if (h >= 800) return conv_pxd_kernel_name(h - 800);
    if (h >= 700) return conv_wreg_kernel_name(h - 700);
    if (h >= 600) return conv_tile1_kernel_name(h - 600);
    if (h >= 500) return conv_halo_s2_kernel_name(h - 500);
    if (h >= 400) return conv_dma_lc_kernel_name(h - 400);
    if (h >= 300) return conv_dma_p_kernel_name(h - 300);
    if (h >= 200) return conv_halo_p_kernel_name(h - 200);
    if (h >= 100) return conv_halo_kernel_name(h - 100);
    if (dtype == DT_BF16 && conv_dma_supported(p)) return conv_dma_kernel_name(p);
    static const char* names[2][4] = {
        {"conv_igemm_kernel<bf16,128,32,4,1>", "conv_igemm_kernel<bf16,128,64,2,2>", "conv_igemm_kernel<bf16,128,128,2,2>", "conv_igemm_kernel<bf16,128,16,4,1>"},
        {"conv_igemm_kernel<f32,128,32,4,1>", "conv_igemm_kernel<f32,128,64,2,2>", "conv_igemm_kernel<f32,128,128,2,2>", "conv_igemm_kernel<f32,128,16,4,1>"}};
    return names[dtype == DT_BF16 ? 0 : 1][conv_tile_choice(p)];
}

template <typename T>
static hipError_t launch_conv_t(const ConvParams& p, hipStream_t st) {
    const int M = p.M;
    dim3 blk(256);
    const int choice = conv_tile_choice(p);
    if (choice == 3) {
        dim3 grid((M + 127) / 128, 1);
        hipLaunchKernelGGL((conv_igemm_kernel<T, 128, 16, 4, 1>), grid, blk, 0, st, p);
    } else if (choice == 0) {
        dim3 grid((M + 127) / 128, 1);
        hipLaunchKernelGGL((conv_igemm_kernel<T, 128, 32, 4, 1>), grid, blk, 0, st, p);
    } else if (choice == 1) {
        // 64-wide cout tiles: least padding for 64/80/192/320-class widths, and more CTAs for the 20x20 layers
        dim3 grid((M + 127) / 128, (p.Cout + 63) / 64);
        hipLaunchKernelGGL((conv_igemm_kernel<T, 128, 64, 2, 2>), grid, blk, 0, st, p);
    } else {
        dim3 grid((M + 127) / 128, (p.Cout + 127) / 128);
        hipLaunchKernelGGL((conv_igemm_kernel<T, 128, 128, 2, 2>), grid, blk, 0, st, p);
    }
    return hipGetLastError();
}

hipError_t launch_conv_igemm(const ConvParams& p, int dtype, hipStream_t st) {
    if (dtype == DT_BF16) return launch_conv_t<__bf16>(p, st);
    return launch_conv_t<float>(p, st);
}

hipError_t launch_conv(const ConvParams& p, int dtype, hipStream_t st) {
    const int h = halo_choice(p, dtype);
    if (p.x2_C > 0 && h < 300) return hipErrorInvalidValue;      // (the plan folds an upsample only when such a configuration exists)
    if (h >= 1200) return launch_conv_wrs(p, h - 1200, st);
    if (h >= 1100) return launch_conv_wres(p, h - 1100, st);
    if (h >= 900) return launch_conv_ks(p, h - 900, st);
    if (h >= 800) return launch_conv_pxd(p, h - 800, st);
    if (h >= 700) return launch_conv_wreg(p, h - 700, st);
    if (h >= 600) return launch_conv_tile1(p, h - 600, st);
    if (h >= 500) return launch_conv_halo_s2(p, h - 500, st);
    if (h >= 400) return launch_conv_dma_lc(p, h - 400, st);
    if (h >= 300) return launch_conv_dma_p(p, h - 300, st);
    if (h >= 200) return launch_conv_halo_p(p, h - 200, st);
    if (h >= 100) return launch_conv_halo(p, h - 100, st);
    if (dtype == DT_BF16 && conv_dma_supported(p)) return launch_conv_dma(p, st);
    if (dtype == DT_BF16) return launch_conv_t<__bf16>(p, st);
    return launch_conv_t<float>(p, st);
}

}  // namespace yp
