// 3x3 (dilated) convolution for SMALL problems (fp32 and bf16) - the deep levels of U^2-Net's RSU blocks (16 / 32 / 64 / 128 -> 16 / 64
// channels on 12^2 .. 95^2 maps; reference yolo_seg/tasks/models/U2Net.py:11-26 REBNCONV inside RSU7 .. RSU4F, called per frame by
// yolo_seg/app.py:184 through tasks/unet_segment.py:53-73).
//
// The general implicit GEMM (conv_igemm.hip) walks K in 32-wide steps with one workgroup barrier and one dependent global -> LDS
// round trip per step: on a 12x12 map that is two workgroups running 5 .. 36 serial round trips, 10 .. 74 us per layer for
// 1 .. 40 MFLOP, whatever the map size. Here a workgroup owns 16 output pixels and ALL output channels and its four waves split K:
// a unit of work is one tap x 16 input channels; a lane loads one float4 of its pixel (channels 4g .. 4g+3) and one float4 per 16
// output channels of the weight row it owns, straight from global memory into registers (everything is L2-resident at these sizes),
// several units in flight, and feeds them to `v_mfma_f32_16x16x4_f32` with the k order permuted identically on both operands
// (MFMA j of a unit sums k = j, 4+j, 8+j, 12+j; in bf16 a unit is a single `v_mfma_f32_16x16x16_bf16`). No LDS, no barrier in the K loop; the four partial tiles are summed through LDS
// once, followed by bias, activation, residual and vector stores. The `POOL` form takes the 2x2 ceil-mode max pool in front of the
// convolution while loading (max of up to four source pixels per operand: exact), which removes the pool launch and its tensor. In fp32: fp32 products and sums throughout (the engine's parity mode).
// U^2-Net-P, 380x380 crop: 110 of its 119 convolutions are faster here than in conv_igemm (8 .. 17 us instead of 10 .. 74 us at 12^2 .. 48^2,
// still 26 vs 37 us for 64 -> 16 at 190^2); the engine times both per layer in a plan's first pass. A no-K-split sibling for the
// full-resolution layers (64 pixels per wave, weights reused over four pixel fragments, still no LDS) lost to conv_igemm everywhere
// (145 vs 110 us for 32 -> 64 at 380^2): operands straight from L2 do not feed 64 MFMAs per unit fast enough; it was removed.
#include "common.h"

namespace yp {

typedef __attribute__((ext_vector_type(4))) float cs_f32x4;
typedef __attribute__((ext_vector_type(4))) __bf16 cs_bf16x4;

__device__ __forceinline__ float cs_act(float v, int act) {
    if (act == ACT_RELU) return fmaxf(v, 0.f);
    if (act == ACT_SILU) return v / (1.0f + __expf(-v));
    return v;
}

// element type: fp32 -> one float4 per lane and unit, four `v_mfma_f32_16x16x4_f32`; bf16 -> 8 bytes per lane and unit, one
// `v_mfma_f32_16x16x16_bf16` (a lane's four k values are its four consecutive channels in both forms)
template <typename T> struct CsT;
template <> struct CsT<float> {
    typedef float4 Vec;
    static __device__ __forceinline__ Vec zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
    static __device__ __forceinline__ Vec vmax(const Vec& a, const Vec& b) { return make_float4(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w)); }
    // bilinear tap, the expression of u2_up_kernel (u2net.hip)
    static __device__ __forceinline__ Vec lerp(const Vec& a, const Vec& b, const Vec& c, const Vec& d, float ly0, float ly1, float lx0, float lx1) {
        return make_float4(ly0 * (lx0 * a.x + lx1 * b.x) + ly1 * (lx0 * c.x + lx1 * d.x), ly0 * (lx0 * a.y + lx1 * b.y) + ly1 * (lx0 * c.y + lx1 * d.y),
                           ly0 * (lx0 * a.z + lx1 * b.z) + ly1 * (lx0 * c.z + lx1 * d.z), ly0 * (lx0 * a.w + lx1 * b.w) + ly1 * (lx0 * c.w + lx1 * d.w));
    }
    static __device__ __forceinline__ void mma(cs_f32x4& acc, const Vec& w, const Vec& x) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, x.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, x.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, x.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, x.w, acc, 0, 0, 0);
    }
};
template <> struct CsT<__bf16> {
    typedef uint2 Vec;
    static __device__ __forceinline__ Vec zero() { return make_uint2(0u, 0u); }
    static __device__ __forceinline__ unsigned max2(unsigned a, unsigned b) {           // two packed bf16: exact (a bf16 is the top half of an fp32)
        const float lo = fmaxf(__uint_as_float(a << 16), __uint_as_float(b << 16)), hi = fmaxf(__uint_as_float(a & 0xffff0000u), __uint_as_float(b & 0xffff0000u));
        return (__float_as_uint(lo) >> 16) | (__float_as_uint(hi) & 0xffff0000u);
    }
    static __device__ __forceinline__ Vec vmax(const Vec& a, const Vec& b) { return make_uint2(max2(a.x, b.x), max2(a.y, b.y)); }
    static __device__ __forceinline__ unsigned lerp2(unsigned a, unsigned b, unsigned c, unsigned d, float ly0, float ly1, float lx0, float lx1) {
        const float lo = ly0 * (lx0 * __uint_as_float(a << 16) + lx1 * __uint_as_float(b << 16)) + ly1 * (lx0 * __uint_as_float(c << 16) + lx1 * __uint_as_float(d << 16));
        const float hi = ly0 * (lx0 * __uint_as_float(a & 0xffff0000u) + lx1 * __uint_as_float(b & 0xffff0000u)) +
                         ly1 * (lx0 * __uint_as_float(c & 0xffff0000u) + lx1 * __uint_as_float(d & 0xffff0000u));
        __attribute__((aligned(4))) __bf16 o[2] = {(__bf16)lo, (__bf16)hi};                 // (rounded as the stand-alone kernel stores it)
        return *(const unsigned*)o;
    }
    static __device__ __forceinline__ Vec lerp(const Vec& a, const Vec& b, const Vec& c, const Vec& d, float ly0, float ly1, float lx0, float lx1) {
        return make_uint2(lerp2(a.x, b.x, c.x, d.x, ly0, ly1, lx0, lx1), lerp2(a.y, b.y, c.y, d.y, ly0, ly1, lx0, lx1));
    }
    static __device__ __forceinline__ void mma(cs_f32x4& acc, const Vec& w, const Vec& x) {
        union { uint2 u; cs_bf16x4 v; } a, b;
        a.u = w; b.u = x;
        acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a.v, b.v, acc, 0, 0, 0);
    }
};

// F.upsample(size=..., mode='bilinear') = upsample_bilinear2d, align_corners=False: src = scale*(dst+0.5)-0.5 clamped at 0 (as u2net.hip u2_bil)
__device__ __forceinline__ void cs_bil(int dst, int n_in, int n_out, int& i0, int& i1, float& l0, float& l1) {
    const float scale = (float)n_in / (float)n_out;
    float f = scale * ((float)dst + 0.5f) - 0.5f;
    f = fmaxf(f, 0.f);
    i0 = (int)f;
    i1 = i0 + ((i0 < n_in - 1) ? 1 : 0);
    l1 = f - (float)i0;
    l0 = 1.f - l1;
}

// MODE 0: plain; 1: the 2x2 ceil-mode max pool in front of the convolution taken while loading; 2: input channels [0, x2_C) are the
// bilinear resize of the low-resolution tensor x2 to the convolution's input size, taken while loading (the up-sample launch and its
// half of the concat buffer disappear); channels >= x2_C come from x as usual
template <typename T, int FN, int UB, int MODE>
__global__ __launch_bounds__(256) void conv_small_kernel(const ConvParams p) {
    constexpr bool POOL = MODE == 1, UP = MODE == 2;
    typedef typename CsT<T>::Vec Vec;
    __shared__ float4 part[4][FN][64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int HoWo = p.Ho * p.Wo;
    const int m = blockIdx.x * 16 + fr;
    const bool valid = m < p.M;
    int hi0 = 0, wi0 = 0, pbase = 0;
    if (valid) {
        const int b = m / HoWo, r = m - b * HoWo;
        const int ho = r / p.Wo, wo = r - ho * p.Wo;
        hi0 = ho * p.stride - p.pad;
        wi0 = wo * p.stride - p.pad;
        pbase = POOL ? b * p.src_H * p.src_W : b * p.H * p.W;
    }
    const int dil = p.dil > 0 ? p.dil : 1;
    const int nch = p.Cin >> 4, units = p.ks * p.ks * nch;
    const T* xb = (const T*)p.x + p.x_coff + 4 * g;
    const T* wb = (const T*)p.w + (size_t)fr * p.Kpad + 4 * g;
    const T* x2b = UP ? (const T*)p.x2 + p.x2_coff + 4 * g : nullptr;
    const int up_units = UP ? p.x2_C >> 4 : 0, bimg = valid ? m / HoWo : 0;

    cs_f32x4 acc[FN];
#pragma unroll
    for (int a = 0; a < FN; ++a) acc[a] = cs_f32x4{0.f, 0.f, 0.f, 0.f};

    for (int u0 = wave; u0 < units; u0 += 4 * UB) {
        Vec xv[UB], wv[UB][FN];
#pragma unroll
        for (int i = 0; i < UB; ++i) {
            const int u = u0 + 4 * i;                          // (wave-uniform)
            xv[i] = CsT<T>::zero();
#pragma unroll
            for (int a = 0; a < FN; ++a) wv[i][a] = CsT<T>::zero();
            if (u < units) {
                const int tap = u / nch, cc = u - tap * nch;
                const int ky = tap / p.ks, kx = tap - ky * p.ks;
                const int hi = hi0 + ky * dil, wi = wi0 + kx * dil;
                if (valid && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W) {
                    if (POOL) {                                // 2x2 / stride 2 / ceil mode: the window is clipped at the bottom / right edge
                        const int sh = 2 * hi, sw = 2 * wi;
                        const T* q = xb + (size_t)(pbase + sh * p.src_W + sw) * p.x_stride + cc * 16;
                        const bool right = sw + 1 < p.src_W, down = sh + 1 < p.src_H;
                        Vec v = *(const Vec*)q;
                        if (right) v = CsT<T>::vmax(v, *(const Vec*)(q + p.x_stride));
                        if (down) {
                            const T* q2 = q + (size_t)p.src_W * p.x_stride;
                            v = CsT<T>::vmax(v, *(const Vec*)q2);
                            if (right) v = CsT<T>::vmax(v, *(const Vec*)(q2 + p.x_stride));
                        }
                        xv[i] = v;
                    } else if (UP && cc < up_units) {
                        int ya, yb, xa, xc;
                        float ly0, ly1, lx0, lx1;
                        cs_bil(hi, p.x2_H, p.H, ya, yb, ly0, ly1);
                        cs_bil(wi, p.x2_W, p.W, xa, xc, lx0, lx1);
                        const T* q0 = x2b + (size_t)((bimg * p.x2_H + ya) * p.x2_W) * p.x2_stride + cc * 16;
                        const T* q1 = x2b + (size_t)((bimg * p.x2_H + yb) * p.x2_W) * p.x2_stride + cc * 16;
                        xv[i] = CsT<T>::lerp(*(const Vec*)(q0 + (size_t)xa * p.x2_stride), *(const Vec*)(q0 + (size_t)xc * p.x2_stride),
                                             *(const Vec*)(q1 + (size_t)xa * p.x2_stride), *(const Vec*)(q1 + (size_t)xc * p.x2_stride), ly0, ly1, lx0, lx1);
                    } else {
                        xv[i] = *(const Vec*)(xb + (size_t)(pbase + hi * p.W + wi) * p.x_stride + cc * 16);
                    }
                }
#pragma unroll
                for (int a = 0; a < FN; ++a) wv[i][a] = *(const Vec*)(wb + (size_t)(a * 16) * p.Kpad + tap * p.Cin + cc * 16);
            }
        }
#pragma unroll
        for (int i = 0; i < UB; ++i)
#pragma unroll
            for (int a = 0; a < FN; ++a) CsT<T>::mma(acc[a], wv[i][a], xv[i]);
    }

#pragma unroll
    for (int a = 0; a < FN; ++a) part[wave][a][lane] = make_float4(acc[a][0], acc[a][1], acc[a][2], acc[a][3]);
    __syncthreads();

    // wave a finishes fragment a: lane = (pixel fr, couts 16a + 4g .. +3)
    if (wave < FN && valid) {
        const int a = wave;
        float4 s = part[0][a][lane];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float4 t = part[w][a][lane];
            s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
        const int co = a * 16 + 4 * g;
        if (co < p.Cout) {
            float v[4] = {s.x, s.y, s.z, s.w};
            const bool vec = co + 4 <= p.Cout && ((p.y_stride | p.y_coff) & 3) == 0;
            const T* rp = p.res ? (const T*)p.res + (size_t)m * p.res_stride + p.res_coff + co : nullptr;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (co + j < p.Cout) {
                    v[j] = cs_act(v[j] + p.bias[co + j], p.act);
                    if (rp) v[j] += (float)rp[j];
                }
            }
            if (sizeof(T) == 4 || p.out_f32) {
                float* yo = (float*)p.y + (size_t)m * p.y_stride + p.y_coff + co;
                if (vec) *(float4*)yo = make_float4(v[0], v[1], v[2], v[3]);
                else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (co + j < p.Cout) yo[j] = v[j];
                }
            } else {
                __bf16* yo = (__bf16*)p.y + (size_t)m * p.y_stride + p.y_coff + co;
                if (vec) {
                    __attribute__((aligned(8))) __bf16 o[4] = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                    *(uint2*)yo = *(const uint2*)o;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (co + j < p.Cout) yo[j] = (__bf16)v[j];
                }
            }
        }
    }
}

bool conv_small_valid(const ConvParams& p, int dtype) {
    if (p.up != 1) return false;
    if (p.x2_C > 0 && (!p.up_bilinear || (p.x2_C & 15) || (p.x2_stride & 3) || (p.x2_coff & 3) || p.pool_in)) return false;
    if ((p.Cin & 15) || p.Cout > 64 || p.ks != 3) return false;
    if ((p.x_stride & 3) || (p.x_coff & 3) || (p.Kpad & 3)) return false;
    if (dtype == DT_F32 && p.out_f32) return false;
    if (p.pool_in && ((p.src_H + 1) / 2 != p.H || (p.src_W + 1) / 2 != p.W)) return false;
    return true;
}

template <typename T>
static hipError_t launch_conv_small_t(const ConvParams& p, hipStream_t st) {
    const dim3 grid((unsigned)((p.M + 15) / 16)), blk(256);
    if (p.pool_in) {
        if (p.Cout <= 16) hipLaunchKernelGGL((conv_small_kernel<T, 1, 4, 1>), grid, blk, 0, st, p);
        else if (p.Cout <= 32) hipLaunchKernelGGL((conv_small_kernel<T, 2, 4, 1>), grid, blk, 0, st, p);
        else hipLaunchKernelGGL((conv_small_kernel<T, 4, 3, 1>), grid, blk, 0, st, p);
    } else if (p.x2_C > 0) {
        if (p.Cout <= 16) hipLaunchKernelGGL((conv_small_kernel<T, 1, 4, 2>), grid, blk, 0, st, p);
        else if (p.Cout <= 32) hipLaunchKernelGGL((conv_small_kernel<T, 2, 4, 2>), grid, blk, 0, st, p);
        else hipLaunchKernelGGL((conv_small_kernel<T, 4, 3, 2>), grid, blk, 0, st, p);
    } else {
        if (p.Cout <= 16) hipLaunchKernelGGL((conv_small_kernel<T, 1, 4, 0>), grid, blk, 0, st, p);
        else if (p.Cout <= 32) hipLaunchKernelGGL((conv_small_kernel<T, 2, 4, 0>), grid, blk, 0, st, p);
        else hipLaunchKernelGGL((conv_small_kernel<T, 4, 3, 0>), grid, blk, 0, st, p);
    }
    return hipGetLastError();
}

hipError_t launch_conv_small(const ConvParams& p, int dtype, hipStream_t st) {
    if (!conv_small_valid(p, dtype)) return hipErrorInvalidValue;
    return dtype == DT_BF16 ? launch_conv_small_t<__bf16>(p, st) : launch_conv_small_t<float>(p, st);
}

}  // namespace yp
