// HBM-bound helpers of the YOLOv10 graph: stem conv (u8 BGR -> first feature map), depthwise convs,
// SPPF 5x5 max-pool, nearest x2 upsample. All NHWC, 8 channels (one 16-B bf16 vector) per thread, fp32 math.
// Blocks: SURVEY.md Appendix A.2 [U] (`Conv` with g=c, `SCDown.cv2`, `CIB`, `RepVGGDW`, `SPPF.m`,
// `nn.Upsample`), executed inside `.predict` (reference yolo_seg/app.py:91).
#include "common.h"

namespace yp {

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }

template <typename T> struct Vec8;
template <> struct Vec8<__bf16> {
    uint4 raw;
    __device__ inline void load(const __bf16* p) { raw = *(const uint4*)p; }
    __device__ inline void store(__bf16* p) const { *(uint4*)p = raw; }
    __device__ inline void unpack(float* f) const {
        const uint32_t u[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i] = __uint_as_float(u[i] << 16);
            f[2 * i + 1] = __uint_as_float(u[i] & 0xffff0000u);
        }
    }
    __device__ inline void pack(const float* f) {
        __attribute__((aligned(16))) __bf16 o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (__bf16)f[i];
        raw = *(const uint4*)o;
    }
};
template <> struct Vec8<float> {
    float4 a, b;
    __device__ inline void load(const float* p) { a = *(const float4*)p; b = *(const float4*)(p + 4); }
    __device__ inline void store(float* p) const { *(float4*)p = a; *(float4*)(p + 4) = b; }
    __device__ inline void unpack(float* f) const {
        f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
    }
    __device__ inline void pack(const float* f) {
        a = make_float4(f[0], f[1], f[2], f[3]); b = make_float4(f[4], f[5], f[6], f[7]);
    }
};

template <typename T> __device__ __forceinline__ float round_to(float x);
template <> __device__ __forceinline__ float round_to<__bf16>(float x) { return (float)(__bf16)x; }
template <> __device__ __forceinline__ float round_to<float>(float x) { return x; }

// ---------------------------------------------------------------------------------------------------------
// stem: uint8 BGR NHWC -> Conv(3->C0, k3 s2 p1)+bias+SiLU. x/255 (IEEE division, as `im.float()/255`),
// BGR->RGB folded into the weight order. One thread = one output pixel x 8 output channels.
// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void stem_kernel(const StemParams p) {
    extern __shared__ float sw[];   // [27][C0] weights + [C0] bias
    const int nW = 27 * p.C0;
    for (int i = threadIdx.x; i < nW + p.C0; i += blockDim.x) sw[i] = (i < nW) ? p.w[i] : p.bias[i - nW];
    __syncthreads();
    const int groups = p.C0 >> 3;
    const long total = (long)p.B * p.Ho * p.Wo * groups;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int g = (int)(gid % groups);
    const long pix = gid / groups;
    const int wo = (int)(pix % p.Wo);
    const int ho = (int)((pix / p.Wo) % p.Ho);
    const int b = (int)(pix / ((long)p.Wo * p.Ho));
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = sw[nW + g * 8 + j];
    const uint8_t* xb = p.x + (size_t)b * p.H * p.W * 3;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int hi = ho * 2 - 1 + ky;
        if ((unsigned)hi >= (unsigned)p.H) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int wi = wo * 2 - 1 + kx;
            if ((unsigned)wi >= (unsigned)p.W) continue;
            const uint8_t* px = xb + ((size_t)hi * p.W + wi) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float v = round_to<T>((float)px[c] / 255.0f);
                const float* wv = sw + ((ky * 3 + kx) * 3 + c) * p.C0 + g * 8;
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = fmaf(v, wv[j], acc[j]);
            }
        }
    }
    if (p.act == ACT_SILU) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = silu_f(acc[j]);
    }
    Vec8<T> o;
    o.pack(acc);
    o.store((T*)p.y + (size_t)pix * p.y_stride + p.y_coff + g * 8);
}

hipError_t launch_stem(const StemParams& p, int dtype, hipStream_t st) {
    const long total = (long)p.B * p.Ho * p.Wo * (p.C0 / 8);
    const int blk = 256;
    const unsigned grid = (unsigned)((total + blk - 1) / blk);
    const size_t sh = (size_t)(28 * p.C0) * sizeof(float);
    if (dtype == DT_BF16) hipLaunchKernelGGL(stem_kernel<__bf16>, dim3(grid), dim3(blk), sh, st, p);
    else hipLaunchKernelGGL(stem_kernel<float>, dim3(grid), dim3(blk), sh, st, p);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// depthwise conv k in {3,7}, stride in {1,2}; optional input-channel gather (PSA `pe` reads the v rows of
// the interleaved qkv tensor), fused bias / SiLU / residual(after act).
// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void dwconv_kernel(const DwParams p) {
    const int groups = p.C >> 3;
    const long total = (long)p.B * p.Ho * p.Wo * groups;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int g = (int)(gid % groups);
    const long pix = gid / groups;
    const int wo = (int)(pix % p.Wo);
    const int ho = (int)((pix / p.Wo) % p.Ho);
    const int b = (int)(pix / ((long)p.Wo * p.Ho));
    const int c = g * 8;
    const int cin = p.gs ? (p.x_coff + (c / p.gs) * p.gstride + (c % p.gs)) : (p.x_coff + c);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = p.bias[c + j];
    const T* xb = (const T*)p.x + (size_t)b * p.H * p.W * p.x_stride + cin;
    const T* wb = (const T*)p.w + c;
    for (int ky = 0; ky < p.ks; ++ky) {
        const int hi = ho * p.stride - p.pad + ky;
        if ((unsigned)hi >= (unsigned)p.H) continue;
        for (int kx = 0; kx < p.ks; ++kx) {
            const int wi = wo * p.stride - p.pad + kx;
            if ((unsigned)wi >= (unsigned)p.W) continue;
            Vec8<T> xv, wv;
            xv.load(xb + ((size_t)hi * p.W + wi) * p.x_stride);
            wv.load(wb + (size_t)(ky * p.ks + kx) * p.C);
            float xf[8], wf[8];
            xv.unpack(xf);
            wv.unpack(wf);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = fmaf(xf[j], wf[j], acc[j]);
        }
    }
    if (p.act == ACT_SILU) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = silu_f(acc[j]);
    }
    if (p.res) {
        Vec8<T> rv;
        rv.load((const T*)p.res + (size_t)pix * p.res_stride + p.res_coff + c);
        float rf[8];
        rv.unpack(rf);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += rf[j];
    }
    Vec8<T> o;
    o.pack(acc);
    o.store((T*)p.y + (size_t)pix * p.y_stride + p.y_coff + c);
}

hipError_t launch_dwconv(const DwParams& p, int dtype, hipStream_t st) {
    const long total = (long)p.B * p.Ho * p.Wo * (p.C / 8);
    const int blk = 256;
    const unsigned grid = (unsigned)((total + blk - 1) / blk);
    if (dtype == DT_BF16) hipLaunchKernelGGL(dwconv_kernel<__bf16>, dim3(grid), dim3(blk), 0, st, p);
    else hipLaunchKernelGGL(dwconv_kernel<float>, dim3(grid), dim3(blk), 0, st, p);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// MaxPool2d(5, 1, 2) (implicit -inf padding), channel-slice in / channel-slice out of the SPPF concat buffer
// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void pool5_kernel(const PoolParams p) {
    const int groups = p.C >> 3;
    const long total = (long)p.B * p.H * p.W * groups;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int g = (int)(gid % groups);
    const long pix = gid / groups;
    const int x = (int)(pix % p.W);
    const int y = (int)((pix / p.W) % p.H);
    const int b = (int)(pix / ((long)p.W * p.H));
    float m[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) m[j] = -INFINITY;
    const T* xb = (const T*)p.x + (size_t)b * p.H * p.W * p.x_stride + p.x_coff + g * 8;
    for (int dy = -2; dy <= 2; ++dy) {
        const int yy = y + dy;
        if ((unsigned)yy >= (unsigned)p.H) continue;
        for (int dx = -2; dx <= 2; ++dx) {
            const int xx = x + dx;
            if ((unsigned)xx >= (unsigned)p.W) continue;
            Vec8<T> v;
            v.load(xb + ((size_t)yy * p.W + xx) * p.x_stride);
            float f[8];
            v.unpack(f);
#pragma unroll
            for (int j = 0; j < 8; ++j) m[j] = fmaxf(m[j], f[j]);
        }
    }
    Vec8<T> o;
    o.pack(m);
    o.store((T*)p.y + (size_t)pix * p.y_stride + p.y_coff + g * 8);
}

hipError_t launch_pool5(const PoolParams& p, int dtype, hipStream_t st) {
    const long total = (long)p.B * p.H * p.W * (p.C / 8);
    const int blk = 256;
    const unsigned grid = (unsigned)((total + blk - 1) / blk);
    if (dtype == DT_BF16) hipLaunchKernelGGL(pool5_kernel<__bf16>, dim3(grid), dim3(blk), 0, st, p);
    else hipLaunchKernelGGL(pool5_kernel<float>, dim3(grid), dim3(blk), 0, st, p);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// nn.Upsample(scale_factor=2, mode="nearest") into a channel slice of the consumer's concat buffer
// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void upsample2_kernel(const UpParams p) {
    const int groups = p.C >> 3;
    const int Ho = p.H * 2, Wo = p.W * 2;
    const long total = (long)p.B * Ho * Wo * groups;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int g = (int)(gid % groups);
    const long pix = gid / groups;
    const int x = (int)(pix % Wo);
    const int y = (int)((pix / Wo) % Ho);
    const int b = (int)(pix / ((long)Wo * Ho));
    Vec8<T> v;
    v.load((const T*)p.x + ((size_t)(b * p.H + (y >> 1)) * p.W + (x >> 1)) * p.x_stride + p.x_coff + g * 8);
    v.store((T*)p.y + (size_t)pix * p.y_stride + p.y_coff + g * 8);
}

hipError_t launch_upsample(const UpParams& p, int dtype, hipStream_t st) {
    const long total = (long)p.B * p.H * 2 * p.W * 2 * (p.C / 8);
    const int blk = 256;
    const unsigned grid = (unsigned)((total + blk - 1) / blk);
    if (dtype == DT_BF16) hipLaunchKernelGGL(upsample2_kernel<__bf16>, dim3(grid), dim3(blk), 0, st, p);
    else hipLaunchKernelGGL(upsample2_kernel<float>, dim3(grid), dim3(blk), 0, st, p);
    return hipGetLastError();
}

}  // namespace yp
