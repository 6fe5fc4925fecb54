// HBM-bound helpers of the YOLOv10 graph: stem conv (u8 BGR -> first feature map), depthwise convs,
// SPPF 5x5 max-pool, nearest x2 upsample. All NHWC, 8 channels (one 16-B bf16 vector) per thread, fp32 math.
// Blocks: SURVEY.md Appendix A.2 [U] (`Conv` with g=c, `SCDown.cv2`, `CIB`, `RepVGGDW`, `SPPF.m`,
// `nn.Upsample`), executed inside `.predict` (reference yolo_seg/app.py:91).
#include "common.h"

namespace yp {

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + expf(-x)); }   // fp32 parity mode: library expf, IEEE division
// bf16 storage: v_rcp_f32 (1 ulp) instead of the IEEE division sequence (~12 instructions) - same form as the conv epilogues
__device__ __forceinline__ float silu_q(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
template <typename T> __device__ __forceinline__ float silu_t(float x) { return sizeof(T) == 2 ? silu_q(x) : silu_f(x); }

template <typename T> struct Vec8;
template <> struct Vec8<__bf16> {
    uint4 raw;
    __device__ inline void load(const __bf16* p) { raw = *(const uint4*)p; }
    __device__ inline void load_buf(__amdgpu_buffer_rsrc_t rs, unsigned off) {
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
        raw = make_uint4(v[0], v[1], v[2], v[3]);
    }
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
    __device__ inline void load_buf(__amdgpu_buffer_rsrc_t rs, unsigned off) {
        const auto v0 = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
        const auto v1 = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 16, 0, 0);
        a = make_float4(__uint_as_float(v0[0]), __uint_as_float(v0[1]), __uint_as_float(v0[2]), __uint_as_float(v0[3]));
        b = make_float4(__uint_as_float(v1[0]), __uint_as_float(v1[1]), __uint_as_float(v1[2]), __uint_as_float(v1[3]));
    }
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

// ---------------------------------------------------------------------------------------------------------
// bf16 stem on the matrix cores: one workgroup = 256 output pixels. Each thread builds the im2col row of its pixel
// (27 taps of uint8 -> x/255 -> bf16, zero-padded to K=32) straight into LDS, then 4 waves run
// D[cout][pixel] = W[cout][32] * X[pixel][32] with MFMA 16x16x32 and store 4 consecutive couts per lane.
// ---------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
__device__ __forceinline__ int swz32(int row) { return ((row >> 2) & 1) << 1; }

template <int FN>
__global__ __launch_bounds__(256) void stem_mfma_kernel(const StemParams p, const __bf16* __restrict__ wpk) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[256 * 64 + FN * 16 * 64];
    unsigned char* Xs = lds;
    unsigned char* Ws = lds + 256 * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int M = p.B * p.Ho * p.Wo;                     // < 2^31 (checked by the launcher): 32-bit pixel arithmetic
    const int m = (int)blockIdx.x * 256 + tid;
    // ---- im2col row of this thread's pixel ---------------------------------------------------------------
    {
        __attribute__((aligned(16))) __bf16 row[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) row[i] = (__bf16)0.f;
        if (m < M) {
            const int wo = m % p.Wo;
            const int ho = (m / p.Wo) % p.Ho;
            const int b = m / (p.Wo * p.Ho);
            const uint8_t* xb = p.x + (size_t)b * p.H * p.W * 3;
            const int rowbytes = p.W * 3;                         // multiple of 4 (W is a multiple of 32)
            const int sb = (wo * 2 - 1) * 3;                      // first byte of the 9-byte (3 px x BGR) segment; -3 at wo=0
            const int ab = sb & ~3;                               // aligned dword holding it (floor, also for -3 -> -4)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int hi = ho * 2 - 1 + ky;
                const bool rok = (unsigned)hi < (unsigned)p.H;
                const uint8_t* rp = xb + (size_t)(rok ? hi : 0) * rowbytes;
                unsigned d[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {                      // three aligned dwords cover the 9 bytes; out-of-row dwords read as 0
                    const int o = ab + 4 * k;
                    const bool ok = rok && o >= 0 && o < rowbytes;
                    d[k] = *(const unsigned*)(rp + (ok ? o : 0));
                    d[k] = ok ? d[k] : 0u;
                }
                // the 9 bytes (3 px x BGR) of this tap row, byte-aligned into three dwords (v_alignbyte_b32); a pixel left
                // of the image (kx = 0 at wo = 0) is zero padding. The right neighbour 2*wo+1 is always inside (W even).
                const unsigned sbytes = (unsigned)(sb - ab);
                unsigned w0 = __builtin_amdgcn_alignbyte(d[1], d[0], sbytes);      // bytes 0..3
                const unsigned w1 = __builtin_amdgcn_alignbyte(d[2], d[1], sbytes); // bytes 4..7
                const unsigned w2 = __builtin_amdgcn_alignbyte(0u, d[2], sbytes);   // byte 8
                if (wo == 0) w0 &= 0xff000000u;
                // x / 255 then bf16: for all 256 byte values bf16(x * fl(1/255)) == bf16(x / 255) (checked exhaustively), so the
                // bf16 stem multiplies; byte -> float is one v_cvt_f32_ubyteN
                const float k = 1.0f / 255.0f;
                __bf16* rr = row + ky * 9;
                rr[0] = (__bf16)((float)(w0 & 0xffu) * k);          rr[1] = (__bf16)((float)((w0 >> 8) & 0xffu) * k);
                rr[2] = (__bf16)((float)((w0 >> 16) & 0xffu) * k);  rr[3] = (__bf16)((float)(w0 >> 24) * k);
                rr[4] = (__bf16)((float)(w1 & 0xffu) * k);          rr[5] = (__bf16)((float)((w1 >> 8) & 0xffu) * k);
                rr[6] = (__bf16)((float)((w1 >> 16) & 0xffu) * k);  rr[7] = (__bf16)((float)(w1 >> 24) * k);
                rr[8] = (__bf16)((float)(w2 & 0xffu) * k);
            }
        }
        const uint4* r4 = (const uint4*)row;
#pragma unroll
        for (int c = 0; c < 4; ++c) *(uint4*)(Xs + tid * 64 + ((c ^ swz32(tid)) * 16)) = r4[c];
    }
    for (int i = tid; i < FN * 16 * 4; i += 256) {
        const int r = i >> 2, c = i & 3;
        *(uint4*)(Ws + r * 64 + ((c ^ swz32(r)) * 16)) = *(const uint4*)(wpk + r * 32 + c * 8);
    }
    __syncthreads();
    const int fr = lane & 15, fc = lane >> 4;
    bf16x8_t wf[FN], xf[4];
#pragma unroll
    for (int a = 0; a < FN; ++a) { const int r = a * 16 + fr; wf[a] = *(const bf16x8_t*)(Ws + r * 64 + ((fc ^ swz32(r)) * 16)); }
#pragma unroll
    for (int b = 0; b < 4; ++b) { const int r = wave * 64 + b * 16 + fr; xf[b] = *(const bf16x8_t*)(Xs + r * 64 + ((fc ^ swz32(r)) * 16)); }
#pragma unroll
    for (int a = 0; a < FN; ++a) {
        const int co = a * 16 + fc * 4;
        const float4 bs = *(const float4*)(p.bias + co);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[a], xf[b], acc, 0, 0, 0);
            const int mm = (int)blockIdx.x * 256 + wave * 64 + b * 16 + fr;
            if (mm >= M) continue;
            float v[4] = {acc[0] + bs.x, acc[1] + bs.y, acc[2] + bs.z, acc[3] + bs.w};
            if (p.act == ACT_SILU) silu4_packed(v);
            __attribute__((aligned(8))) __bf16 o[4] = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
            *(uint2*)((__bf16*)p.y + (size_t)mm * p.y_stride + p.y_coff + co) = *(const uint2*)o;
        }
    }
}

hipError_t launch_stem(const StemParams& p, int dtype, hipStream_t st) {
    const long total = (long)p.B * p.Ho * p.Wo * (p.C0 / 8);
    const int blk = 256;
    if (dtype == DT_BF16 && p.wpk && (p.C0 % 16) == 0 && p.C0 <= 80 && (p.y_stride & 3) == 0 && (p.y_coff & 3) == 0 &&
        (long)p.B * p.Ho * p.Wo + 256 < (1l << 31)) {
        const long M = (long)p.B * p.Ho * p.Wo;
        const unsigned grid = (unsigned)((M + 255) / 256);
        const __bf16* w = (const __bf16*)p.wpk;
        switch (p.C0 / 16) {
            case 1: hipLaunchKernelGGL(stem_mfma_kernel<1>, dim3(grid), dim3(256), 0, st, p, w); break;
            case 2: hipLaunchKernelGGL(stem_mfma_kernel<2>, dim3(grid), dim3(256), 0, st, p, w); break;
            case 3: hipLaunchKernelGGL(stem_mfma_kernel<3>, dim3(grid), dim3(256), 0, st, p, w); break;
            case 4: hipLaunchKernelGGL(stem_mfma_kernel<4>, dim3(grid), dim3(256), 0, st, p, w); break;
            default: hipLaunchKernelGGL(stem_mfma_kernel<5>, dim3(grid), dim3(256), 0, st, p, w); break;
        }
        return hipGetLastError();
    }
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
        for (int j = 0; j < 8; ++j) acc[j] = silu_t<T>(acc[j]);
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

// NOUT consecutive output pixels of one row per thread: the input columns are loaded once per kernel row and reused
// by every output they cover, the weights of a kernel row are loaded once per thread.
template <typename T, int KS, int S, int NOUT>
__global__ __launch_bounds__(256) void dwconv_row_kernel(const DwParams p) {
    constexpr int NCOL = (NOUT - 1) * S + KS;
    const int groups = p.C >> 3;
    const int wq = (p.Wo + NOUT - 1) / NOUT;
    const long total = (long)p.B * p.Ho * wq * groups;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int g = (int)(gid % groups);
    long t = gid / groups;
    const int wo0 = (int)(t % wq) * NOUT;
    t /= wq;
    const int ho = (int)(t % p.Ho);
    const int b = (int)(t / p.Ho);
    const int c = g * 8;
    const int cin = p.gs ? (p.x_coff + (c / p.gs) * p.gstride + (c % p.gs)) : (p.x_coff + c);
    float acc[NOUT][8];
#pragma unroll
    for (int o = 0; o < NOUT; ++o)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[o][j] = p.bias[c + j];
    const T* wb = (const T*)p.w + c;
    const int wi0 = wo0 * S - p.pad;
    // taps outside the image are read through a buffer descriptor with an out-of-range offset (hardware returns 0):
    // no clamping, no masking arithmetic, and all loads of a kernel row issue back to back
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
    const unsigned xbase = (unsigned)(((size_t)b * p.H * p.W * p.x_stride + cin) * sizeof(T));
    constexpr int KY_UNROLL = (KS == 7) ? 1 : KS;      // 7x7: keep one kernel row of loads in flight (register budget)
#pragma unroll KY_UNROLL
    for (int ky = 0; ky < KS; ++ky) {
        const int hi = ho * S - p.pad + ky;
        const bool rok = (unsigned)hi < (unsigned)p.H;
        Vec8<T> xv[NCOL], wv[KS];
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) wv[kx].load(wb + (size_t)(ky * KS + kx) * p.C);
#pragma unroll
        for (int col = 0; col < NCOL; ++col) {
            const int wi = wi0 + col;
            const bool ok = rok && ((unsigned)wi < (unsigned)p.W);
            const unsigned off = ok ? xbase + (unsigned)((hi * p.W + wi) * p.x_stride) * (unsigned)sizeof(T) : 0x80000000u;
            xv[col].load_buf(xrs, off);
        }
        float wf[KS][8];
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) wv[kx].unpack(wf[kx]);
#pragma unroll
        for (int col = 0; col < NCOL; ++col) {
            float xf[8];
            xv[col].unpack(xf);
#pragma unroll
            for (int o = 0; o < NOUT; ++o) {
                const int kx = col - o * S;
                if (kx >= 0 && kx < KS) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[o][j] = fmaf(xf[j], wf[kx][j], acc[o][j]);
                }
            }
        }
    }
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
        const int wo = wo0 + o;
        if (wo >= p.Wo) break;
        const size_t pix = ((size_t)b * p.Ho + ho) * p.Wo + wo;
        if (p.act == ACT_SILU) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[o][j] = silu_t<T>(acc[o][j]);
        }
        if (p.res) {
            Vec8<T> rv;
            rv.load((const T*)p.res + pix * p.res_stride + p.res_coff + c);
            float rf[8];
            rv.unpack(rf);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[o][j] += rf[j];
        }
        Vec8<T> ov;
        ov.pack(acc[o]);
        ov.store((T*)p.y + pix * p.y_stride + p.y_coff + c);
    }
}

template <typename T, int KS, int S, int NOUT>
static void launch_dw_row(const DwParams& p, hipStream_t st) {
    const long total = (long)p.B * p.Ho * ((p.Wo + NOUT - 1) / NOUT) * (p.C / 8);
    hipLaunchKernelGGL((dwconv_row_kernel<T, KS, S, NOUT>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, p);
}

template <typename T>
static hipError_t launch_dwconv_t(const DwParams& p, hipStream_t st) {
    const bool small = p.x_bytes < (1ull << 31);      // 32-bit buffer offsets
    if (small && p.ks == 3 && p.stride == 1) launch_dw_row<T, 3, 1, 4>(p, st);
    else if (small && p.ks == 3 && p.stride == 2) launch_dw_row<T, 3, 2, 2>(p, st);
    else if (small && p.ks == 7 && p.stride == 1) launch_dw_row<T, 7, 1, 2>(p, st);
    else {
        const long total = (long)p.B * p.Ho * p.Wo * (p.C / 8);
        hipLaunchKernelGGL(dwconv_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, p);
    }
    return hipGetLastError();
}

hipError_t launch_dwconv(const DwParams& p, int dtype, hipStream_t st) {
    if (dwconv_mfma_valid(p, dtype)) return launch_dwconv_mfma(p, st);
    return dtype == DT_BF16 ? launch_dwconv_t<__bf16>(p, st) : launch_dwconv_t<float>(p, st);
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
// SPPF: the three chained MaxPool2d(5,1,2) in ONE launch. A workgroup owns (image, 8-channel group): the H x W tile
// lives in LDS as packed bf16x8 / f32x8 vectors and is pooled three times in place (two buffers), each result written
// to its slice of the SPPF concat buffer. Max is exact in any precision, so this equals the chained reference bit for bit.
// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void sppf_pool3_kernel(const PoolParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char plds[];
    constexpr int VB = 8 * sizeof(T);
    const int HW = p.H * p.W;
    unsigned char* buf[2] = {plds, plds + (size_t)HW * VB};
    const int groups = p.C >> 3;
    const int g = blockIdx.x % groups, b = blockIdx.x / groups;
    const T* xb = (const T*)p.x + (size_t)b * HW * p.x_stride + p.x_coff + g * 8;
    for (int i = threadIdx.x; i < HW; i += 256) {
        Vec8<T> v;
        v.load(xb + (size_t)i * p.x_stride);
        *(Vec8<T>*)(buf[0] + (size_t)i * VB) = v;
    }
    __syncthreads();
    for (int stage = 0; stage < 3; ++stage) {
        const unsigned char* src = buf[stage & 1];
        unsigned char* dst = buf[(stage + 1) & 1];
        T* yb = (T*)p.y + (size_t)b * HW * p.y_stride + p.y_coff + stage * p.C + g * 8;
        for (int i = threadIdx.x; i < HW; i += 256) {
            const int y = i / p.W, x = i - y * p.W;
            float m[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) m[j] = -INFINITY;
            for (int dy = -2; dy <= 2; ++dy) {
                const int yy = y + dy;
                if ((unsigned)yy >= (unsigned)p.H) continue;
                for (int dx = -2; dx <= 2; ++dx) {
                    const int xx = x + dx;
                    if ((unsigned)xx >= (unsigned)p.W) continue;
                    const Vec8<T> v = *(const Vec8<T>*)(src + (size_t)(yy * p.W + xx) * VB);
                    float f[8];
                    v.unpack(f);
#pragma unroll
                    for (int j = 0; j < 8; ++j) m[j] = fmaxf(m[j], f[j]);
                }
            }
            Vec8<T> o;
            o.pack(m);
            *(Vec8<T>*)(dst + (size_t)i * VB) = o;
            o.store(yb + (size_t)i * p.y_stride);
        }
        __syncthreads();
    }
}

// bf16 fast form: 32 channels (four 16-B pieces) of one image per workgroup, separable 5x5 max (row pass, column pass),
// values kept in LDS as order-preserving int16 keys (k = x ^ ((x >> 15) & 0x7fff), an involution) so that one
// v_pk_max_i16 handles two channels; clamped neighbour indices replace the -inf padding (max ignores duplicates).
typedef short short2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned bf16x2_key(unsigned d) {
    const unsigned s = (d >> 15) & 0x00010001u;
    return d ^ ((s << 15) - s);
}
__device__ __forceinline__ uint4 key4(const uint4 v) { return make_uint4(bf16x2_key(v.x), bf16x2_key(v.y), bf16x2_key(v.z), bf16x2_key(v.w)); }
__device__ __forceinline__ unsigned pkmax(unsigned a, unsigned b) {
    const short2v r = __builtin_elementwise_max(__builtin_bit_cast(short2v, a), __builtin_bit_cast(short2v, b));
    return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ uint4 pkmax4(const uint4 a, const uint4 b) {
    return make_uint4(pkmax(a.x, b.x), pkmax(a.y, b.y), pkmax(a.z, b.z), pkmax(a.w, b.w));
}

__global__ __launch_bounds__(256) void sppf_pool3_bf16_kernel(const PoolParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char plds[];
    const int HW = p.H * p.W, W = p.W, H = p.H;
    unsigned char* const bufA = plds;
    unsigned char* const bufB = plds + (size_t)HW * 64;
    unsigned char* const tmp = plds + (size_t)HW * 128;
    const int groups = p.C >> 5;
    const int g = blockIdx.x % groups, b = blockIdx.x / groups;
    const __bf16* xb = (const __bf16*)p.x + (size_t)b * HW * p.x_stride + p.x_coff + g * 32;
    const int items = HW * 4;
    for (int i = threadIdx.x; i < items; i += 256) {
        const uint4 v = *(const uint4*)(xb + (size_t)(i >> 2) * p.x_stride + (i & 3) * 8);
        *(uint4*)(bufA + i * 16) = key4(v);
    }
    __syncthreads();
    for (int stage = 0; stage < 3; ++stage) {
        const unsigned char* src = (stage & 1) ? bufB : bufA;
        unsigned char* dst = (stage & 1) ? bufA : bufB;
        for (int i = threadIdx.x; i < items; i += 256) {          // row pass
            const int px = i >> 2, y = px / W, x = px - y * W;
            const int x0 = max(x - 2, 0), x1 = max(x - 1, 0), x3 = min(x + 1, W - 1), x4 = min(x + 2, W - 1);
            const unsigned char* row = src + (size_t)(y * W) * 64 + (i & 3) * 16;
            uint4 m = *(const uint4*)(row + x * 64);
            m = pkmax4(m, *(const uint4*)(row + x0 * 64));
            m = pkmax4(m, *(const uint4*)(row + x1 * 64));
            m = pkmax4(m, *(const uint4*)(row + x3 * 64));
            m = pkmax4(m, *(const uint4*)(row + x4 * 64));
            *(uint4*)(tmp + i * 16) = m;
        }
        __syncthreads();
        __bf16* yb = (__bf16*)p.y + (size_t)b * HW * p.y_stride + p.y_coff + stage * p.C + g * 32;
        for (int i = threadIdx.x; i < items; i += 256) {          // column pass + store
            const int px = i >> 2, y = px / W, x = px - y * W;
            const int y0 = max(y - 2, 0), y1 = max(y - 1, 0), y3 = min(y + 1, H - 1), y4 = min(y + 2, H - 1);
            const unsigned char* col = tmp + (size_t)x * 64 + (i & 3) * 16;
            uint4 m = *(const uint4*)(col + (size_t)(y * W) * 64);
            m = pkmax4(m, *(const uint4*)(col + (size_t)(y0 * W) * 64));
            m = pkmax4(m, *(const uint4*)(col + (size_t)(y1 * W) * 64));
            m = pkmax4(m, *(const uint4*)(col + (size_t)(y3 * W) * 64));
            m = pkmax4(m, *(const uint4*)(col + (size_t)(y4 * W) * 64));
            *(uint4*)(dst + i * 16) = m;
            *(uint4*)(yb + (size_t)px * p.y_stride + (i & 3) * 8) = key4(m);
        }
        __syncthreads();
    }
}

bool sppf_pool3_fits(const PoolParams& p, int dtype) {
    if (dtype == DT_BF16 && (p.C & 31) == 0 && (p.x_stride & 7) == 0 && (p.x_coff & 7) == 0 && (p.y_stride & 7) == 0 && (p.y_coff & 7) == 0)
        return (size_t)p.H * p.W * 192 <= 128 * 1024;
    return (size_t)2 * p.H * p.W * 8 * (dtype == DT_BF16 ? 2 : 4) <= 64 * 1024;
}

// p.y/y_coff = first pooled slice; the three results go to consecutive C-channel slices
hipError_t launch_sppf_pool3(const PoolParams& p, int dtype, hipStream_t st) {
    if (!sppf_pool3_fits(p, dtype)) return hipErrorInvalidValue;
    if (dtype == DT_BF16 && (p.C & 31) == 0 && (p.x_stride & 7) == 0 && (p.x_coff & 7) == 0 && (p.y_stride & 7) == 0 && (p.y_coff & 7) == 0) {
        const size_t sh = (size_t)p.H * p.W * 192;
        static bool attr = false;
        if (!attr) {
            hipError_t e = hipFuncSetAttribute((const void*)sppf_pool3_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
            if (e != hipSuccess) return e;
            attr = true;
        }
        hipLaunchKernelGGL(sppf_pool3_bf16_kernel, dim3((unsigned)(p.B * (p.C / 32))), dim3(256), sh, st, p);
        return hipGetLastError();
    }
    const size_t es = dtype == DT_BF16 ? 2 : 4;
    const size_t sh = (size_t)2 * p.H * p.W * 8 * es;
    const unsigned grid = (unsigned)(p.B * (p.C / 8));
    if (dtype == DT_BF16) hipLaunchKernelGGL(sppf_pool3_kernel<__bf16>, dim3(grid), dim3(256), sh, st, p);
    else hipLaunchKernelGGL(sppf_pool3_kernel<float>, dim3(grid), dim3(256), sh, st, p);
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

// ---------------------------------------------------------------------------------------------------------
// The forward's results leave the engine-owned buffers in ONE launch (instead of up to three device-to-device memcpy
// commands behind the graph replay): dword copies of det [rows,6], idx [rows] and coeff [rows,32]; null destinations skip.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void copy_out_kernel(const unsigned* __restrict__ s0, unsigned* __restrict__ d0, unsigned n0,
                                                       const unsigned* __restrict__ s1, unsigned* __restrict__ d1, unsigned n1,
                                                       const unsigned* __restrict__ s2, unsigned* __restrict__ d2, unsigned n2) {
    unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i < n0) { d0[i] = s0[i]; return; }
    i -= n0;
    if (i < n1) { d1[i] = s1[i]; return; }
    i -= n1;
    if (i < n2) d2[i] = s2[i];
}

hipError_t launch_copy_out(const float* det, float* det_out, const int32_t* idx, int32_t* idx_out, const float* coeff, float* coeff_out,
                           size_t rows, hipStream_t st) {
    const unsigned n0 = det_out ? (unsigned)(rows * 6) : 0u, n1 = idx_out ? (unsigned)rows : 0u, n2 = coeff_out ? (unsigned)(rows * 32) : 0u;
    const unsigned total = n0 + n1 + n2;
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(copy_out_kernel, dim3((total + 255u) / 256u), dim3(256), 0, st, (const unsigned*)det, (unsigned*)det_out, n0,
                       (const unsigned*)idx, (unsigned*)idx_out, n1, (const unsigned*)coeff, (unsigned*)coeff_out, n2);
    return hipGetLastError();
}

}  // namespace yp
