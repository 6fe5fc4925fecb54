// PSA attention core (SURVEY.md Appendix A.2 `Attention` [U]; runs inside `.predict`, reference
// yolo_seg/app.py:91):   A = softmax((q^T k) * kd^-0.5) over keys,   o[:, n] = sum_m v[:, m] * A[n, m]
// qkv is the NHWC output of the 1x1 `qkv` conv: per token n, per head h a block [q(kd) | k(kd) | v(hd)].
// One workgroup = (image, head, 16 query tokens): scores for the 16 rows live in LDS, K/V stream from L2.
// fp32 math; in bf16 mode the probabilities are rounded to bf16 before P.V (the oracle's bf16emu spec).
#include "common.h"

namespace yp {

constexpr int QT = 16;

template <typename T> __device__ __forceinline__ float rnd(float x);
template <> __device__ __forceinline__ float rnd<__bf16>(float x) { return (float)(__bf16)x; }
template <> __device__ __forceinline__ float rnd<float>(float x) { return x; }

template <typename T> __device__ __forceinline__ void load4(const T* p, float* f);
template <> __device__ __forceinline__ void load4<__bf16>(const __bf16* p, float* f) {
    const uint2 r = *(const uint2*)p;
    f[0] = __uint_as_float(r.x << 16); f[1] = __uint_as_float(r.x & 0xffff0000u);
    f[2] = __uint_as_float(r.y << 16); f[3] = __uint_as_float(r.y & 0xffff0000u);
}
template <> __device__ __forceinline__ void load4<float>(const float* p, float* f) {
    const float4 r = *(const float4*)p;
    f[0] = r.x; f[1] = r.y; f[2] = r.z; f[3] = r.w;
}
template <typename T> __device__ __forceinline__ void store4(T* p, const float* f);
template <> __device__ __forceinline__ void store4<__bf16>(__bf16* p, const float* f) {
    __attribute__((aligned(8))) __bf16 o[4] = {(__bf16)f[0], (__bf16)f[1], (__bf16)f[2], (__bf16)f[3]};
    *(uint2*)p = *(const uint2*)o;
}
template <> __device__ __forceinline__ void store4<float>(float* p, const float* f) {
    *(float4*)p = make_float4(f[0], f[1], f[2], f[3]);
}

template <typename T>
__global__ __launch_bounds__(256) void attention_kernel(const AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Qs = lds;                 // [QT][kd]
    float* S = lds + QT * p.kd;      // [QT][N]
    const int tid = threadIdx.x;
    const int qt = blockIdx.x, bh = blockIdx.y;
    const int b = bh / p.nh, h = bh - b * p.nh;
    const int blk = 2 * p.kd + p.hd;
    const int n0 = qt * QT;
    const T* base = (const T*)p.qkv + (size_t)b * p.N * p.q_stride + p.q_coff + h * blk;

    // stage the 16 query rows (pre-scaled) as fp32
    for (int i = tid; i < QT * p.kd; i += 256) {
        const int q = i / p.kd, j = i - q * p.kd;
        const int n = n0 + q;
        Qs[i] = (n < p.N) ? (float)base[(size_t)n * p.q_stride + j] : 0.f;
    }
    __syncthreads();

    // scores: each thread owns keys n = tid, tid+256, ... for all 16 queries
    for (int n = tid; n < p.N; n += 256) {
        const T* kp = base + (size_t)n * p.q_stride + p.kd;
        float acc[QT];
#pragma unroll
        for (int q = 0; q < QT; ++q) acc[q] = 0.f;
        for (int jc = 0; jc < p.kd; jc += 4) {
            float kv[4];
            load4<T>(kp + jc, kv);
#pragma unroll
            for (int q = 0; q < QT; ++q) {
                const float* qp = Qs + q * p.kd + jc;
                acc[q] = fmaf(qp[0], kv[0], acc[q]);
                acc[q] = fmaf(qp[1], kv[1], acc[q]);
                acc[q] = fmaf(qp[2], kv[2], acc[q]);
                acc[q] = fmaf(qp[3], kv[3], acc[q]);
            }
        }
#pragma unroll
        for (int q = 0; q < QT; ++q) S[q * p.N + n] = acc[q] * p.scale;
    }
    __syncthreads();

    // softmax over keys: 16 threads per query row
    {
        const int q = tid >> 4, l = tid & 15;
        float* row = S + q * p.N;
        float mx = -INFINITY;
        for (int n = l; n < p.N; n += 16) mx = fmaxf(mx, row[n]);
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 16));
        float sum = 0.f;
        for (int n = l; n < p.N; n += 16) {
            const float e = expf(row[n] - mx);
            row[n] = e;
            sum += e;
        }
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) sum += __shfl_xor(sum, o, 16);
        const float inv = 1.0f / sum;
        for (int n = l; n < p.N; n += 16) row[n] = rnd<T>(row[n] * inv);
    }
    __syncthreads();

    // o[q][d] = sum_n P[q][n] * v[n][d] ; work item = (q, 4-wide d chunk)
    const int nch = p.hd >> 2;
    for (int item = tid; item < QT * nch; item += 256) {
        const int q = item / nch, dc = item - q * nch;
        const int n = n0 + q;
        if (n >= p.N) continue;
        const float* row = S + q * p.N;
        const T* vp = base + 2 * p.kd + dc * 4;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int m = 0; m < p.N; ++m) {
            float vv[4];
            load4<T>(vp + (size_t)m * p.q_stride, vv);
            const float pr = row[m];
            acc[0] = fmaf(pr, vv[0], acc[0]);
            acc[1] = fmaf(pr, vv[1], acc[1]);
            acc[2] = fmaf(pr, vv[2], acc[2]);
            acc[3] = fmaf(pr, vv[3], acc[3]);
        }
        store4<T>((T*)p.o + ((size_t)b * p.N + n) * p.o_stride + p.o_coff + h * p.hd + dc * 4, acc);
    }
}

hipError_t launch_attention(const AttnParams& p, int dtype, hipStream_t st) {
    const size_t sh = (size_t)(QT * p.kd + QT * p.N) * sizeof(float);
    if (sh > 150 * 1024 || (p.kd & 3) || (p.hd & 3)) return hipErrorInvalidValue;
    dim3 grid((p.N + QT - 1) / QT, p.B * p.nh);
    if (dtype == DT_BF16) {
        if (sh > 64 * 1024)
            (void)hipFuncSetAttribute((const void*)attention_kernel<__bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        hipLaunchKernelGGL(attention_kernel<__bf16>, grid, dim3(256), sh, st, p);
    } else {
        if (sh > 64 * 1024)
            (void)hipFuncSetAttribute((const void*)attention_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        hipLaunchKernelGGL(attention_kernel<float>, grid, dim3(256), sh, st, p);
    }
    return hipGetLastError();
}

}  // namespace yp
