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


// ---------------------------------------------------------------------------------------------------------------
// MFMA form (bf16, key_dim 32, head_dim 64, N <= 400 tokens - every 640x640 v10 variant except M).
// One wave = 16 queries. S^T = K.Q^T is computed with K as the MFMA A operand and Q as B, so a lane ends up with ONE
// query (lane&15) and 4 consecutive keys per 16-key tile: the softmax reductions are in-lane plus two shuffles, and the
// probabilities are already laid out as the A operand of P.V (lane = query row, 8 k-slots per lane group) - the k order
// inside a 32-key step is the permutation {4g..4g+3, 16+4g..16+4g+3}, applied identically to the V fragment, which is
// read from a transposed LDS image Vt[d][key] (two ds_read_b64 per fragment, conflict-free with a 424-element row).
// ---------------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) __bf16 abf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 abf16x4;
typedef __attribute__((ext_vector_type(4))) float af32x4;

constexpr int A_NT = 25;            // key tiles of 16 (N <= 400)
constexpr int A_NPAD = 416;         // 26 tiles = 13 steps of 32 keys
constexpr int A_VS = 424;           // Vt row stride (elements)
constexpr int A_WAVES = 5;          // 80 queries per workgroup

__device__ __forceinline__ int aswz(int row) { return ((row >> 2) & 1) << 1; }

__global__ __launch_bounds__(A_WAVES * 64) void attention_mfma_kernel(const AttnParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[A_NPAD * 64 + 64 * A_VS * 2];
    unsigned char* Ks = lds;                                   // [A_NPAD keys][32] bf16, chunk-swizzled
    __bf16* Vt = (__bf16*)(lds + A_NPAD * 64);                 // [64 d][A_VS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bh = blockIdx.y, b = bh / p.nh, h = bh - b * p.nh;
    const int blk = 2 * p.kd + p.hd;
    const __bf16* base = (const __bf16*)p.qkv + (size_t)b * p.N * p.q_stride + p.q_coff + h * blk;

    // ---- stage K (rows = keys) and V transposed ---------------------------------------------------------------
    for (int i = tid; i < A_NPAD * 4; i += A_WAVES * 64) {
        const int key = i >> 2, c = i & 3;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (key < p.N) v = *(const uint4*)(base + (size_t)key * p.q_stride + p.kd + c * 8);
        *(uint4*)(Ks + key * 64 + ((c ^ aswz(key)) * 16)) = v;
    }
    for (int i = tid; i < A_NPAD * 8; i += A_WAVES * 64) {
        const int d8 = i / A_NPAD, key = i - d8 * A_NPAD;
        __attribute__((aligned(16))) __bf16 v[8];
        if (key < p.N) *(uint4*)v = *(const uint4*)(base + (size_t)key * p.q_stride + 2 * p.kd + d8 * 8);
        else *(uint4*)v = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 8; ++j) Vt[(d8 * 8 + j) * A_VS + key] = v[j];
    }
    __syncthreads();

    const int fr = lane & 15, g = lane >> 4;
    const int q0 = (blockIdx.x * A_WAVES + wave) * 16;
    if (q0 >= p.N) return;
    const int qi = q0 + fr;
    abf16x8 qf;
    if (qi < p.N) qf = *(const abf16x8*)(base + (size_t)qi * p.q_stride + g * 8);
    else {
#pragma unroll
        for (int j = 0; j < 8; ++j) qf[j] = (__bf16)0.f;
    }
    // ---- S^T tiles: rows = keys (A operand), cols = queries (B operand) ------------------------------------------------
    af32x4 st[A_NT];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < A_NT; ++j) {
        const int row = j * 16 + fr;
        const abf16x8 kf = *(const abf16x8*)(Ks + row * 64 + ((g ^ aswz(row)) * 16));
        af32x4 z = {0.f, 0.f, 0.f, 0.f};
        st[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf, z, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = j * 16 + g * 4 + r;
            const float v = (key < p.N) ? st[j][r] * p.scale : -INFINITY;
            st[j][r] = v;
            mx = fmaxf(mx, v);
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < A_NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float e = __expf(st[j][r] - mx);
            st[j][r] = e;
            sum += e;
        }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    // ---- O = P.V over 13 steps of 32 keys ---------------------------------------------------------------------------------
    af32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = af32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < A_NPAD / 32; ++s) {
        abf16x8 pf;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            pf[r] = (__bf16)(st[2 * s][r] * inv);
            pf[4 + r] = (2 * s + 1 < A_NT) ? (__bf16)(st[2 * s + 1 < A_NT ? 2 * s + 1 : 0][r] * inv) : (__bf16)0.f;
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const __bf16* vp = Vt + (dt * 16 + fr) * A_VS + s * 32 + g * 4;
            const abf16x4 lo = *(const abf16x4*)vp, hi = *(const abf16x4*)(vp + 16);
            abf16x8 vf;
#pragma unroll
            for (int r = 0; r < 4; ++r) { vf[r] = lo[r]; vf[4 + r] = hi[r]; }
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, vf, o[dt], 0, 0, 0);
        }
    }
    // D: col = d (lane&15), rows = queries g*4 + r
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int q = q0 + g * 4 + r;
        if (q >= p.N) continue;
        __bf16* op = (__bf16*)p.o + ((size_t)b * p.N + q) * p.o_stride + p.o_coff + h * p.hd;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) op[dt * 16 + fr] = (__bf16)o[dt][r];
    }
}

hipError_t launch_attention(const AttnParams& p, int dtype, hipStream_t st) {
    if (dtype == DT_BF16 && p.kd == 32 && p.hd == 64 && p.N <= 16 * A_NT && (p.q_stride & 7) == 0 && (p.q_coff & 7) == 0) {
        dim3 grid((p.N + A_WAVES * 16 - 1) / (A_WAVES * 16), p.B * p.nh);
        hipLaunchKernelGGL(attention_mfma_kernel, grid, dim3(A_WAVES * 64), 0, st, p);
        return hipGetLastError();
    }
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
