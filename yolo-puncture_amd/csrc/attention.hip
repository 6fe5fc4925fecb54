// PSA attention core (SURVEY.md Appendix A.2 `Attention` [U]; runs inside `.predict`, reference
// yolo_seg/app.py:91):   A = softmax((q^T k) * kd^-0.5) over keys,   o[:, n] = sum_m v[:, m] * A[n, m]
// qkv is the NHWC output of the 1x1 `qkv` conv: per token n, per head h a block [q(kd) | k(kd) | v(hd)].
// One workgroup = (image, head, 16 query tokens): scores for the 16 rows live in LDS, K/V stream from L2.
// fp32 math; in bf16 mode the probabilities are rounded to bf16 before P.V (the oracle's bf16emu spec).
#include "common.h"
#include <algorithm>
#include <cstdlib>

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
// One wave = 16 queries at a time. S^T = K.Q^T is computed with K as the MFMA A operand and Q as B, so a lane ends up with ONE
// query (lane&15) and 4 consecutive keys per 16-key tile: the softmax reductions are in-lane plus two shuffles, and the
// probabilities are already laid out as the A operand of P.V (lane = query row, 8 k-slots per lane group) - the k order
// inside a 32-key step is the permutation {4g..4g+3, 16+4g..16+4g+3}, applied identically to the V fragment.
//
// One workgroup = (image, head, a contiguous run of query tiles): K and V of the head are staged ONCE per workgroup by LDS-DMA,
// both row-major as they lie in HBM ([key][32] and [key][64], 16-B chunks swizzled), and the workgroup's waves walk its query
// tiles. The P.V B operand (8 keys of one d per lane) comes out of the row-major V image through the transposing LDS read
// `ds_read_b64_tr_b16` (a 16-lane group reads 4 key rows x 16 d and each lane receives one d of the 4 keys) - two reads per
// fragment, conflict-free with the 32-B pair swizzle `(key>>1)&3`. The first version restaged K and a register-transposed V
// per 80 queries (5 workgroups per head: 3.7x the algorithmic bytes from L2 misses across XCDs, 83 scalar LDS writes per
// thread); the run length is picked by the launcher so that the grid still covers the chip.
// ---------------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) __bf16 abf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 abf16x4;
typedef __attribute__((ext_vector_type(4))) short as16x4;
typedef __attribute__((ext_vector_type(4))) float af32x4;
typedef __attribute__((address_space(3))) void a_lds_void;
typedef __attribute__((address_space(3))) as16x4 a_lds_s4;

constexpr int A_NT = 25;            // key tiles of 16 (N <= 400)
constexpr int A_NPAD = 416;         // 26 tiles = 13 steps of 32 keys
constexpr int A_MAXW = 8;           // waves per workgroup

__device__ __forceinline__ int aswz(int row) { return ((row >> 2) & 1) << 1; }      // K image: 64-B rows
__device__ __forceinline__ int vswz(int row) { return ((row >> 1) & 3) << 1; }      // V image: 128-B rows, 32-B pairs swizzled

__global__ __launch_bounds__(A_MAXW * 64) void attention_mfma_kernel(const AttnParams p, const int tiles_per_wg, const unsigned qkv_bytes) {
    constexpr unsigned OOB = 0x80000000u;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[A_NPAD * 64 + A_NPAD * 128];
    unsigned char* const Ks = lds;                             // [A_NPAD keys][32] bf16, chunk-swizzled
    unsigned char* const Vs = lds + A_NPAD * 64;               // [A_NPAD keys][64] bf16, pair-swizzled
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int nw = A_MAXW;
    const int bh = blockIdx.y, b = bh / p.nh, h = bh - b * p.nh;
    const int blk = 2 * p.kd + p.hd;
    const size_t base_el = (size_t)b * p.N * p.q_stride + p.q_coff + h * blk;
    const __bf16* base = (const __bf16*)p.qkv + base_el;

    // ---- stage K and V rows of this head (keys >= N: out-of-range offset = zero fill) ---------------------------------
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.qkv, 0, (int)qkv_bytes, 0x00020000);
    for (int ii = wave; ii < A_NPAD / 16; ii += nw) {          // 16 keys x 64 B per instruction
        const int key = ii * 16 + (lane >> 2), c = (lane & 3) ^ aswz(key);
        const unsigned voff = (key < p.N) ? (unsigned)((base_el + (size_t)key * p.q_stride + p.kd + c * 8) * 2) : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (a_lds_void*)(Ks + ii * 1024), 16, voff, 0, 0, 0);
    }
    // the first tile's query fragment, issued between the K and V rows so that the counted wait below covers it (an asm load: the
    // compiler would wait for it with vmcnt(0), i.e. for the V rows behind it as well)
    const int fr = lane & 15, g = lane >> 4;
    const int ntiles = (p.N + 15) >> 4;
    const int t0 = blockIdx.x * tiles_per_wg, t1 = min(t0 + tiles_per_wg, ntiles);
    abf16x8 qf0;
    {
        const __bf16* qp = base + (size_t)min((t0 + wave) * 16 + fr, p.N - 1) * p.q_stride + g * 8;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(qf0) : "v"(qp) : "memory");
    }
    constexpr int NV = (A_NPAD / 8 + A_MAXW - 1) / A_MAXW;     // V instructions per wave: the same count in every wave (a surplus
#pragma unroll                                                  // slot re-issues the last group), so ONE counted wait serves all
    for (int k = 0; k < NV; ++k) {                              // 8 keys x 128 B per instruction
        const int ii = min(wave + k * A_MAXW, A_NPAD / 8 - 1);
        const int key = ii * 8 + (lane >> 3), c = (lane & 7) ^ vswz(key);
        const unsigned voff = (key < p.N) ? (unsigned)((base_el + (size_t)key * p.q_stride + 2 * p.kd + c * 8) * 2) : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (a_lds_void*)(Vs + ii * 1024), 16, voff, 0, 0, 0);
    }
    // K (and the query fragment) first: the V rows may still be in flight while the first tile's scores and softmax run (they are
    // waited for, once per wave, in front of its first P.V)
    static_assert(NV == 7, "the counted wait below is written for 7 V instructions per wave");
    asm volatile("s_waitcnt vmcnt(7)" : "+v"(qf0) :: "memory");
    __builtin_amdgcn_s_barrier();
    bool v_ready = false;

    const float cexp = p.scale * 1.44269504088896341f;          // exp((s - m) * scale) = exp2((s - m) * scale * log2 e)
    // transposed-read addresses: lane 4q+pp of group g supplies key row 4g+q (+16 for the second half of a 32-key step), d columns
    // dt*16 + 4pp .. +3 -> 16-B chunk 2dt + (pp>>1), 8-B half pp&1
    const unsigned char* vaddr[4];
    {
        const int q = (lane >> 2) & 3, pp = lane & 3, row = 4 * g + q;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
            vaddr[dt] = Vs + row * 128 + (((2 * dt + (pp >> 1)) ^ vswz(row)) * 16) + 8 * (pp & 1);
    }

    for (int t = t0 + wave; t < t1; t += nw) {                  // wave-uniform: the transposing reads below need EXEC all ones
        const int q0 = t * 16;
        const int qi = q0 + fr;
        abf16x8 qf;
        if (!v_ready) qf = qf0;                                  // (uniform) the wave's first tile
        else qf = *(const abf16x8*)(base + (size_t)min(qi, p.N - 1) * p.q_stride + g * 8);
        if (qi >= p.N) {
#pragma unroll
            for (int j = 0; j < 8; ++j) qf[j] = (__bf16)0.f;
        }
        // ---- S^T tiles: rows = keys (A operand), cols = queries (B operand) --------------------------------------------
        af32x4 st[A_NT];
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < A_NT; ++j) {
            const int row = j * 16 + fr;
            const abf16x8 kf = *(const abf16x8*)(Ks + row * 64 + ((g ^ aswz(row)) * 16));
            af32x4 z = {0.f, 0.f, 0.f, 0.f};
            st[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf, z, 0, 0, 0);
            if (j * 16 + 16 > p.N) {                             // (uniform) only a ragged or absent key tile needs masking
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (j * 16 + g * 4 + r >= p.N) st[j][r] = -INFINITY;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[j][r]);  // scale > 0: the max of the raw scores is the max of the scaled ones
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < A_NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __builtin_amdgcn_exp2f((st[j][r] - mx) * cexp);
                st[j][r] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
        // ---- O = P.V over 13 steps of 32 keys --------------------------------------------------------------------------
        if (!v_ready) {
            __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (0xF << 8));
            __builtin_amdgcn_s_barrier();
            v_ready = true;
        }
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
                const as16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((a_lds_s4*)(vaddr[dt] + s * 4096));
                const as16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((a_lds_s4*)(vaddr[dt] + s * 4096 + 2048));
                union { as16x4 h[2]; abf16x8 v; } u;
                u.h[0] = lo; u.h[1] = hi;
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, u.v, o[dt], 0, 0, 0);
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
    if (!v_ready) {                                              // a wave without tiles still owes the workgroup its V rows
        __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (0xF << 8));
        __builtin_amdgcn_s_barrier();
    }
}

hipError_t launch_attention(const AttnParams& p, int dtype, hipStream_t st) {
    const size_t qkv_bytes = (size_t)p.B * p.N * p.q_stride * 2;
    if (dtype == DT_BF16 && p.kd == 32 && p.hd == 64 && p.N <= 16 * A_NT && (p.q_stride & 7) == 0 && (p.q_coff & 7) == 0 && qkv_bytes < (1ull << 31)) {
        // query tiles per workgroup: as long as possible (K/V are staged once per workgroup) while the grid still covers the chip
        static const int target = [] { const char* s = getenv("YOLOP_ATTN_WGS"); const int v = s ? atoi(s) : 0; return v > 0 ? v : 256; }();
        const int ntiles = (p.N + 15) / 16, BH = p.B * p.nh;
        int nsplit = std::min(ntiles, std::max(1, (target + BH - 1) / BH));
        const int tpw = (ntiles + nsplit - 1) / nsplit;
        nsplit = (ntiles + tpw - 1) / tpw;
        hipLaunchKernelGGL(attention_mfma_kernel, dim3((unsigned)nsplit, (unsigned)BH), dim3(A_MAXW * 64), 0, st, p, tpw, (unsigned)qkv_bytes);
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
