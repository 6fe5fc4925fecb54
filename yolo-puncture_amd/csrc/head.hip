// v10Detect one-to-one head epilogue + NMS-free post-process, one workgroup per image, all in LDS:
//   sigmoid -> per-anchor class max -> top-k anchors -> top-k of (k x nc) -> DFL decode of the winners only.
// Replaces Detect._inference (DFL, dist2bbox, make_anchors) + v10postprocess inside `.predict`
// (reference yolo_seg/app.py:91); spec SURVEY.md A.4 / A.6 [U]. The [B,8400,4+nc] decoded tensor of the
// reference is never materialised. Ordering rule: score descending, ties by flat index ascending
// (stage 1: anchor index; stage 2: stage-1 rank * nc + class) - encoded in 64-bit keys so that one descending
// sort implements it exactly.
#include "common.h"

namespace yp {

constexpr int HT = 1024;        // threads
constexpr int NKEYS = 16384;    // LDS key capacity (128 KiB)
constexpr int MAXK = 1024;      // max top-k supported

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// descending bitonic sort of n (power of two) 64-bit keys in LDS by all HT threads
__device__ void bitonic_desc(unsigned long long* keys, int n) {
    for (int k2 = 2; k2 <= n; k2 <<= 1) {
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < (n >> 1); t += HT) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int ixj = i + j;
                const unsigned long long a = keys[i], b = keys[ixj];
                const bool desc = (i & k2) == 0;
                if ((a < b) == desc) { keys[i] = b; keys[ixj] = a; }
            }
            __syncthreads();
        }
    }
}

__device__ __forceinline__ int next_pow2(int v) {
    int n = 2;
    while (n < v) n <<= 1;
    return n;
}

__global__ __launch_bounds__(HT) void head_topk_kernel(const HeadParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];   // [NKEYS]
    int* sel = (int*)(keys + NKEYS);                                            // [MAXK] selected anchors (stage-1 order)
    const int b = blockIdx.x, tid = threadIdx.x;
    const int A = p.A;
    const int k = min(p.max_det, A);
    const int A0 = p.hw[0][0] * p.hw[0][1];
    const int A1 = (p.nlev > 1) ? p.hw[1][0] * p.hw[1][1] : 0;

    auto locate = [&](int a, int& l, int& loc, int& HWl) {
        if (a < A0) { l = 0; loc = a; HWl = A0; }
        else if (a < A0 + A1) { l = 1; loc = a - A0; HWl = A1; }
        else { l = 2; loc = a - A0 - A1; HWl = p.hw[2][0] * p.hw[2][1]; }
    };

    // ---- stage 1: m[a] = max_c sigmoid(cls[a][c]) = sigmoid(max_c logit) -------------------------------
    const int n1 = next_pow2(A);
    for (int a = tid; a < n1; a += HT) {
        unsigned long long key = 0ull;
        if (a < A) {
            int l, loc, HWl;
            locate(a, l, loc, HWl);
            const float* cp = p.cls[l] + ((size_t)b * HWl + loc) * p.nc;
            float mx = cp[0];
            for (int c = 1; c < p.nc; ++c) mx = fmaxf(mx, cp[c]);
            const float s = sigmoidf_(mx);
            key = ((unsigned long long)__float_as_uint(s) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)a);
        }
        keys[a] = key;
    }
    __syncthreads();
    bitonic_desc(keys, n1);
    for (int r = tid; r < k; r += HT) sel[r] = (int)(0xFFFFFFFFu - (unsigned)(keys[r] & 0xFFFFFFFFull));
    __syncthreads();

    // ---- stage 2: top-k of the k*nc (rank, class) candidates, in rounds that fit the LDS key buffer --------
    const int total = k * p.nc;
    for (int i = tid; i < k; i += HT) keys[i] = 0ull;   // running best-k (0 = empty, sorts last)
    __syncthreads();
    for (int done = 0; done < total;) {
        const int take = min(total - done, NKEYS - k);
        const int n2 = next_pow2(k + take);
        for (int i = tid; i < n2 - k; i += HT) {
            unsigned long long key = 0ull;
            if (i < take) {
                const int f = done + i;
                const int r = f / p.nc, c = f - r * p.nc;
                int l, loc, HWl;
                locate(sel[r], l, loc, HWl);
                const float s = sigmoidf_(p.cls[l][((size_t)b * HWl + loc) * p.nc + c]);
                key = ((unsigned long long)__float_as_uint(s) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)f);
            }
            keys[k + i] = key;
        }
        __syncthreads();
        bitonic_desc(keys, n2);
        done += take;
    }

    // ---- winners: DFL decode (softmax expectation over 16 bins per side), dist2bbox (xyxy) * stride ----------
    for (int r = tid; r < p.max_det; r += HT) {
        float* d = p.det + ((size_t)b * p.max_det + r) * 6;
        if (r >= k) {
#pragma unroll
            for (int j = 0; j < 6; ++j) d[j] = 0.f;
            if (p.idx) p.idx[(size_t)b * p.max_det + r] = -1;
            if (p.coeff)
                for (int j = 0; j < 32; ++j) p.coeff[((size_t)b * p.max_det + r) * 32 + j] = 0.f;
            continue;
        }
        const unsigned long long key = keys[r];
        const float score = __uint_as_float((unsigned)(key >> 32));
        const int f = (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull));
        const int row = f / p.nc, cls = f - row * p.nc;
        const int a = sel[row];
        int l, loc, HWl;
        locate(a, l, loc, HWl);
        const int Wl = p.hw[l][1];
        const int y = loc / Wl, x = loc - y * Wl;
        const float stride = (float)(8 << l);
        const float* bp = p.box[l] + ((size_t)b * HWl + loc) * 64;
        float dist[4];
#pragma unroll
        for (int sd = 0; sd < 4; ++sd) {
            float v[16], mx = -INFINITY;
#pragma unroll
            for (int i = 0; i < 16; ++i) { v[i] = bp[sd * 16 + i]; mx = fmaxf(mx, v[i]); }
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) { v[i] = expf(v[i] - mx); sum += v[i]; }
            float e = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) e += (v[i] / sum) * (float)i;
            dist[sd] = e;
        }
        const float ax = (float)x + 0.5f, ay = (float)y + 0.5f;
        d[0] = (ax - dist[0]) * stride;
        d[1] = (ay - dist[1]) * stride;
        d[2] = (ax + dist[2]) * stride;
        d[3] = (ay + dist[3]) * stride;
        d[4] = score;
        d[5] = (float)cls;
        if (p.idx) p.idx[(size_t)b * p.max_det + r] = a;
        if (p.coeff) {
            const float* cf = p.cf[l] + ((size_t)b * HWl + loc) * 32;
            for (int j = 0; j < 32; ++j) p.coeff[((size_t)b * p.max_det + r) * 32 + j] = cf[j];
        }
    }
}

hipError_t launch_head(const HeadParams& p, hipStream_t st) {
    if (p.A > NKEYS || p.max_det > MAXK || p.max_det * 2 > NKEYS) return hipErrorInvalidValue;
    const size_t sh = (size_t)NKEYS * 8 + MAXK * 4;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)head_topk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(head_topk_kernel, dim3(p.B), dim3(HT), sh, st, p);
    return hipGetLastError();
}

}  // namespace yp
