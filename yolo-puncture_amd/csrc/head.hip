// v10Detect one-to-one head epilogue + NMS-free post-process:
//   sigmoid -> per-anchor class max -> top-k anchors -> top-k of (k x nc) -> DFL decode of the winners only.
// Replaces Detect._inference (DFL, dist2bbox, make_anchors) + v10postprocess inside `.predict`
// (reference yolo_seg/app.py:91); spec SURVEY.md A.4 / A.6 [U]. The [B,8400,4+nc] decoded tensor of the
// reference is never materialised. Ordering rule: score descending, ties by flat index ascending
// (stage 1: anchor index; stage 2: stage-1 rank * nc + class) - encoded in unique 64-bit keys
//   key = float_bits(score) << 32 | (0xFFFFFFFF - flat_index)
// so "top-k by key" is exactly that rule. Two kernels:
//   anchor_max_kernel  (whole chip)   : coalesced pass over the class logits, one key per anchor
//   head_select_kernel (1 WG / image) : exact k-th-key radix select in LDS (8 x 8-bit passes over LDS histograms),
//                                       compaction, 512-key bitonic sort; stage 2 prefilters the k*nc candidates
//                                       with the stage-1 threshold and runs the same select in bounded rounds.
#include "common.h"

namespace yp {

constexpr int HT = 1024;         // threads of the select kernel
constexpr int CAP = 12288;       // LDS key capacity of the select kernel (96 KiB)
constexpr int MAXK = 512;        // max top-k supported (rank-sorted in one step)

// phase timestamps (s_memrealtime, 100 MHz) of image 0's workgroup in the last launch: tools read them through
// yp_debug_head_clocks to see where the select kernel's time goes
__device__ unsigned long long g_head_clk[8];
#define HEAD_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_head_clk[i] = wall_clock64(); } while (0)

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ unsigned long long make_key(float score, unsigned flat) {
    return ((unsigned long long)__float_as_uint(score) << 32) | (unsigned long long)(0xFFFFFFFFu - flat);
}

struct Locate {
    int A0, A1, A2;
    __device__ __forceinline__ void operator()(int a, int& l, int& loc, int& HWl) const {
        if (a < A0) { l = 0; loc = a; HWl = A0; }
        else if (a < A0 + A1) { l = 1; loc = a - A0; HWl = A1; }
        else { l = 2; loc = a - A0 - A1; HWl = A2; }
    }
};

// ---------------------------------------------------------------------------------------------------------------
// kernel 1: m[a] = sigmoid(max_c logit[a][c]); 16 lanes per anchor, coalesced 64-B reads
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void anchor_max_kernel(const HeadParams p, unsigned* __restrict__ mkey) {
    const Locate locate{p.hw[0][0] * p.hw[0][1], p.hw[1][0] * p.hw[1][1], p.hw[2][0] * p.hw[2][1]};
    const int sub = threadIdx.x & 15;
    const long item = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 4;   // (image, anchor)
    const long total = (long)p.B * p.A;
    float mx = -INFINITY;
    if (item < total) {
        const int b = (int)(item / p.A), a = (int)(item - (long)b * p.A);
        int l, loc, HWl;
        locate(a, l, loc, HWl);
        const float* cp = p.cls[l] + ((size_t)b * HWl + loc) * p.nc;
        for (int c = sub; c < p.nc; c += 16) mx = fmaxf(mx, cp[c]);
    }
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 16));
    if (sub == 0 && item < total) mkey[item] = __float_as_uint(sigmoidf_(mx));
}

// nc % 4 == 0 form: a workgroup owns 64 consecutive anchors of one level of one image = 16*nc consecutive float4s, read
// with full-width coalesced loads; per-float4 maxima go through LDS and one thread per anchor finishes the row.
__global__ __launch_bounds__(256) void anchor_max4_kernel(const HeadParams p, unsigned* __restrict__ mkey, const int blocks_per_image,
                                                          const int nb0, const int nb1) {
    __shared__ float part[64 * 64];                    // [64 anchors][nc/4 <= 64]
    const int b = blockIdx.x / blocks_per_image, blk = blockIdx.x - b * blocks_per_image;
    const int A0 = p.hw[0][0] * p.hw[0][1], A1 = p.hw[1][0] * p.hw[1][1], A2 = p.hw[2][0] * p.hw[2][1];
    int l, a0, HWl, abase;
    if (blk < nb0) { l = 0; a0 = blk * 64; HWl = A0; abase = 0; }
    else if (blk < nb0 + nb1) { l = 1; a0 = (blk - nb0) * 64; HWl = A1; abase = A0; }
    else { l = 2; a0 = (blk - nb0 - nb1) * 64; HWl = A2; abase = A0 + A1; }
    const int na = min(64, HWl - a0);
    const int q = p.nc >> 2;                            // float4s per anchor
    const float4* src = (const float4*)(p.cls[l] + ((size_t)b * HWl + a0) * p.nc);
    const int n4 = na * q;
    for (int i = threadIdx.x; i < n4; i += 256) {
        const float4 v = src[i];
        part[i] = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
    }
    __syncthreads();
    // 4 threads per anchor
    const int a = threadIdx.x >> 2, sub = threadIdx.x & 3;
    float mx = -INFINITY;
    if (a < na)
        for (int j = sub; j < q; j += 4) mx = fmaxf(mx, part[a * q + j]);
    mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
    if (a < na && sub == 0) mkey[(size_t)b * p.A + abase + a0 + a] = __float_as_uint(sigmoidf_(mx));
}

// one level, one launch (OP_AMAX): the class-max pass of a level runs on that level's class lane as soon as its logits exist,
// so only the select kernel is left on the tail of the graph. Same arithmetic as anchor_max4_kernel.
__global__ __launch_bounds__(256) void anchor_max_level_kernel(const float* __restrict__ cls, const int B, const int HW, const int nc,
                                                               unsigned* __restrict__ out) {
    __shared__ float part[64 * 64];
    const int bpi = (HW + 63) / 64;
    const int b = blockIdx.x / bpi, a0 = (blockIdx.x - b * bpi) * 64;
    const int na = min(64, HW - a0);
    const int a = threadIdx.x >> 2, sub = threadIdx.x & 3;
    float mx = -INFINITY;
    if ((nc & 3) == 0 && nc <= 256) {
        const int q = nc >> 2;
        const float4* src = (const float4*)(cls + ((size_t)b * HW + a0) * nc);
        const int n4 = na * q;
        for (int i = threadIdx.x; i < n4; i += 256) {
            const float4 v = src[i];
            part[i] = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
        }
        __syncthreads();
        if (a < na)
            for (int j = sub; j < q; j += 4) mx = fmaxf(mx, part[a * q + j]);
    } else if (a < na) {
        const float* cp = cls + ((size_t)b * HW + a0 + a) * nc;
        for (int c = sub; c < nc; c += 4) mx = fmaxf(mx, cp[c]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
    if (a < na && sub == 0) out[(size_t)b * HW + a0 + a] = __float_as_uint(sigmoidf_(mx));
}

hipError_t launch_anchor_max_level(const float* cls, int B, int HW, int nc, unsigned* out, hipStream_t st) {
    hipLaunchKernelGGL(anchor_max_level_kernel, dim3((unsigned)(B * ((HW + 63) / 64))), dim3(256), 0, st, cls, B, HW, nc, out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// LDS helpers of kernel 2 (all HT threads participate)
// ---------------------------------------------------------------------------------------------------------------
struct SelectShared {
    unsigned hist[256];
    unsigned long long prefix;
    unsigned want;
    unsigned count;
};

// keys[0..n) -> the k largest keys, sorted descending, in out512[0..k) (zero keys behind them). n >= k; keys are unique.
// Radix select from the top byte down; it stops as soon as the bin that holds the k-th key is wanted whole (then the
// threshold is the smallest key with that prefix) - in practice after the score bytes. The <= 512 survivors are ordered
// by rank counting (two threads per key, broadcast LDS reads) instead of a barrier-bound sorting network.
// `nflat`: exclusive bound of the flat indices in the keys' low words (0xFFFFFFFF - flat): the index bytes every key shares
// (0xFF above the bound's top bit) need no counting pass.
__device__ void select_topk_sorted(const unsigned long long* keys, int n, int k, unsigned long long* out512, unsigned long long* tmp512,
                                   SelectShared& S, unsigned nflat) {
    const int tid = threadIdx.x;
    if (tid == 0) { S.prefix = 0ull; S.want = (unsigned)k; S.count = 0xFFFFFFFFu; }
    __syncthreads();
    unsigned long long kth = 0ull;
    for (int pass = 0; pass < 8; ++pass) {
        const int shift = 56 - 8 * pass;
        if (pass >= 4 && ((nflat - 1u) >> shift) == 0u) {      // every key has 0xFF here (block-uniform branch)
            __syncthreads();                                    // (everyone has taken its copy of the previous prefix)
            kth = (kth << 8) | 0xFFull;
            if (tid == 0) S.prefix = kth;
            continue;
        }
        if (tid < 256) S.hist[tid] = 0;
        __syncthreads();
        const unsigned long long pre = S.prefix;
        for (int i = tid; i < n; i += HT) {
            const unsigned long long key = keys[i];
            const unsigned bin = (unsigned)(key >> shift) & 255u;
            // wave-aggregated counting: scores crowd into a few exponent bins (and tie-heavy inputs into one bin per pass), where
            // per-lane LDS atomics serialise; up to 4 rounds peel the bin of the first pending lane, stragglers go one by one
            unsigned long long pend = __ballot(pass == 0 || (key >> (shift + 8)) == pre);
            for (int r = 0; r < 4 && pend; ++r) {
                const int leader = __ffsll((long long)pend) - 1;
                const unsigned lb = (unsigned)__shfl((int)bin, leader, 64);
                const unsigned long long same = __ballot(bin == lb) & pend;
                if ((int)(threadIdx.x & 63) == leader) atomicAdd(&S.hist[lb], (unsigned)__popcll(same));
                pend &= ~same;
            }
            if ((pend >> (threadIdx.x & 63)) & 1ull) atomicAdd(&S.hist[bin], 1u);
        }
        __syncthreads();
        if (tid < 64) {   // one wave: find the bin holding the want-th largest key among the keys that match the prefix
            unsigned c0 = S.hist[4 * tid], c1 = S.hist[4 * tid + 1], c2 = S.hist[4 * tid + 2], c3 = S.hist[4 * tid + 3];
            unsigned mine = c0 + c1 + c2 + c3;
            unsigned run = mine;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned t = __shfl_down(run, o, 64);
                if (tid + o < 64) run += t;
            }
            const unsigned above = run - mine;   // keys in bins of higher lanes
            const unsigned want = S.want;
            if (above < want && want <= above + mine) {      // the target bin is in this (unique) lane
                unsigned acc = above, cb;
                int bin;
                if (want <= acc + c3) { bin = 3; cb = c3; }
                else { acc += c3; if (want <= acc + c2) { bin = 2; cb = c2; } else { acc += c2; if (want <= acc + c1) { bin = 1; cb = c1; } else { acc += c1; bin = 0; cb = c0; } } }
                S.prefix = (pre << 8) | (unsigned long long)(4 * tid + bin);
                S.want = want - acc;
                S.count = (want - acc == cb) ? (unsigned)shift : 0xFFFFFFFFu;   // whole bin wanted -> done
            }
        }
        __syncthreads();
        if (S.count != 0xFFFFFFFFu) { kth = S.prefix << S.count; break; }
        kth = S.prefix;
    }
    __syncthreads();
    if (tid == 0) S.count = 0;
    for (int i = tid; i < 512; i += HT) { out512[i] = 0ull; tmp512[i] = 0ull; }
    __syncthreads();
    for (int i = tid; i < n; i += HT) {
        const unsigned long long key = keys[i];
        if (key >= kth) tmp512[atomicAdd(&S.count, 1u)] = key;   // exactly k of them
    }
    __syncthreads();
    {
        const int i = tid >> 1, half = tid & 1;
        const unsigned long long mykey = tmp512[i];
        unsigned rank = 0;
        const unsigned long long* q = tmp512 + half * 256;
#pragma unroll 8
        for (int j = 0; j < 256; ++j) rank += (q[j] > mykey) ? 1u : 0u;
        rank += __shfl_xor(rank, 1, 64);
        if (half == 0 && i < k) out512[rank] = mykey;
    }
    __syncthreads();
}

__global__ __launch_bounds__(HT) void head_select_kernel(const HeadParams p, const unsigned* __restrict__ mkey) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long lds[];
    unsigned long long* keys = lds;                 // [CAP]
    unsigned long long* best = lds + CAP;           // [512] sorted result of the last select
    unsigned long long* carry = best + 512;         // [512] running best-k between stage-2 rounds
    unsigned long long* tmp = carry + 512;          // [512] unsorted survivors of a select
    int* sel = (int*)(tmp + 512);                   // [MAXK] stage-1 winners (anchor ids, rank order)
    const float** selrow = (const float**)(sel + MAXK);   // [MAXK] their class-logit rows
    __shared__ SelectShared S;
    __shared__ unsigned nfill;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int A = p.A, k = min(p.max_det, A);
    const Locate locate{p.hw[0][0] * p.hw[0][1], p.hw[1][0] * p.hw[1][1], p.hw[2][0] * p.hw[2][1]};

    HEAD_STAMP(0);
    // ---- stage 1: top-k anchors by (max score desc, anchor asc) ---------------------------------------------------------
    if (p.mk[0]) {
        int off = 0;
        for (int l = 0; l < 3; ++l) {                      // level by level: coalesced, no per-key level search
            const int HWl = p.hw[l][0] * p.hw[l][1];
            const unsigned* src = p.mk[l] + (size_t)b * HWl;
            for (int a = tid; a < HWl; a += HT)
                keys[off + a] = ((unsigned long long)src[a] << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)(off + a));
            off += HWl;
        }
    } else {
        for (int a = tid; a < A; a += HT)
            keys[a] = ((unsigned long long)mkey[(size_t)b * A + a] << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)a);
    }
    __syncthreads();
    HEAD_STAMP(1);
    select_topk_sorted(keys, A, k, best, tmp, S, (unsigned)A);
    HEAD_STAMP(2);
    for (int r = tid; r < k; r += HT) {
        const int a = (int)(0xFFFFFFFFu - (unsigned)(best[r] & 0xFFFFFFFFull));
        sel[r] = a;
        int l, loc, HWl;
        locate(a, l, loc, HWl);
        selrow[r] = p.cls[l] + ((size_t)b * HWl + loc) * p.nc;           // class-logit row of the r-th selected anchor
    }
    const unsigned thr_bits = (unsigned)(best[k - 1] >> 32);   // every selected anchor has a class with score >= this
    __syncthreads();

    // ---- stage 2: top-k of the k*nc (rank, class) candidates. A candidate below the stage-1 threshold can never be
    //      in the result (>= k candidates reach it), so only survivors enter LDS; rounds bound the LDS use exactly. ----
    // sigmoid is monotone, so a candidate can reach the stage-1 threshold only if its logit reaches logit(thr) - a margin that
    // covers the rounding of both evaluations (1e-3 in logit space moves a score by >= 2.5e-4 * s * (1 - s), far above 1 ulp
    // unless the score saturates; above 0.999 the filter is switched off)
    const float thr_f = __uint_as_float(thr_bits);
    const float lthr = (thr_f > 0.f && thr_f < 0.999f) ? (logf(thr_f / (1.0f - thr_f)) - 1e-3f) : -INFINITY;
    const int total = k * p.nc;
    int have = 0;                       // keys carried from earlier rounds (sorted, in carry[0..have))
    for (int done = 0; done < total;) {
        if (tid == 0) nfill = (unsigned)have;
        for (int i = tid; i < have; i += HT) keys[i] = carry[i];
        __syncthreads();
        // consume candidates until the buffer could overflow: each pass takes HT*? candidates; stop when nfill + chunk > CAP
        int f0 = done;
        while (f0 < total) {
            const int chunk = min(total - f0, CAP - (int)nfill);
            if (chunk <= 0) break;
            const int take = min(chunk, total - f0);
            constexpr int U = 24;                                  // independent gathers in flight per thread
            const int qs = HT / p.nc, rs = HT - qs * p.nc;         // (r, c) of candidate f advance by (qs, rs) per HT candidates
            for (int i0 = tid; i0 < take; i0 += U * HT) {
                float lg[U];
                int r = (f0 + i0) / p.nc, c = (f0 + i0) - r * p.nc;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int i = i0 + u * HT;
                    lg[u] = -INFINITY;
                    if (i < take) lg[u] = selrow[r][c];
                    c += rs; r += qs;
                    if (c >= p.nc) { c -= p.nc; ++r; }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int i = i0 + u * HT;
                    if (i < take && lg[u] >= lthr) {                  // (cheap necessary condition first: most candidates stop here)
                        const float s = sigmoidf_(lg[u]);
                        if (__float_as_uint(s) >= thr_bits) keys[atomicAdd(&nfill, 1u)] = make_key(s, (unsigned)(f0 + i));
                    }
                }
            }
            __syncthreads();
            f0 += take;
            if ((int)nfill + 1 >= CAP) break;
        }
        done = f0;
        HEAD_STAMP(3);
        const int n = (int)nfill;
        const int kk = min(k, n);
        select_topk_sorted(keys, n, kk, best, tmp, S, (unsigned)(A * p.nc));
        for (int i = tid; i < kk; i += HT) carry[i] = best[i];
        have = kk;
        __syncthreads();
    }

    HEAD_STAMP(4);
    // ---- winners: DFL decode (softmax expectation over 16 bins per side), dist2bbox (xyxy) * stride ----------------------
    for (int r = tid; r < p.max_det; r += HT) {
        float* d = p.det + ((size_t)b * p.max_det + r) * 6;
        if (r >= have) {
#pragma unroll
            for (int j = 0; j < 6; ++j) d[j] = 0.f;
            if (p.idx) p.idx[(size_t)b * p.max_det + r] = -1;
            if (p.coeff)
                for (int j = 0; j < 32; ++j) p.coeff[((size_t)b * p.max_det + r) * 32 + j] = 0.f;
            continue;
        }
        const unsigned long long key = carry[r];
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
    HEAD_STAMP(5);
}

hipError_t head_read_clocks(unsigned long long* out8) { return hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_head_clk), 8 * sizeof(unsigned long long)); }

size_t head_scratch_bytes(int B, int A) { return (size_t)B * A * sizeof(unsigned); }

hipError_t launch_head(const HeadParams& p, hipStream_t st) {
    if (p.A > CAP || p.max_det > MAXK || (p.scratch == nullptr && p.mk[0] == nullptr)) return hipErrorInvalidValue;
    const size_t sh = (size_t)(CAP + 1536) * 8 + MAXK * 4 + MAXK * 8;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)head_select_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    unsigned* mkey = (unsigned*)p.scratch;
    if (p.mk[0]) {
        // class-max keys were produced per level by OP_AMAX
    } else if ((p.nc & 3) == 0 && p.nc <= 256) {
        const int nb0 = (p.hw[0][0] * p.hw[0][1] + 63) / 64, nb1 = (p.hw[1][0] * p.hw[1][1] + 63) / 64, nb2 = (p.hw[2][0] * p.hw[2][1] + 63) / 64;
        const int bpi = nb0 + nb1 + nb2;
        hipLaunchKernelGGL(anchor_max4_kernel, dim3((unsigned)(p.B * bpi)), dim3(256), 0, st, p, mkey, bpi, nb0, nb1);
    } else {
        const long items = (long)p.B * p.A * 16;
        hipLaunchKernelGGL(anchor_max_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, p, mkey);
    }
    hipLaunchKernelGGL(head_select_kernel, dim3(p.B), dim3(HT), sh, st, p, mkey);
    return hipGetLastError();
}

}  // namespace yp
