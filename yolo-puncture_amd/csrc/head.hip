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

// The k-th largest of keys[0..n) (n >= k >= 1; keys unique): radix select from the top byte down; it stops as soon as the bin that holds the
// k-th key is wanted whole (then the threshold is the smallest key with that prefix) - in practice after the score bytes.
// `nflat`: exclusive bound of the flat indices in the keys' low words (0xFFFFFFFF - flat): the index bytes every key shares
// (0xFF above the bound's top bit) need no counting pass. Ends with a barrier.
__device__ unsigned long long radix_kth(const unsigned long long* keys, int n, int k, SelectShared& S, unsigned nflat) {
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
    return kth;
}

// keys[0..n) -> the k largest keys, sorted descending, in out512[0..k) (zero keys behind them). n >= k; keys are unique.
// The <= 512 survivors of the k-th-key threshold are ordered by rank counting (two threads per key, broadcast LDS reads) instead of a
// barrier-bound sorting network.
__device__ void select_topk_sorted(const unsigned long long* keys, int n, int k, unsigned long long* out512, unsigned long long* tmp512,
                                   SelectShared& S, unsigned nflat) {
    const int tid = threadIdx.x;
    // up to 512 keys need no threshold at all: rank every one of them and keep the ranks below k (the radix passes cost ~1.2 us each
    // whatever n is - three barriers and a one-wave scan - and stage 2 usually arrives here with little more than k candidates)
    int m = n;                                                  // keys that get ranked (they sit in tmp512[0..m))
    if (n > 512) {
        const unsigned long long kth = radix_kth(keys, n, k, S, nflat);
        if (tid == 0) S.count = 0;
        for (int i = tid; i < 512; i += HT) { out512[i] = 0ull; tmp512[i] = 0ull; }
        __syncthreads();
        for (int i = tid; i < n; i += HT) {
            const unsigned long long key = keys[i];
            if (key >= kth) tmp512[atomicAdd(&S.count, 1u)] = key;   // exactly k of them
        }
        m = k;
    } else {
        for (int i = tid; i < 512; i += HT) { out512[i] = 0ull; tmp512[i] = i < n ? keys[i] : 0ull; }
    }
    __syncthreads();
    {
        const int i = tid >> 1, half = tid & 1;
        const unsigned long long mykey = tmp512[i];
        unsigned rank = 0;
        const int hm = (m + 1) >> 1;                            // two threads per key, half of the m keys each
        const unsigned long long* q = tmp512 + half * hm;
        const int nq = half ? m - hm : hm;
        if (i < m)
            for (int j = 0; j < nq; ++j) rank += (q[j] > mykey) ? 1u : 0u;
        rank += __shfl_xor(rank, 1, 64);
        if (half == 0 && i < m && rank < (unsigned)k) out512[rank] = mykey;
    }
    __syncthreads();
}

// Append this lane's item to an LDS list with ONE atomic per wave: returns the lane's slot (valid where `have`).
__device__ __forceinline__ unsigned wave_append(bool have, unsigned* counter) {
    const unsigned long long m = __ballot(have);
    if (m == 0ull) return 0u;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)m) - 1;
    unsigned base = 0;
    if (lane == leader) base = atomicAdd(counter, (unsigned)__popcll(m));
    base = (unsigned)__shfl((int)base, leader, 64);
    return base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
}

// MODE 0: stage 1 + stage 2 + decode from the dense box / coefficient maps (one launch).
// MODE 1: stage 1 only - the winners go to global memory (anchor ids in rank order, per-level rank lists for head_branch.hip, the
//         stage-1 threshold); MODE 2: stage 2 + decode, reading the winners back and the box / coefficient rows the branch kernel made for
//         them (p.sp_box / p.sp_cf, indexed by rank) - the "winners-only" head.
template <int MODE>
__global__ __launch_bounds__(HT) void head_select_kernel(const HeadParams p, const unsigned* __restrict__ mkey) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long lds[];
    unsigned long long* keys = lds;                 // [CAP]
    unsigned long long* best = lds + CAP;           // [512] sorted result of the last select
    unsigned long long* carry = best + 512;         // [512] running best-k between stage-2 rounds
    unsigned long long* tmp = carry + 512;          // [512] unsorted survivors of a select
    unsigned long long* tmaxs = tmp + 512;          // [HT]  stage 1: the largest key of every thread
    int* sel = (int*)(tmaxs + HT);                  // [MAXK] stage-1 winners (anchor ids, rank order)
    // [MAXK] their class-logit rows, kept as GLOBAL-address-space pointers: through a generic pointer read back from LDS the gathers below
    // become flat_load, which counts in lgkmcnt as well - every wait for the next row pointer then drains the gathers in flight
    typedef const __attribute__((address_space(1))) float* gfptr;
    gfptr* selrow = (gfptr*)(sel + MAXK);
    // (kernel arguments indexed by a run-time level are re-read from the argument segment with a vector load + full wait per use)
    const unsigned* const mk0 = p.mk[0]; const unsigned* const mk1 = p.mk[1]; const unsigned* const mk2 = p.mk[2];
    const float* const cls0 = p.cls[0]; const float* const cls1 = p.cls[1]; const float* const cls2 = p.cls[2];
    const float* const box0 = p.box[0]; const float* const box1 = p.box[1]; const float* const box2 = p.box[2];
    const float* const cf0 = p.cf[0]; const float* const cf1 = p.cf[1]; const float* const cf2 = p.cf[2];
    const int w0 = p.hw[0][1], w1 = p.hw[1][1], w2 = p.hw[2][1];
    __shared__ SelectShared S;
    __shared__ unsigned nfill;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int A = p.A, k = min(p.max_det, A);
    const Locate locate{p.hw[0][0] * p.hw[0][1], p.hw[1][0] * p.hw[1][1], p.hw[2][0] * p.hw[2][1]};

    HEAD_STAMP(0);
    if (blockIdx.x == 0 && tid == 0) g_head_clk[7] = 0ull;
    unsigned thr_bits = 0u;
    if constexpr (MODE != 2) {
    // ---- stage 1: top-k anchors by (max score desc, anchor asc) ---------------------------------------------------------
    // Every thread keeps its <= CAP / HT keys in registers (anchor a = tid + i * HT: all loads in flight at once). The k-th largest of the
    // HT per-thread maxima is a LOWER bound T0 of the k-th largest key (k threads hold a key >= it), so the exact select only has to
    // look at the keys >= T0 - a few hundred instead of all 8400: one cheap select over HT keys + one over the survivors instead of
    // four counting passes over everything (17.6 -> see DESIGN us on the tail of the graph, where this kernel runs alone).
    constexpr int NPT = CAP / HT;
    unsigned long long kreg[NPT];
    unsigned long long tmx = 0ull;
    // (unconditional loads from clamped addresses: behind a per-key `if` the compiler waits for every load before it issues the next -
    //  nine dependent trips to memory were the 4.5 us this phase took)
    unsigned sbits[NPT];
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
        const int a = min(tid + i * HT, A - 1);
        if (mk0) {                                                // (uniform)
            int l, loc, HWl;
            locate(a, l, loc, HWl);
            sbits[i] = (l == 0 ? mk0 : l == 1 ? mk1 : mk2)[(size_t)b * HWl + loc];
        } else sbits[i] = mkey[(size_t)b * A + a];
    }
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
        const int a = tid + i * HT;
        kreg[i] = a < A ? (((unsigned long long)sbits[i] << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)a)) : 0ull;
    }
#pragma unroll
    for (int i = 0; i < NPT; ++i) tmx = kreg[i] > tmx ? kreg[i] : tmx;
    tmaxs[tid] = tmx;
    if (tid == 0) nfill = 0u;
    __syncthreads();
    HEAD_STAMP(1);
    const unsigned long long T0 = radix_kth(tmaxs, HT, k, S, (unsigned)A);
    HEAD_STAMP(2);
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
        const bool have = kreg[i] != 0ull && kreg[i] >= T0;
        const unsigned pos = wave_append(have, &nfill);
        if (have) keys[pos] = kreg[i];
    }
    __syncthreads();
    select_topk_sorted(keys, (int)nfill, k, best, tmp, S, (unsigned)A);
    HEAD_STAMP(3);
    thr_bits = (unsigned)(best[k - 1] >> 32);   // every selected anchor has a class with score >= this
    if constexpr (MODE == 1) {
        // hand the winners over: anchor ids by rank, and per level the ranks that lie on it (order inside a level's list is irrelevant)
        __shared__ unsigned s_wc[3], s_pc[3], s_pb[3];
        unsigned* const bits = (unsigned*)keys;                         // (the select is done with `keys`) one bit per anchor, levels word-aligned
        const int hw0 = locate.A0, hw1 = locate.A1, hw2 = locate.A2;
        const int wb1 = (hw0 + 31) >> 5, wb2 = wb1 + ((hw1 + 31) >> 5), nwt = wb2 + ((hw2 + 31) >> 5);
        if (tid < 3) { s_wc[tid] = 0u; s_pc[tid] = 0u; }
        if (p.sp_plist) for (int i = tid; i < nwt; i += HT) bits[i] = 0u;
        __syncthreads();
        for (int r = tid; r < k; r += HT) {
            const int a = (int)(0xFFFFFFFFu - (unsigned)(best[r] & 0xFFFFFFFFull));
            p.sp_sel[(size_t)b * HEAD_MAXK + r] = a;
            int l, loc, HWl;
            locate(a, l, loc, HWl);
            p.sp_wlist[((size_t)b * 3 + l) * HEAD_MAXK + atomicAdd(&s_wc[l], 1u)] = r | (loc << 9);     // rank (< 512) and level-local pixel in one word
            if (p.sp_plist) {
                // the positions whose first-convolution outputs this winner's second 3x3 reads: its in-frame 3x3 neighbourhood
                const int Wl = l == 0 ? w0 : l == 1 ? w1 : w2, Hl = HWl / Wl, wbl = l == 0 ? 0 : l == 1 ? wb1 : wb2;
                const int y = loc / Wl, x = loc - y * Wl;
                for (int dy = -1; dy <= 1; ++dy)
                    for (int dx = -1; dx <= 1; ++dx) {
                        const int yy = y + dy, xx = x + dx;
                        if ((unsigned)yy < (unsigned)Hl && (unsigned)xx < (unsigned)Wl) {
                            const int n = yy * Wl + xx;
                            atomicOr(&bits[wbl + (n >> 5)], 1u << (n & 31));
                        }
                    }
            }
        }
        __syncthreads();
        if (tid < 3) p.sp_wcount[b * 3 + tid] = (int)s_wc[tid];
        if (tid == 0) p.sp_thr[b] = thr_bits;
        if (p.sp_plist) {
            // distinct positions per level -> this image's share of the level's list (all images append to one list per level)
            for (int i = tid; i < nwt; i += HT) {
                const unsigned w = bits[i];
                if (w) atomicAdd(&s_pc[i >= wb2 ? 2 : i >= wb1 ? 1 : 0], (unsigned)__popc(w));
            }
            __syncthreads();
            if (tid < 3) { s_pb[tid] = (unsigned)atomicAdd(&p.sp_pcount[tid], (int)s_pc[tid]); s_pc[tid] = 0u; }
            __syncthreads();
            const int off0 = p.sp_plist_off[0], off1 = p.sp_plist_off[1], off2 = p.sp_plist_off[2];
            const int cap0 = p.sp_plist_cap[0], cap1 = p.sp_plist_cap[1], cap2 = p.sp_plist_cap[2];
            for (int i = tid; i < nwt; i += HT) {
                unsigned w = bits[i];
                if (!w) continue;
                const int l = i >= wb2 ? 2 : i >= wb1 ? 1 : 0;
                unsigned at = s_pb[l] + atomicAdd(&s_pc[l], (unsigned)__popc(w));
                const int wbl = l == 0 ? 0 : l == 1 ? wb1 : wb2, off = l == 0 ? off0 : l == 1 ? off1 : off2, cap = l == 0 ? cap0 : l == 1 ? cap1 : cap2;
                while (w) {
                    const int bit = __ffs((int)w) - 1;
                    w &= w - 1u;
                    if ((int)at < cap) p.sp_plist[off + at] = (b << 20) | (((i - wbl) << 5) + bit);
                    ++at;
                }
            }
        }
        HEAD_STAMP(6);
        return;
    }
    }   // MODE != 2
    else {
        thr_bits = p.sp_thr[b];
        for (int r = tid; r < k; r += HT) best[r] = ((unsigned long long)0u << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)p.sp_sel[(size_t)b * HEAD_MAXK + r]);
        __syncthreads();
        HEAD_STAMP(1); HEAD_STAMP(2); HEAD_STAMP(3);
    }
    for (int r = tid; r < k; r += HT) {
        const int a = (int)(0xFFFFFFFFu - (unsigned)(best[r] & 0xFFFFFFFFull));
        sel[r] = a;
        int l, loc, HWl;
        locate(a, l, loc, HWl);
        selrow[r] = (gfptr)((l == 0 ? cls0 : l == 1 ? cls1 : cls2) + ((size_t)b * HWl + loc) * p.nc);   // class-logit row of the r-th selected anchor
    }
    __syncthreads();

    // ---- stage 2: top-k of the k*nc (rank, class) candidates. A candidate below the stage-1 threshold can never be
    //      in the result (>= k candidates reach it), so only survivors enter LDS; rounds bound the LDS use exactly. ----
    // sigmoid is monotone, so a candidate can reach the stage-1 threshold only if its logit reaches logit(thr) - a margin that
    // covers the rounding of both evaluations (1e-3 in logit space moves a score by >= 2.5e-4 * s * (1 - s), far above 1 ulp
    // unless the score saturates; above 0.999 the filter is switched off). The scan only COMPARES logits and appends the survivors'
    // (logit, flat index) pairs, one LDS atomic per wave and step; their scores are evaluated afterwards on the dense list (one or two
    // per thread) - evaluating inside the scan ran the sigmoid + atomic path of nearly every one of the 24 unrolled steps for the few
    // lanes that needed it (15.4 us). Survivors below the exact threshold stay in the list: a superset selects the same top k.
    const float thr_f = __uint_as_float(thr_bits);
    const float lthr = (thr_f > 0.f && thr_f < 0.999f) ? (logf(thr_f / (1.0f - thr_f)) - 1e-3f) : -INFINITY;
    const int total = k * p.nc;
    int have = 0;                       // keys carried from earlier rounds (sorted, in carry[0..have))
    for (int done = 0; done < total;) {
        if (tid == 0) nfill = (unsigned)have;
        for (int i = tid; i < have; i += HT) keys[i] = carry[i];
        __syncthreads();
        // consume candidates until the buffer could overflow: stop when nfill + chunk > CAP
        int f0 = done;
        const bool vec = (p.nc & 3) == 0 && p.nc <= 256;           // whole rows as float4s (below); else candidate by candidate
        while (f0 < total) {
            const int before = (int)nfill;
            int chunk = min(total - f0, CAP - before);
            if (vec) chunk = (chunk / p.nc) * p.nc;                // (f0 is then always a row boundary)
            if (chunk <= 0) break;
            const int take = chunk;
            __syncthreads();                                         // (everyone has read nfill before anyone appends)
            if (vec) {
                // One CU scans k * nc candidates, so instructions per candidate are what this phase costs (80 of them per candidate in the
                // scalar form: 11 us). Here a lane takes one float4 of a row (nc / 4 lanes per row, 64 / (nc / 4) rows per wave instruction),
                // four compares, and the wave appends its survivors with ONE atomic per instruction.
                const int q = p.nc >> 2, rpw = 64 / q;
                const int lane = tid & 63, wave = tid >> 6;
                const int lr = lane / q, lc = (lane - lr * q) * 4;
                const int row0 = f0 / p.nc, row1 = row0 + take / p.nc;
                typedef float f32x4 __attribute__((ext_vector_type(4)));
                typedef const __attribute__((address_space(1))) f32x4* gf4ptr;
                constexpr int UR = 4;                              // wave instructions in flight
                for (int rb = row0 + wave * rpw; rb < row1; rb += UR * (HT / 64) * rpw) {
                    f32x4 v[UR];
#pragma unroll
                    for (int u = 0; u < UR; ++u) {
                        const int r = rb + u * (HT / 64) * rpw + lr;
                        v[u] = *(gf4ptr)(selrow[min(r, row1 - 1)] + lc);          // (unconditional: clamped row, value unused when out of range)
                    }
#pragma unroll
                    for (int u = 0; u < UR; ++u) {
                        const int r = rb + u * (HT / 64) * rpw + lr;
                        const bool on = lr < rpw && r < row1;
                        const float e[4] = {v[u][0], v[u][1], v[u][2], v[u][3]};
                        bool kp[4];
                        unsigned long long m[4];
                        unsigned cnt = 0;
#pragma unroll
                        for (int j = 0; j < 4; ++j) { kp[j] = on && e[j] >= lthr; m[j] = __ballot(kp[j]); cnt += (unsigned)__popcll(m[j]); }
                        if (cnt == 0u) continue;                   // (wave-uniform)
                        unsigned base = 0;
                        if (lane == 0) base = atomicAdd(&nfill, cnt);
                        base = (unsigned)__shfl((int)base, 0, 64);
                        const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if (kp[j]) keys[base + (unsigned)__popcll(m[j] & below)] =
                                ((unsigned long long)__float_as_uint(e[j]) << 32) | (unsigned long long)(unsigned)(r * p.nc + lc + j);
                            base += (unsigned)__popcll(m[j]);
                        }
                    }
                }
            } else {
                constexpr int U = 24;                                  // independent gathers in flight per thread
                const int qs = HT / p.nc, rs = HT - qs * p.nc;         // (r, c) of candidate f advance by (qs, rs) per HT candidates
                for (int i0 = tid; i0 < take; i0 += U * HT) {
                    float lg[U];
                    int r = (f0 + i0) / p.nc, c = (f0 + i0) - r * p.nc;
                    const int rlast = (f0 + take - 1) / p.nc;
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        // unconditional load (past the end: the last row's same column - a valid address, the value is not used): behind
                        // `if (i < take)` the 24 gathers went to memory one after the other
                        lg[u] = selrow[min(r, rlast)][c];
                        c += rs; r += qs;
                        if (c >= p.nc) { c -= p.nc; ++r; }
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int i = i0 + u * HT;
                        const bool keep = i < take && lg[u] >= lthr;
                        const unsigned pos = wave_append(keep, &nfill);
                        if (keep) keys[pos] = ((unsigned long long)__float_as_uint(lg[u]) << 32) | (unsigned long long)(unsigned)(f0 + i);
                    }
                }
            }
            __syncthreads();
            for (int j = before + tid; j < (int)nfill; j += HT) {   // the survivors' scores, on the dense list
                const unsigned long long e = keys[j];
                keys[j] = make_key(sigmoidf_(__uint_as_float((unsigned)(e >> 32))), (unsigned)(e & 0xFFFFFFFFull));
            }
            __syncthreads();
            f0 += take;
            if ((int)nfill + 1 >= CAP) break;
        }
        done = f0;
        HEAD_STAMP(4);
        const int n = (int)nfill;
        const int kk = min(k, n);
        if (blockIdx.x == 0 && tid == 0) g_head_clk[7] = (g_head_clk[7] & 0xffffffff00000000ull) + (1ull << 32) + (unsigned long long)n;   // [7] = rounds << 32 | keys of the last round
        select_topk_sorted(keys, n, kk, best, tmp, S, (unsigned)(A * p.nc));
        for (int i = tid; i < kk; i += HT) carry[i] = best[i];
        have = kk;
        __syncthreads();
    }

    HEAD_STAMP(5);
    // ---- winners: DFL decode (softmax expectation over 16 bins per side), dist2bbox (xyxy) * stride ----------------------
    // four threads per row, one per box side (16 loads + 16 expf each instead of 64 + 64 on a quarter of the threads); lane 0 of the quad
    // collects the distances and writes the row. Same arithmetic per side, so the same bits as one thread per row.
    for (int r0 = 0; r0 < p.max_det; r0 += HT / 4) {
        const int r = r0 + (tid >> 2), sd = tid & 3;
        const bool live = r < p.max_det && r < have;
        float dist = 0.f, score = 0.f;
        int a = -1, cls = 0, l = 0, loc = 0, HWl = 1, srow = 0;
        if (live) {
            const unsigned long long key = carry[r];
            score = __uint_as_float((unsigned)(key >> 32));
            const int f = (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull));
            const int row = f / p.nc;
            srow = row;
            cls = f - row * p.nc;
            a = sel[row];
            locate(a, l, loc, HWl);
            const float4* bp = MODE == 2 ? (const float4*)(p.sp_box + ((size_t)b * p.max_det + row) * 64 + sd * 16)
                                         : (const float4*)((l == 0 ? box0 : l == 1 ? box1 : box2) + ((size_t)b * HWl + loc) * 64 + sd * 16);
            const float4 q0 = bp[0], q1 = bp[1], q2 = bp[2], q3 = bp[3];
            float v[16] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
            float mx = -INFINITY;
#pragma unroll
            for (int i = 0; i < 16; ++i) mx = fmaxf(mx, v[i]);
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) { v[i] = expf(v[i] - mx); sum += v[i]; }
            float e = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) e += (v[i] / sum) * (float)i;
            dist = e;
        }
        const float d0 = __shfl(dist, (tid & 60) + 0, 64), d1 = __shfl(dist, (tid & 60) + 1, 64), d2 = __shfl(dist, (tid & 60) + 2, 64), d3 = __shfl(dist, (tid & 60) + 3, 64);
        if (r >= p.max_det) continue;
        float* d = p.det + ((size_t)b * p.max_det + r) * 6;
        if (r >= have) {
            if (sd == 0) {
#pragma unroll
                for (int j = 0; j < 6; ++j) d[j] = 0.f;
                if (p.idx) p.idx[(size_t)b * p.max_det + r] = -1;
            }
            if (p.coeff)
                for (int j = sd * 8; j < sd * 8 + 8; ++j) p.coeff[((size_t)b * p.max_det + r) * 32 + j] = 0.f;
            continue;
        }
        if (sd == 0) {
            const int Wl = l == 0 ? w0 : l == 1 ? w1 : w2;
            const int y = loc / Wl, x = loc - y * Wl;
            const float stride = (float)(8 << l);
            const float ax = (float)x + 0.5f, ay = (float)y + 0.5f;
            d[0] = (ax - d0) * stride;
            d[1] = (ay - d1) * stride;
            d[2] = (ax + d2) * stride;
            d[3] = (ay + d3) * stride;
            d[4] = score;
            d[5] = (float)cls;
            if (p.idx) p.idx[(size_t)b * p.max_det + r] = a;
        }
        if (p.coeff) {
            const float* cf = (MODE == 2 && p.sp_cf) ? p.sp_cf + ((size_t)b * p.max_det + srow) * 32 : (l == 0 ? cf0 : l == 1 ? cf1 : cf2) + ((size_t)b * HWl + loc) * 32;
            for (int j = sd * 8; j < sd * 8 + 8; ++j) p.coeff[((size_t)b * p.max_det + r) * 32 + j] = cf[j];
        }
    }
    if constexpr (MODE == 2) {
        // the position lists of this forward have been consumed (the branch kernels ran between the two stages): empty them for the next one
        if (b == 0 && tid < 3 && p.sp_pcount) { p.sp_pcount[4 + tid] = p.sp_pcount[tid]; p.sp_pcount[tid] = 0; }   // ([4..7): what yp_debug_head_positions reports)
    }
    HEAD_STAMP(6);
}

hipError_t head_read_clocks(unsigned long long* out8) { return hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_head_clk), 8 * sizeof(unsigned long long)); }

size_t head_scratch_bytes(int B, int A) { return (size_t)B * A * sizeof(unsigned); }

static hipError_t head_attrs() {
    const size_t sh = (size_t)(CAP + 1536 + HT) * 8 + MAXK * 4 + MAXK * 8;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)head_select_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)head_select_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)head_select_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    return hipSuccess;
}

hipError_t launch_head(const HeadParams& p, hipStream_t st) {
    if (p.A > CAP || p.max_det > MAXK || (p.scratch == nullptr && p.mk[0] == nullptr)) return hipErrorInvalidValue;
    const size_t sh = (size_t)(CAP + 1536 + HT) * 8 + MAXK * 4 + MAXK * 8;
    hipError_t e = head_attrs();
    if (e != hipSuccess) return e;
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
    hipLaunchKernelGGL(head_select_kernel<0>, dim3(p.B), dim3(HT), sh, st, p, mkey);
    return hipGetLastError();
}

// winners-only head: stage 1 (the per-level class-max keys must exist: p.mk), then the caller runs head_branch.hip, then stage 2 + decode
hipError_t launch_head_stage1(const HeadParams& p, hipStream_t st) {
    if (p.A > CAP || p.max_det > MAXK || p.mk[0] == nullptr || !p.sp_sel || !p.sp_wlist || !p.sp_wcount || !p.sp_thr) return hipErrorInvalidValue;
    const size_t sh = (size_t)(CAP + 1536 + HT) * 8 + MAXK * 4 + MAXK * 8;
    hipError_t e = head_attrs();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(head_select_kernel<1>, dim3(p.B), dim3(HT), sh, st, p, (const unsigned*)nullptr);
    return hipGetLastError();
}
hipError_t launch_head_stage2(const HeadParams& p, hipStream_t st) {
    if (p.A > CAP || p.max_det > MAXK || !p.sp_sel || !p.sp_thr || !p.sp_box) return hipErrorInvalidValue;
    const size_t sh = (size_t)(CAP + 1536 + HT) * 8 + MAXK * 4 + MAXK * 8;
    hipError_t e = head_attrs();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(head_select_kernel<2>, dim3(p.B), dim3(HT), sh, st, p, (const unsigned*)nullptr);
    return hipGetLastError();
}

}  // namespace yp
