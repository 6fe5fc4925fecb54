// Post-process of the anchor-free heads that still need NMS (YOLOv8 / YOLO11 Detect + Segment - the checkpoints the reference's UI
// offers, yolo_seg/app.py:218-223): what `ops.non_max_suppression` does inside `.predict` [U], one workgroup per image:
//   candidates = anchors whose best class score exceeds conf (the per-level class-max keys of OP_AMAX)
//   -> sorted by (score desc, anchor asc) with a bitonic network in LDS
//   -> per candidate: class = first arg-max of the sigmoid scores, box = DFL expectation -> dist2bbox -> xywh -> xyxy (the
//      round trip of Detect._inference + xywh2xyxy is part of the reference's arithmetic)
//   -> greedy NMS on class-offset boxes (box + cls * 7680, IoU = inter / (a_i + a_j - inter) in fp32, drop when > iou): the scan over
//      the sorted list is sequential, the suppression by each KEPT box is one parallel sweep + one barrier (at most max_det of them)
//   -> rows [x1,y1,x2,y2,score,cls], anchor index, mask coefficients; rows past the kept count are zero / -1.
// A box never suppresses a higher-scoring one, so the rows above any conf' >= conf are the rows NMS at conf' would give.
#include "common.h"

namespace yp {

constexpr int NT = 1024;
constexpr int NCAP = 16384;           // sorted-key capacity (power of two >= the 12288-anchor limit of the plan)
constexpr int NMAXK = 512;

__device__ __forceinline__ float nms_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

struct NmsLocate {
    int A0, A1, A2;
    __device__ __forceinline__ void operator()(int a, int& l, int& loc, int& HWl) const {
        if (a < A0) { l = 0; loc = a; HWl = A0; }
        else if (a < A0 + A1) { l = 1; loc = a - A0; HWl = A1; }
        else { l = 2; loc = a - A0 - A1; HWl = A2; }
    }
};

// Kernel A (whole chip): for every anchor whose best score exceeds conf - class = first arg-max of the sigmoid scores, box = DFL
// expectation -> dist2bbox -> xywh -> xyxy - written at the ANCHOR's slot of the scratch [B][A][8]. One anchor per thread.
__global__ __launch_bounds__(256) void head_nms_decode_kernel(const HeadParams p) {
    const NmsLocate locate{p.hw[0][0] * p.hw[0][1], p.hw[1][0] * p.hw[1][1], p.hw[2][0] * p.hw[2][1]};
    const long item = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (item >= (long)p.B * p.A) return;
    const int b = (int)(item / p.A), a = (int)(item - (long)b * p.A);
    int l, loc, HWl;
    locate(a, l, loc, HWl);
    const float conf = p.nms_params[0];
    if (!(p.mk[l][(size_t)b * HWl + loc] > __float_as_uint(fmaxf(conf, 0.f)))) return;
    const float* cp = p.cls[l] + ((size_t)b * HWl + loc) * p.nc;
    // `conf, j = cls.max(1)` takes the first maximum of the SIGMOID scores. sigmoid is monotone, so only classes whose logit is
    // within a hair of the largest logit - or, once that one saturates to 1.0f, any logit in the saturated range - can tie with it:
    // the sigmoid is evaluated for those alone
    float m = -INFINITY;
    for (int c = 0; c < p.nc; ++c) m = fmaxf(m, cp[c]);
    const float sm = nms_sigmoid(m);
    const float lo = (sm >= 1.0f) ? 15.0f : m - fmaxf(1e-3f, 1e-4f * fabsf(m));
    float best = -1.f;
    int cls = 0;
    for (int c = 0; c < p.nc; ++c) {
        const float v = cp[c];
        if (v >= lo) {
            const float s = nms_sigmoid(v);
            if (s > best) { best = s; cls = c; }
        }
    }
    const int Wl = p.hw[l][1];
    const int y = loc / Wl, x = loc - y * Wl;
    const float stride = (float)(8 << l);
    const float* bp = p.box[l] + ((size_t)b * HWl + loc) * 64;
    float dist[4];
#pragma unroll
    for (int sd = 0; sd < 4; ++sd) {
        float v[16], mx = -INFINITY;
#pragma unroll
        for (int q = 0; q < 16; ++q) { v[q] = bp[sd * 16 + q]; mx = fmaxf(mx, v[q]); }
        float sum = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) { v[q] = expf(v[q] - mx); sum += v[q]; }
        float e = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) e += (v[q] / sum) * (float)q;
        dist[sd] = e;
    }
    const float ax = (float)x + 0.5f, ay = (float)y + 0.5f;
    const float x1 = (ax - dist[0]) * stride, y1 = (ay - dist[1]) * stride, x2 = (ax + dist[2]) * stride, y2 = (ay + dist[3]) * stride;
    // dist2bbox(xywh=True) then xywh2xyxy
    const float cx = (x1 + x2) / 2.f, cy = (y1 + y2) / 2.f, w = x2 - x1, h = y2 - y1;
    float* o = p.nms_ws + ((size_t)b * p.A + a) * 8;
    o[0] = cx - w / 2.f; o[1] = cy - h / 2.f; o[2] = cx + w / 2.f; o[3] = cy + h / 2.f;
    o[4] = (float)cls;
}

// Kernel B (one workgroup per image): candidates -> sort -> greedy NMS -> rows
__global__ __launch_bounds__(NT) void head_nms_kernel(const HeadParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];     // [NCAP]
    __shared__ unsigned s_n;
    __shared__ unsigned s_dead[NCAP / 32];
    __shared__ int s_kept[NMAXK];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int A = p.A;
    const NmsLocate locate{p.hw[0][0] * p.hw[0][1], p.hw[1][0] * p.hw[1][1], p.hw[2][0] * p.hw[2][1]};
    const float conf = p.nms_params[0], iou_thr = p.nms_params[1];
    const unsigned conf_bits = __float_as_uint(fmaxf(conf, 0.f));
    if (tid == 0) s_n = 0;
    for (int i = tid; i < NCAP / 32; i += NT) s_dead[i] = 0u;
    __syncthreads();
    // ---- candidates: score > conf (scores are sigmoids, >= 0: bit patterns order like the floats) ------------------------------
    {
        int off = 0;
        for (int l = 0; l < 3; ++l) {
            const int HWl = p.hw[l][0] * p.hw[l][1];
            const unsigned* src = p.mk[l] + (size_t)b * HWl;
            for (int a = tid; a < HWl; a += NT) {
                const unsigned sb = src[a];
                if (sb > conf_bits) keys[atomicAdd(&s_n, 1u)] = ((unsigned long long)sb << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)(off + a));
            }
            off += HWl;
        }
    }
    __syncthreads();
    const int n = (int)s_n;
    int np2 = 2;
    while (np2 < n) np2 <<= 1;
    for (int i = n + tid; i < np2; i += NT) keys[i] = 0ull;
    __syncthreads();
    // ---- bitonic sort, descending ---------------------------------------------------------------------------------------------
    for (int k = 2; k <= np2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < np2; i += NT) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = keys[i], c = keys[ixj];
                    const bool up = (i & k) == 0;           // descending blocks first
                    if (up ? (a < c) : (a > c)) { keys[i] = c; keys[ixj] = a; }
                }
            }
            __syncthreads();
        }
    // ---- per candidate: class and box were computed per anchor by head_nms_decode_kernel; bring them into sorted order -----------
    const float* wsa = p.nms_ws + (size_t)b * A * 8;      // by anchor
    // the first `ncache` boxes of the sorted list live in the part of the key area the sort did not need (5 floats each): the sweep of
    // a kept box then costs LDS latency instead of dependent trips to L2
    float* cache = (float*)(keys + np2);
    const int ncache = min(n, (int)(((size_t)(NCAP - np2) * 8) / 20));
    for (int i = tid; i < ncache; i += NT) {
        const int a = (int)(0xFFFFFFFFu - (unsigned)(keys[i] & 0xFFFFFFFFull));
        const float* src = wsa + (size_t)a * 8;
        float* c5 = cache + (size_t)i * 5;
        c5[0] = src[0]; c5[1] = src[1]; c5[2] = src[2]; c5[3] = src[3]; c5[4] = src[4];
    }
    __syncthreads();
    auto box_of = [&](int i) -> const float* {
        if (i < ncache) return cache + (size_t)i * 5;
        return wsa + (size_t)(int)(0xFFFFFFFFu - (unsigned)(keys[i] & 0xFFFFFFFFull)) * 8;
    };
    // ---- greedy NMS ----------------------------------------------------------------------------------------------------------------
    const int kmax = min(p.max_det, NMAXK);
    int nk = 0;
    // walk the sorted list word by word: the next survivor is the lowest clear bit at or after the cursor (every thread reads the same
    // LDS word: a broadcast, the branches are uniform); only KEPT boxes cost a sweep and a barrier
    for (int w = 0; w * 32 < n && nk < kmax; ++w) {
        unsigned done = 0u;                                   // bits of this word already handled (kept) in this pass over it
        for (;;) {
            const int lim = min(32, n - w * 32);
            const unsigned valid = lim == 32 ? 0xffffffffu : ((1u << lim) - 1u);
            const unsigned alive = ~s_dead[w] & ~done & valid;
            if (!alive || nk >= kmax) break;
            const int bit = __builtin_ctz(alive);
            const int i = w * 32 + bit;
            done |= 1u << bit;
            if (tid == 0) s_kept[nk] = i;
            ++nk;
            const float* bi = box_of(i);
            const float ci = bi[4];
            const float off = ci * 7680.0f;
            const float ix1 = bi[0] + off, iy1 = bi[1] + off, ix2 = bi[2] + off, iy2 = bi[3] + off;
            const float iarea = (ix2 - ix1) * (iy2 - iy1);
            for (int j = i + 1 + tid; j < n; j += NT) {
                if ((s_dead[j >> 5] >> (j & 31)) & 1u) continue;
                const float* bj = box_of(j);
                if (bj[4] != ci) continue;                   // other classes sit 7680 px away: no intersection
                const float jx1 = bj[0] + off, jy1 = bj[1] + off, jx2 = bj[2] + off, jy2 = bj[3] + off;
                const float xx1 = fmaxf(ix1, jx1), yy1 = fmaxf(iy1, jy1), xx2 = fminf(ix2, jx2), yy2 = fminf(iy2, jy2);
                const float inter = fmaxf(xx2 - xx1, 0.f) * fmaxf(yy2 - yy1, 0.f);
                const float ovr = inter / (iarea + (jx2 - jx1) * (jy2 - jy1) - inter);
                if (ovr > iou_thr) atomicOr(&s_dead[j >> 5], 1u << (j & 31));
            }
            __syncthreads();
        }
    }
    __syncthreads();
    // ---- rows ------------------------------------------------------------------------------------------------------------------------
    for (int r = tid; r < p.max_det; r += NT) {
        float* d = p.det + ((size_t)b * p.max_det + r) * 6;
        if (r >= nk) {
#pragma unroll
            for (int j = 0; j < 6; ++j) d[j] = 0.f;
            if (p.idx) p.idx[(size_t)b * p.max_det + r] = -1;
            if (p.coeff)
                for (int j = 0; j < 32; ++j) p.coeff[((size_t)b * p.max_det + r) * 32 + j] = 0.f;
            continue;
        }
        const int i = s_kept[r];
        const unsigned long long key = keys[i];
        const int a = (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull));
        const float* bi = box_of(i);
        d[0] = bi[0]; d[1] = bi[1]; d[2] = bi[2]; d[3] = bi[3];
        d[4] = __uint_as_float((unsigned)(key >> 32));
        d[5] = bi[4];
        if (p.idx) p.idx[(size_t)b * p.max_det + r] = a;
        if (p.coeff) {
            int l, loc, HWl;
            locate(a, l, loc, HWl);
            const float* cf = p.cf[l] + ((size_t)b * HWl + loc) * 32;
            for (int j = 0; j < 32; ++j) p.coeff[((size_t)b * p.max_det + r) * 32 + j] = cf[j];
        }
    }
}

size_t head_nms_scratch_bytes(int B, int A) { return (size_t)B * A * 8 * sizeof(float); }

hipError_t launch_head_nms(const HeadParams& p, hipStream_t st) {
    if (p.A > 12288 || p.max_det > NMAXK || !p.mk[0] || !p.nms_params || !p.nms_ws) return hipErrorInvalidValue;
    const size_t sh = (size_t)NCAP * 8;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)head_nms_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const long items = (long)p.B * p.A;
    hipLaunchKernelGGL(head_nms_decode_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, p);
    hipLaunchKernelGGL(head_nms_kernel, dim3(p.B), dim3(NT), sh, st, p);
    return hipGetLastError();
}

}  // namespace yp
