// Segmentation tail on the GPU: mask = coeff[n,32] x proto[32,Hp*Wp]  ->  bilinear resize  ->  box crop  ->  > 0
// -> (optional) the id painting of `auto_segment`.
// Replaces ultralytics ops.process_mask_native (retina_masks=True, reference yolo_seg/app.py:49,91 and
// yolo_seg/yolo_with_deva.py:51) / ops.process_mask (dev_tools/auto_speed_calc.py:62) [U: SURVEY.md A.7] and the
// per-mask Python loop yolo_seg/yolo_with_deva.py:62-86 (`output_mask[mask > 0.5] = curr_id`, later ids overwrite
// earlier ones, masks with area < MIN_AREA_THRESHOLD skipped, ids consecutive over kept masks).
// Order of arithmetic follows the reference: fp32 GEMM at prototype resolution first, then 4-tap interpolation
// (PyTorch upsample_bilinear2d, align_corners=False: src = scale*(dst+0.5)-0.5 clamped at 0).
#include "common.h"
#include <cstdlib>

namespace yp {

template <typename T> __device__ __forceinline__ void load32(const T* p, float* f);
template <> __device__ __forceinline__ void load32<__bf16>(const __bf16* p, float* f) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint4 r = ((const uint4*)p)[q];
        const uint32_t u[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[q * 8 + 2 * i] = __uint_as_float(u[i] << 16);
            f[q * 8 + 2 * i + 1] = __uint_as_float(u[i] & 0xffff0000u);
        }
    }
}
template <> __device__ __forceinline__ void load32<float>(const float* p, float* f) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float4 r = ((const float4*)p)[q];
        f[q * 4] = r.x; f[q * 4 + 1] = r.y; f[q * 4 + 2] = r.z; f[q * 4 + 3] = r.w;
    }
}

// M[i][y][x] = sum_c coeff[i][c] * proto[t+y][l+x][c]   over the crop rectangle (ch x cw)
template <typename T>
__global__ __launch_bounds__(256) void mask_gemm_kernel(const MaskParams p, float* M) {
    extern __shared__ float cs[];   // [n][32]
    for (int i = threadIdx.x; i < p.n * 32; i += blockDim.x) cs[i] = p.coeff[i];
    __syncthreads();
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= p.ch * p.cw) return;
    const int y = pix / p.cw, x = pix - y * p.cw;
    float pv[32];
    load32<T>((const T*)p.proto + ((size_t)(p.t + y) * p.Wp + (p.l + x)) * 32, pv);
    for (int i = 0; i < p.n; ++i) {
        const float* c = cs + i * 32;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) acc = fmaf(c[k], pv[k], acc);
        M[(size_t)i * p.ch * p.cw + pix] = acc;
    }
}

// The same product on the matrix cores (the "batched MFMA GEMM" of the prototype tail): D[mask][pixel] = coeff[mask][k] *
// proto[pixel][k] with v_mfma_f32_16x16x4_f32 - exact fp32 FMAs accumulated in k order, i.e. the very chain
// fmaf(c[k], p[k], acc) of the scalar kernel above, so the two give the same bits (tests compare the tail with the oracle
// either way). One wave = 16 pixels x all masks, 16 masks per accumulator; coefficients are read from LDS as the A operand.
typedef __attribute__((ext_vector_type(4))) float mf32x4;
template <typename T>
__global__ __launch_bounds__(256) void mask_gemm_mfma_kernel(const MaskParams p, float* M) {
    extern __shared__ float cs[];   // [ceil16(n)][32], rows >= n zero
    const int npad = (p.n + 15) & ~15;
    for (int i = threadIdx.x; i < npad * 32; i += blockDim.x) cs[i] = (i < p.n * 32) ? p.coeff[i] : 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fk = lane >> 4;
    const int npix = p.ch * p.cw;
    const int pix0 = (blockIdx.x * 4 + wave) * 16;
    if (pix0 >= npix) return;
    const int pix = min(pix0 + fr, npix - 1);                     // padding lanes recompute the last pixel, never stored
    const int y = pix / p.cw, x = pix - y * p.cw;
    const T* pp = (const T*)p.proto + ((size_t)(p.t + y) * p.Wp + (p.l + x)) * 32;
    float b[8];                                                   // B operand: proto[pixel fr][k = 4j + fk]
#pragma unroll
    for (int j = 0; j < 8; ++j) b[j] = (float)pp[4 * j + fk];
    for (int m0 = 0; m0 < npad; m0 += 16) {
        mf32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(cs[(m0 + fr) * 32 + 4 * j + fk], b[j], acc, 0, 0, 0);
        if (pix0 + fr < npix) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = m0 + fk * 4 + i;                    // D rows: mask index
                if (m < p.n) M[(size_t)m * npix + pix0 + fr] = acc[i];
            }
        }
    }
}

__global__ __launch_bounds__(256) void mask_resize_kernel(const MaskParams p, const float* M, uint8_t* masks, int32_t* area) {
    const int i = blockIdx.y;
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    int on = 0;
    if (pix < p.oh * p.ow) {
        const int y = pix / p.ow, x = pix - y * p.ow;
        const float x1 = p.boxes[i * 4 + 0], y1 = p.boxes[i * 4 + 1], x2 = p.boxes[i * 4 + 2], y2 = p.boxes[i * 4 + 3];
        const float* Mi = M + (size_t)i * p.ch * p.cw;
        bool inside = true;
        if (!p.crop_before) inside = ((float)x >= x1) && ((float)x < x2) && ((float)y >= y1) && ((float)y < y2);
        if (inside) {
            const float sy = (float)p.ch / (float)p.oh, sx = (float)p.cw / (float)p.ow;
            float fy = sy * ((float)y + 0.5f) - 0.5f, fx = sx * ((float)x + 0.5f) - 0.5f;
            fy = fmaxf(fy, 0.f);
            fx = fmaxf(fx, 0.f);
            const int y0 = (int)fy, x0 = (int)fx;
            const int y1i = y0 + ((y0 < p.ch - 1) ? 1 : 0), x1i = x0 + ((x0 < p.cw - 1) ? 1 : 0);
            const float ly1 = fy - (float)y0, lx1 = fx - (float)x0, ly0 = 1.f - ly1, lx0 = 1.f - lx1;
            float v00 = Mi[y0 * p.cw + x0], v01 = Mi[y0 * p.cw + x1i], v10 = Mi[y1i * p.cw + x0], v11 = Mi[y1i * p.cw + x1i];
            if (p.crop_before) {
                // process_mask: zero outside the box scaled to prototype resolution, BEFORE interpolation
                const float bx1 = x1 * p.bsx, bx2 = x2 * p.bsx, by1 = y1 * p.bsy, by2 = y2 * p.bsy;
                auto in = [&](int yy, int xx) { return ((float)xx >= bx1) && ((float)xx < bx2) && ((float)yy >= by1) && ((float)yy < by2); };
                if (!in(y0, x0)) v00 = 0.f;
                if (!in(y0, x1i)) v01 = 0.f;
                if (!in(y1i, x0)) v10 = 0.f;
                if (!in(y1i, x1i)) v11 = 0.f;
            }
            const float v = ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11);
            on = v > 0.f ? 1 : 0;
        }
        masks[(size_t)i * p.oh * p.ow + pix] = (uint8_t)on;
    }
    // mask.sum() per row (yolo_with_deva.py:75)
    const unsigned long long bal = __ballot(on);
    if ((threadIdx.x & 63) == 0 && bal) atomicAdd(&area[i], __popcll(bal));
}

__global__ void mask_ids_kernel(const MaskParams p, const int32_t* area, int32_t* kept) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int cur = 1;
    for (int i = 0; i < p.n; ++i) {
        if (p.suppress_small && area[i] < p.min_area) kept[i] = 0;
        else kept[i] = cur++;
    }
}

__global__ __launch_bounds__(256) void mask_paint_kernel(const MaskParams p, const uint8_t* masks, const int32_t* kept, int64_t* ids) {
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= p.oh * p.ow) return;
    int64_t id = 0;
    for (int i = p.n - 1; i >= 0; --i) {   // the last painter wins
        if (kept[i] > 0 && masks[(size_t)i * p.oh * p.ow + pix]) { id = kept[i]; break; }
    }
    ids[pix] = id;
}

// ---------------------------------------------------------------------------------------------------------------
// Second resize of `auto_segment` when the frame was shrunk before predict (min_side > 0, yolo_seg/yolo_with_deva.py:45-48):
// every float {0,1} mask at (oh,ow) is resized to the original frame (rh,rw) by torchvision `F.resize` (:71-72) = bilinear,
// align_corners=False, ANTIALIAS on: a separable triangle filter, horizontal pass first, weights in fp32. The arithmetic below is
// torch's CPU kernel restated operation by operation (ATen UpSampleKernel: _compute_indices_min_size_weights_aa + the two basic
// loops), including the places where C++ promotes to double and the fused multiply-adds of the accumulation, because the decision
// `mask > 0.5` (:79) meets exact ties (a 2:3 upscale puts thousands of pixels at exactly 0.5): bit-equal floats or different ids.
// ---------------------------------------------------------------------------------------------------------------
constexpr int AA_KMAX = 36;
__global__ void aa_weights_kernel(int n_in, int n_out, int* xmin, int* xsize, float* W) {
#pragma clang fp contract(off)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_out) return;
    const float scale = (float)n_in / (float)n_out;
    const float support = scale >= 1.f ? scale : 1.f;
    const float invscale = scale >= 1.f ? 1.f / scale : 1.f;
    const int K = (int)ceilf(support) * 2 + 1;
    const float center = (float)((double)scale * ((double)i + 0.5));
    int lo = (int)((double)(center - support) + 0.5);
    lo = max(lo, 0);
    int hi = (int)((double)(center + support) + 0.5);
    hi = min(hi, n_in);
    const int sz = min(max(hi - lo, 0), min(K, AA_KMAX));
    float tot = 0.f;
    float* w = W + (size_t)i * AA_KMAX;
    for (int j = 0; j < sz; ++j) {
        float x = (float)(((double)((float)(j + lo) - center) + 0.5) * (double)invscale);
        x = fabsf(x);
        const float v = x < 1.f ? 1.f - x : 0.f;
        w[j] = v;
        tot = tot + v;
    }
    if (tot != 0.f)
        for (int j = 0; j < sz; ++j) w[j] = w[j] / tot;
    xmin[i] = lo;
    xsize[i] = sz;
}

// horizontal pass: T[i][y][x] = sum_j w[x][j] * m[i][y][xmin[x]+j]   (m in {0,1}: every product is exact)
__global__ __launch_bounds__(256) void aa_h_kernel(const uint8_t* masks, int oh, int ow, int rw, const int* xmin, const int* xsize, const float* W, float* T) {
    const int i = blockIdx.y;
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= oh * rw) return;
    const int y = pix / rw, x = pix - y * rw;
    const uint8_t* row = masks + ((size_t)i * oh + y) * ow + xmin[x];
    const float* w = W + (size_t)x * AA_KMAX;
    const int sz = xsize[x];
    float t = sz > 0 ? (float)row[0] * w[0] : 0.f;
    for (int j = 1; j < sz; ++j) t = fmaf((float)row[j], w[j], t);
    T[(size_t)i * oh * rw + pix] = t;
}

// vertical pass + threshold + area: v = sum_j w[y][j] * T[i][ymin[y]+j][x] (first term a product, the rest fused multiply-adds)
__global__ __launch_bounds__(256) void aa_v_kernel(const float* T, int oh, int rh, int rw, const int* ymin, const int* ysize, const float* W,
                                                   uint8_t* out, double* area) {
    const int i = blockIdx.y;
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    float v = 0.f;
    if (pix < rh * rw) {
        const int y = pix / rw, x = pix - y * rw;
        const float* col = T + ((size_t)i * oh + ymin[y]) * rw + x;
        const float* w = W + (size_t)y * AA_KMAX;
        const int sz = ysize[y];
        v = sz > 0 ? __fmul_rn(col[0], w[0]) : 0.f;
        for (int j = 1; j < sz; ++j) v = fmaf(col[(size_t)j * rw], w[j], v);
        out[(size_t)i * rh * rw + pix] = v > 0.5f ? 1 : 0;
    }
    // mask.sum() of the resized FLOAT mask (yolo_with_deva.py:75), accumulated in double
    double s = (double)v;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0 && s != 0.0) atomicAdd(&area[i], s);
}

__global__ void mask_ids_f_kernel(const MaskParams p, const double* area, int32_t* kept) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int cur = 1;
    for (int i = 0; i < p.n; ++i) {
        if (p.suppress_small && (float)area[i] < (float)p.min_area) kept[i] = 0;
        else kept[i] = cur++;
    }
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// bytes of device workspace launch_masks needs for these parameters
size_t masks_workspace_bytes(const MaskParams& p) {
    size_t need = 2 * (size_t)((p.n + 3) & ~3) * 4 + (size_t)p.n * p.ch * p.cw * 4 + (p.masks ? 0 : (size_t)p.n * p.oh * p.ow) + 512;
    if (p.rh > 0)
        need += align256((size_t)p.n * 8) + 2 * align256((size_t)(p.rw + p.rh) * 4) + align256((size_t)(p.rw + p.rh) * AA_KMAX * 4) +
                align256((size_t)p.n * p.oh * p.rw * 4) + align256((size_t)p.n * p.rh * p.rw) + 256;
    return need;
}

// workspace layout (device, provided by the engine through p.area):  int32 area[n] | int32 kept[n] | float M[n*ch*cw] | u8 masks[n*oh*ow]
//   (+ when a second resize is asked for: double area_f[n] | int xmin/xsize/ymin/ysize | float Wx/Wy | float T[n*oh*rw] | u8 masks2[n*rh*rw])
hipError_t launch_masks(const MaskParams& p, int dtype, hipStream_t st) {
    const bool second = p.rh > 0;
    const int ph = second ? p.rh : p.oh, pw = second ? p.rw : p.ow;          // size of the painted id image
    if (p.n == 0) {
        if (p.ids) return hipMemsetAsync(p.ids, 0, (size_t)ph * pw * sizeof(int64_t), st);
        return hipSuccess;
    }
    if ((size_t)((p.n + 15) & ~15) * 32 * sizeof(float) > 60 * 1024) return hipErrorInvalidValue;
    int32_t* area = p.area;
    int32_t* kept_ws = area + p.n;
    float* M = (float*)(area + 2 * (size_t)((p.n + 3) & ~3));
    uint8_t* masks = p.masks ? p.masks : (uint8_t*)(M + (size_t)p.n * p.ch * p.cw);
    hipError_t e = hipMemsetAsync(area, 0, (size_t)p.n * sizeof(int32_t), st);
    if (e != hipSuccess) return e;
    const int npix = p.ch * p.cw;
    static const bool valu_gemm = [] { const char* v = std::getenv("YOLOP_MASK_VALU"); return v && *v == '1'; }();   // A/B: scalar form
    if (valu_gemm) {
        const size_t sh = (size_t)p.n * 32 * sizeof(float);
        if (dtype == DT_BF16) hipLaunchKernelGGL(mask_gemm_kernel<__bf16>, dim3((npix + 255) / 256), dim3(256), sh, st, p, M);
        else hipLaunchKernelGGL(mask_gemm_kernel<float>, dim3((npix + 255) / 256), dim3(256), sh, st, p, M);
    } else {
        const size_t sh = (size_t)((p.n + 15) & ~15) * 32 * sizeof(float);
        if (dtype == DT_BF16) hipLaunchKernelGGL(mask_gemm_mfma_kernel<__bf16>, dim3((npix + 63) / 64), dim3(256), sh, st, p, M);
        else hipLaunchKernelGGL(mask_gemm_mfma_kernel<float>, dim3((npix + 63) / 64), dim3(256), sh, st, p, M);
    }
    hipLaunchKernelGGL(mask_resize_kernel, dim3((p.oh * p.ow + 255) / 256, p.n), dim3(256), 0, st, p, M, masks, area);
    if (!p.ids) return hipGetLastError();
    int32_t* kept = p.kept ? p.kept : kept_ws;
    if (!second) {
        hipLaunchKernelGGL(mask_ids_kernel, dim3(1), dim3(64), 0, st, p, area, kept);
        hipLaunchKernelGGL(mask_paint_kernel, dim3((p.oh * p.ow + 255) / 256), dim3(256), 0, st, p, masks, kept, p.ids);
        return hipGetLastError();
    }
    if ((float)p.oh / (float)p.rh > 16.f || (float)p.ow / (float)p.rw > 16.f) return hipErrorInvalidValue;   // (AA_KMAX taps)
    char* q = (char*)(p.masks ? (uint8_t*)(M + (size_t)p.n * p.ch * p.cw) : masks + (size_t)p.n * p.oh * p.ow);
    q = (char*)(((uintptr_t)q + 255) & ~(uintptr_t)255);
    double* area_f = (double*)q; q += align256((size_t)p.n * 8);
    int* xmin = (int*)q; int* xsize = xmin + p.rw; int* ymin = xsize + p.rw; int* ysize = ymin + p.rh; q += 2 * align256((size_t)(p.rw + p.rh) * 4);
    float* Wx = (float*)q; float* Wy = Wx + (size_t)p.rw * AA_KMAX; q += align256((size_t)(p.rw + p.rh) * AA_KMAX * 4);
    float* T = (float*)q; q += align256((size_t)p.n * p.oh * p.rw * 4);
    uint8_t* masks2 = (uint8_t*)q;
    e = hipMemsetAsync(area_f, 0, (size_t)p.n * sizeof(double), st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(aa_weights_kernel, dim3((p.rw + 63) / 64), dim3(64), 0, st, p.ow, p.rw, xmin, xsize, Wx);
    hipLaunchKernelGGL(aa_weights_kernel, dim3((p.rh + 63) / 64), dim3(64), 0, st, p.oh, p.rh, ymin, ysize, Wy);
    hipLaunchKernelGGL(aa_h_kernel, dim3((p.oh * p.rw + 255) / 256, p.n), dim3(256), 0, st, masks, p.oh, p.ow, p.rw, xmin, xsize, Wx, T);
    hipLaunchKernelGGL(aa_v_kernel, dim3((p.rh * p.rw + 255) / 256, p.n), dim3(256), 0, st, T, p.oh, p.rh, p.rw, ymin, ysize, Wy, masks2, area_f);
    hipLaunchKernelGGL(mask_ids_f_kernel, dim3(1), dim3(64), 0, st, p, area_f, kept);
    MaskParams pp = p;
    pp.oh = p.rh; pp.ow = p.rw;
    hipLaunchKernelGGL(mask_paint_kernel, dim3((p.rh * p.rw + 255) / 256), dim3(256), 0, st, pp, masks2, kept, p.ids);
    return hipGetLastError();
}

}  // namespace yp
