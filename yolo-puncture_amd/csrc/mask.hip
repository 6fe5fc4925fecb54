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

// workspace layout (device, provided by the engine through p.area):  int32 area[n] | int32 kept[n] | float M[n*ch*cw] | u8 masks[n*oh*ow]
hipError_t launch_masks(const MaskParams& p, int dtype, hipStream_t st) {
    if (p.n == 0) {
        if (p.ids) return hipMemsetAsync(p.ids, 0, (size_t)p.oh * p.ow * sizeof(int64_t), st);
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
    if (p.ids) {
        int32_t* kept = p.kept ? p.kept : kept_ws;
        hipLaunchKernelGGL(mask_ids_kernel, dim3(1), dim3(64), 0, st, p, area, kept);
        hipLaunchKernelGGL(mask_paint_kernel, dim3((p.oh * p.ow + 255) / 256), dim3(256), 0, st, p, masks, kept, p.ids);
    }
    return hipGetLastError();
}

}  // namespace yp
