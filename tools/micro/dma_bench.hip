// Microbenchmark: operand delivery rate into a CU, LDS-DMA (buffer_load ... lds, 16 B/lane) vs global_load_dwordx4,
// for the access shapes the conv kernels use. One workgroup per CU (grid = 256), 4 or 8 waves, each wave issues
// `iters` x `unroll` 1-KiB loads. Source footprint selects the level that serves it (L2-resident slab vs HBM stream).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// mode 0: LDS-DMA ; mode 1: global->VGPR (sum kept live)
// shape: row_bytes in {64,128,256,1024}: a wave's 1 KiB is (1024/row_bytes) rows, consecutive rows `row_stride` bytes apart
template <int MODE>
__global__ __launch_bounds__(1024) void dma_kernel(const unsigned char* src, size_t src_bytes, int row_bytes, int row_stride,
                                                  size_t wave_stride, size_t iter_stride, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)src_bytes, 0x00020000);
    const int lanes_per_row = row_bytes / 16;
    const int row = lane / lanes_per_row, col = lane % lanes_per_row;
    size_t base = ((size_t)blockIdx.x * nw + wave) * wave_stride + (size_t)row * row_stride + col * 16;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned off = (unsigned)((base + (size_t)u * 1024 * (row_stride / row_bytes)) % (src_bytes - 1024 * 64));
            if (MODE == 0) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(lds + (wave * 8 + u) * 1024), 16, off, 0, 0, 0);
            } else {
                const uint4 v = *(const uint4*)(src + off);
                acc += __uint_as_float(v.x & 0x3f800000u) + __uint_as_float(v.w & 0x3f800000u);
            }
        }
        base += iter_stride;
        if (MODE == 0) __builtin_amdgcn_s_waitcnt((8 & 0xF) | (7 << 4) | (0xF << 8));     // keep <= 8 in flight per wave
    }
    if (MODE == 0) __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (0xF << 8));
    if (acc == 123.456f) sink[0] = acc + lds[threadIdx.x];
}

int main() {
    const size_t big = 1ull << 30;   // 1 GiB source
    unsigned char* src; float* sink;
    CHECK(hipMalloc(&src, big)); CHECK(hipMemset(src, 1, big)); CHECK(hipMalloc(&sink, 4));
    CHECK(hipFuncSetAttribute((const void*)dma_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    struct Case { const char* name; int row_bytes, row_stride; size_t footprint; };
    // footprint: bytes the whole grid cycles through (small -> L2 hits after the first pass)
    const Case cases[] = {
        {"64B rows  / L2-resident 2MB ", 64, 256, 2u << 20},   {"128B rows / L2-resident 2MB ", 128, 256, 2u << 20},
        {"1KB contig/ L2-resident 2MB ", 1024, 1024, 2u << 20}, {"64B rows  / MALL 64MB        ", 64, 256, 64u << 20},
        {"128B rows / MALL 64MB        ", 128, 256, 64u << 20}, {"1KB contig/ MALL 64MB        ", 1024, 1024, 64u << 20},
        {"64B rows  / HBM 1GB          ", 64, 256, big},        {"128B rows / HBM 1GB          ", 128, 256, big},
        {"1KB contig/ HBM 1GB          ", 1024, 1024, big},
    };
    for (int nwaves : {2, 4, 8, 12, 16}) {
        for (int mode = 0; mode < 1; ++mode) {
            for (const Case& c : cases) {
                const int grid = 256, iters = 256;
                // every wave walks its own stream; per-iteration stride chosen so the grid sweeps `footprint`
                const size_t span = (size_t)8 * 1024 * (c.row_stride / c.row_bytes);       // bytes of address space one wave-iteration covers
                const size_t wave_stride = span;
                size_t iter_stride = span * grid * nwaves;
                const size_t fp = c.footprint;
                if (iter_stride >= fp) iter_stride = 0;                                     // tiny footprint: re-read the same lines
                auto launch = [&]() {
                    if (mode == 0) hipLaunchKernelGGL(dma_kernel<0>, dim3(grid), dim3(nwaves * 64), 128 * 1024, 0, src, fp, c.row_bytes, c.row_stride, wave_stride, iter_stride, iters, sink);
                    else hipLaunchKernelGGL(dma_kernel<1>, dim3(grid), dim3(nwaves * 64), 128 * 1024, 0, src, fp, c.row_bytes, c.row_stride, wave_stride, iter_stride, iters, sink);
                };
                launch(); CHECK(hipDeviceSynchronize());
                CHECK(hipEventRecord(e0)); launch(); CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                const double bytes = (double)grid * nwaves * iters * 8 * 1024;
                printf("%s waves=%d %s : %7.1f GB/s total, %6.1f GB/s/CU, %5.1f B/clk/CU @2.4GHz\n", mode == 0 ? "lds-dma " : "vgpr    ", nwaves, c.name,
                       bytes / ms * 1e-6, bytes / ms * 1e-6 / grid, bytes / (ms * 1e-3) / grid / 2.4e9);
            }
        }
    }
    return 0;
}
