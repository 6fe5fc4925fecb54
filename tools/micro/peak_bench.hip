// On-box peaks the roofline fractions are priced against (SURVEY 8d asks for them beside the nominal figures):
//   bf16 MFMA   v_mfma_f32_16x16x32_bf16 back to back, operands in registers, 4 and 8 waves per CU (1 and 2 per SIMD), random data
//   HBM         float4 copy of 1 GiB (read + write counted), and a read-only sum
//   L2 -> CU    every workgroup re-reads a small region that stays in its XCD's L2: 2 MB shared by all (the weights' case) and a private
//               64 KB per workgroup (16 MB in all: L2-resident per XCD, nothing shared), float4 loads, 8 waves per CU - the rate a CU can
//               take operands in at, whatever path brings them
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ __launch_bounds__(512) void mfma_kernel(const float* seed, float* sink, int iters) {
    const int lane = threadIdx.x & 63;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)seed[(lane * 8 + i) & 1023]; b[i] = (__bf16)seed[(lane * 8 + i + 512) & 1023]; }
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
    for (int it = 0; it < iters; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c4, 0, 0, 0);
        c5 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c5, 0, 0, 0);
        c6 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c6, 0, 0, 0);
        c7 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c7, 0, 0, 0);
    }
    const f32x4 s = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
    if (s[0] == 123.456f) sink[threadIdx.x] = s[1] + s[2] + s[3];
}

// fp32 matrix rate: v_mfma_f32_16x16x4_f32 back to back on eight accumulators (the U^2-Net parity mode's instruction)
__global__ __launch_bounds__(512) void mfma_f32_kernel(const float* seed, float* sink, int iters) {
    const int lane = threadIdx.x & 63;
    const float a = seed[lane], b = seed[lane + 512];
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
    for (int it = 0; it < iters; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c4, 0, 0, 0);
        c5 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c5, 0, 0, 0);
        c6 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c6, 0, 0, 0);
        c7 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c7, 0, 0, 0);
    }
    const f32x4 s = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
    if (s[0] == 123.456f) sink[threadIdx.x] = s[1] + s[2] + s[3];
}

__global__ __launch_bounds__(256) void copy_kernel(const float4* __restrict__ src, float4* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void read_kernel(const float4* __restrict__ src, float* sink, size_t n) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const float4 v = src[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 123.456f) sink[0] = acc;
}

// Each workgroup sweeps `span` bytes (a multiple of 16 KiB) starting at base + (shared ? 0 : blockIdx.x * span), `reps` times.
__global__ __launch_bounds__(512) void l2_read_kernel(const float4* __restrict__ src, float* sink, size_t span_f4, int reps, int shared) {
    const float4* p = src + (shared ? 0 : (size_t)blockIdx.x * span_f4);
    float acc = 0.f;
    for (int r = 0; r < reps; ++r)
        for (size_t i = threadIdx.x; i < span_f4; i += 512 * 8) {                                  // eight 16-byte loads per lane in flight (64 KB per CU)
            const float4 v0 = p[i], v1 = p[i + 512], v2 = p[i + 1024], v3 = p[i + 1536], v4 = p[i + 2048], v5 = p[i + 2560], v6 = p[i + 3072], v7 = p[i + 3584];
            acc += v0.x + v1.y + v2.z + v3.w + v4.x + v5.y + v6.z + v7.w;
        }
    if (acc == 123.456f) sink[0] = acc;
}

int main() {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float *seed, *sink;
    std::vector<float> h(1024);
    srand(1);
    for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
    CHECK(hipMalloc(&seed, 4096)); CHECK(hipMalloc(&sink, 4096)); CHECK(hipMemcpy(seed, h.data(), 4096, hipMemcpyHostToDevice));
    for (int waves : {4, 8}) {
        const int iters = 20000;
        hipLaunchKernelGGL(mfma_kernel, dim3(256), dim3(waves * 64), 0, 0, seed, sink, 1000);
        CHECK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int r = 0; r < 3; ++r) {
            CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(mfma_kernel, dim3(256), dim3(waves * 64), 0, 0, seed, sink, iters); CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        const double flop = 256.0 * waves * iters * 8 * (2.0 * 16 * 16 * 32);
        printf("bf16 mfma 16x16x32, %d waves/CU: %.1f TFLOP/s (%.3f ms)\n", waves, flop / best * 1e-9, best);
    }
    for (int waves : {4, 8}) {
        const int iters = 20000;
        hipLaunchKernelGGL(mfma_f32_kernel, dim3(256), dim3(waves * 64), 0, 0, seed, sink, 1000);
        CHECK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int r = 0; r < 3; ++r) {
            CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(mfma_f32_kernel, dim3(256), dim3(waves * 64), 0, 0, seed, sink, iters); CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        const double flop = 256.0 * waves * iters * 8 * (2.0 * 16 * 16 * 4);
        printf("fp32 mfma 16x16x4, %d waves/CU: %.1f TFLOP/s (%.3f ms)\n", waves, flop / best * 1e-9, best);
    }
    const size_t bytes = 1ull << 30;
    float4 *a, *b;
    CHECK(hipMalloc(&a, bytes)); CHECK(hipMalloc(&b, bytes)); CHECK(hipMemset(a, 1, bytes)); CHECK(hipMemset(b, 2, bytes));
    for (int pass = 0; pass < 2; ++pass) {
        float best = 1e30f;
        for (int r = 0; r < 5; ++r) {
            CHECK(hipEventRecord(e0));
            if (pass == 0) hipLaunchKernelGGL(copy_kernel, dim3(256 * 16), dim3(256), 0, 0, a, b, bytes / 16);
            else hipLaunchKernelGGL(read_kernel, dim3(256 * 16), dim3(256), 0, 0, a, sink, bytes / 16);
            CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("%s 1 GiB: %.0f GB/s (%.3f ms)\n", pass == 0 ? "float4 copy (read+write)" : "float4 read", (pass == 0 ? 2.0 : 1.0) * bytes / best * 1e-6, best);
    }
    for (int shared = 1; shared >= 0; --shared) {
        const size_t span = shared ? (2u << 20) : (64u << 10);          // bytes per workgroup sweep
        const int reps = shared ? 40 : 1280;
        hipLaunchKernelGGL(l2_read_kernel, dim3(256), dim3(512), 0, 0, (const float4*)a, sink, span / 16, 2, shared);
        CHECK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int r = 0; r < 3; ++r) {
            CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(l2_read_kernel, dim3(256), dim3(512), 0, 0, (const float4*)a, sink, span / 16, reps, shared); CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        const double bytes_cu = (double)span * reps;
        printf("L2-resident read, %s: %.1f GB/s per CU = %.1f B/clk/CU at 2.4 GHz (%.1f TB/s over 256 CUs; %.3f ms)\n",
               shared ? "one 2-MB region shared by all workgroups" : "a private 64-KB region per workgroup", bytes_cu / best * 1e-6, bytes_cu / best * 1e-6 / 2.4,
               256.0 * bytes_cu / best * 1e-9, best);
    }
    return 0;
}
