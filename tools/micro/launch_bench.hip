// Microbenchmark: fixed cost of a kernel boundary inside a hipGraph chain on MI355X.
//   a) empty kernel, 1 workgroup                       -> pure dispatch + barrier-bit cost
//   b) 256 workgroups x 512 threads x 115 KiB LDS, exit -> + workgroup launch ramp
//   c) streaming copy of `mb` MiB per kernel (ping-pong) -> + L2 write-back at the boundary, vs the same bytes in ONE kernel
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void k_empty(int* p) { if (p && threadIdx.x == 9999) p[0] = 1; }
__global__ __launch_bounds__(512) void k_lds(int* p) {
    extern __shared__ int s[];
    if (p && threadIdx.x == 9999) p[0] = s[0];
}
__global__ __launch_bounds__(256) void k_copy(const uint4* __restrict__ a, uint4* __restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
template <class F> static float graph_time(F body, int reps) {
    hipStream_t st; CHECK(hipStreamCreate(&st));
    hipGraph_t g; hipGraphExec_t ge;
    CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    body(st);
    CHECK(hipStreamEndCapture(st, &g));
    CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipGraphLaunch(ge, st)); CHECK(hipStreamSynchronize(st));
    CHECK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r) CHECK(hipGraphLaunch(ge, st));
    CHECK(hipEventRecord(e1, st)); CHECK(hipStreamSynchronize(st));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipGraphExecDestroy(ge)); CHECK(hipGraphDestroy(g)); CHECK(hipStreamDestroy(st));
    return ms / reps;
}
int main() {
    const int N = 100;
    int* d; CHECK(hipMalloc(&d, 4));
    CHECK(hipFuncSetAttribute((const void*)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 115 * 1024));
    float t = graph_time([&](hipStream_t st) { for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st, d); }, 20);
    printf("a) empty 1-WG kernel chain          : %.2f us per kernel\n", t * 1e3 / N);
    t = graph_time([&](hipStream_t st) { for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, st, d); }, 20);
    printf("a2) empty 256-WG x 256 thr chain    : %.2f us per kernel\n", t * 1e3 / N);
    t = graph_time([&](hipStream_t st) { for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_lds, dim3(256), dim3(512), 115 * 1024, st, d); }, 20);
    printf("b) 256 WG x 512 thr x 115 KiB LDS    : %.2f us per kernel\n", t * 1e3 / N);
    t = graph_time([&](hipStream_t st) { for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_lds, dim3(512), dim3(256), 60 * 1024, st, d); }, 20);
    printf("b2) 512 WG x 256 thr x 60 KiB LDS    : %.2f us per kernel\n", t * 1e3 / N);
    for (int mb : {1, 8, 32, 128}) {
        const size_t n = (size_t)mb * 1024 * 1024 / 16;
        uint4 *a, *b; CHECK(hipMalloc(&a, n * 16)); CHECK(hipMalloc(&b, n * 16)); CHECK(hipMemset(a, 1, n * 16));
        const int grid = 1024;
        t = graph_time([&](hipStream_t st) { for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, st, (i & 1) ? b : a, (i & 1) ? a : b, n); }, 10);
        const double ideal = 2.0 * mb * 1.048576e6 / 6.0e12 * 1e6;   // us at 6 TB/s
        printf("c) copy %3d MiB per kernel (chain)   : %.2f us per kernel  (bytes at 6 TB/s: %.2f us)\n", mb, t * 1e3 / N, ideal);
        CHECK(hipFree(a)); CHECK(hipFree(b));
    }
    return 0;
}
