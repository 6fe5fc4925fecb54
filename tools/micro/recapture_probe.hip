// Probe for the host SIGSEGV of rounds 2-3 (hipGraphLaunch of a RE-captured multi-lane graph; gpurun_out/t11.log, t12.log of round 3):
// is it the re-use of events across stream captures? One mode per process (a crash must not hide the other modes); tools/micro/recapture.sh
// runs them all and prints each exit status.
//   mode a: lanes as streams + events, ONE event set re-used by every capture (the scheme of rounds 1-3), replay on a created stream
//   mode b: as a, replay on the legacy NULL stream
//   mode c: fresh events for every capture, replay on the NULL stream
//   mode d: as b, plus the pattern "fork event recorded before the capturing stream has captured a node" in the second capture
//   mode e: single-stream capture with hipStreamUpdateCaptureDependencies (the scheme of round 4), replay on the NULL stream
// Every mode captures, replays, destroys, re-captures with other kernel arguments and replays again, 20 times over; results are checked.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

__global__ void k_add(int* p, int v) { atomicAdd(p + (blockIdx.x & 7), v); }

static const int NL = 6, NK = 5;           // side lanes, kernels per lane

struct Lanes {
    hipStream_t main_s; std::vector<hipStream_t> s; std::vector<hipEvent_t> fork, join;
};

static void make_events(Lanes& L) {
    L.fork.resize(NL); L.join.resize(NL);
    for (int l = 0; l < NL; ++l) { CHECK(hipEventCreateWithFlags(&L.fork[l], hipEventDisableTiming)); CHECK(hipEventCreateWithFlags(&L.join[l], hipEventDisableTiming)); }
}
static void drop_events(Lanes& L) {
    for (auto e : L.fork) CHECK(hipEventDestroy(e));
    for (auto e : L.join) CHECK(hipEventDestroy(e));
    L.fork.clear(); L.join.clear();
}

// body of the captured work: 3 kernels on the main lane, fork into NL lanes of NK kernels, join, 2 kernels
static void capture_events(Lanes& L, int* d, int v, bool early_fork, hipGraph_t* g) {
    CHECK(hipStreamBeginCapture(L.main_s, hipStreamCaptureModeThreadLocal));
    if (early_fork) {                       // lane 0 forks before the capturing stream holds a node
        CHECK(hipEventRecord(L.fork[0], L.main_s));
        CHECK(hipStreamWaitEvent(L.s[0], L.fork[0], 0));
    }
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_add, dim3(64), dim3(64), 0, L.main_s, d, v);
    for (int l = 0; l < NL; ++l) {
        if (!(early_fork && l == 0)) {
            CHECK(hipEventRecord(L.fork[l], L.main_s));
            CHECK(hipStreamWaitEvent(L.s[l], L.fork[l], 0));
        }
        for (int i = 0; i < NK; ++i) hipLaunchKernelGGL(k_add, dim3(64), dim3(64), 0, L.s[l], d, v);
    }
    for (int l = 0; l < NL; ++l) {
        CHECK(hipEventRecord(L.join[l], L.s[l]));
        CHECK(hipStreamWaitEvent(L.main_s, L.join[l], 0));
    }
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k_add, dim3(64), dim3(64), 0, L.main_s, d, v);
    CHECK(hipStreamEndCapture(L.main_s, g));
}

static void capture_deps(Lanes& L, int* d, int v, hipGraph_t* g) {
    hipStream_t cs = L.main_s;
    auto tail = [&]() {
        hipStreamCaptureStatus st; unsigned long long id; hipGraph_t gg; const hipGraphNode_t* dn; size_t nd;
        CHECK(hipStreamGetCaptureInfo_v2(cs, &st, &id, &gg, &dn, &nd));
        return std::vector<hipGraphNode_t>(dn, dn + nd);
    };
    CHECK(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_add, dim3(64), dim3(64), 0, cs, d, v);
    std::vector<hipGraphNode_t> root = tail(), all;
    for (int l = 0; l < NL; ++l) {
        CHECK(hipStreamUpdateCaptureDependencies(cs, root.data(), root.size(), hipStreamSetCaptureDependencies));
        for (int i = 0; i < NK; ++i) hipLaunchKernelGGL(k_add, dim3(64), dim3(64), 0, cs, d, v);
        std::vector<hipGraphNode_t> t = tail();
        all.insert(all.end(), t.begin(), t.end());
    }
    CHECK(hipStreamUpdateCaptureDependencies(cs, all.data(), all.size(), hipStreamSetCaptureDependencies));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k_add, dim3(64), dim3(64), 0, cs, d, v);
    CHECK(hipStreamEndCapture(cs, g));
}

int main(int argc, char** argv) {
    const char mode = argc > 1 ? argv[1][0] : 'a';
    int* d; CHECK(hipMalloc(&d, 8 * sizeof(int))); CHECK(hipMemset(d, 0, 8 * sizeof(int)));
    Lanes L;
    CHECK(hipStreamCreateWithFlags(&L.main_s, hipStreamNonBlocking));
    L.s.resize(NL);
    for (auto& s : L.s) CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipStream_t user; CHECK(hipStreamCreateWithFlags(&user, hipStreamNonBlocking));
    hipLaunchKernelGGL(k_add, dim3(64), dim3(64), 0, L.main_s, d, 0);      // module loaded before any capture
    CHECK(hipStreamSynchronize(L.main_s));
    if (mode != 'c' && mode != 'e') make_events(L);
    long long expect = 0;
    hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
    const int per_graph = (3 + NL * NK + 2) * 64 * 64 / 8;                  // adds per slot and replay (64 blocks over 8 slots x 64 threads)
    for (int round = 0; round < 20; ++round) {
        const int v = round + 1;
        if (ge) { CHECK(hipDeviceSynchronize()); CHECK(hipGraphExecDestroy(ge)); CHECK(hipGraphDestroy(g)); ge = nullptr; g = nullptr; }
        if (mode == 'c') make_events(L);
        if (mode == 'e') capture_deps(L, d, v, &g);
        else capture_events(L, d, v, mode == 'd' && round > 0, &g);
        size_t nn = 0, ne = 0;
        CHECK(hipGraphGetNodes(g, nullptr, &nn));
        CHECK(hipGraphGetEdges(g, nullptr, nullptr, &ne));
        CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        hipStream_t rs = (mode == 'a') ? user : nullptr;
        for (int rep = 0; rep < 5; ++rep) CHECK(hipGraphLaunch(ge, rs));
        CHECK(hipStreamSynchronize(rs));
        expect += 5ll * per_graph * v;
        if (mode == 'c') drop_events(L);
        if (round == 0 || round == 19) printf("mode %c round %d: %zu nodes, %zu edges\n", mode, round, nn, ne);
    }
    int h[8];
    CHECK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
    for (int i = 0; i < 8; ++i) if (h[i] != (int)expect) { printf("mode %c: WRONG RESULT slot %d: %d != %lld\n", mode, i, h[i], expect); return 3; }
    printf("mode %c: ok (20 captures, 100 replays, results right)\n", mode);
    return 0;
}
