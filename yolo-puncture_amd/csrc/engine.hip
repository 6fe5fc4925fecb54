// libyolop.so: native graph builder + executor behind the C-ABI of include/yolop.h.
//
// yp_create builds the YOLOv10 op graph (SURVEY.md Appendix A.3/A.4 [U]) for a variant from its scale
// triple - the job ultralytics' parse_model does inside `YOLO(path)` (reference yolo_seg/app.py:45) - with
// every `Concat`/`chunk` resolved at build time into channel-slice views of shared NHWC buffers, so no copy
// kernel exists for them. yp_forward replays the op list on a HIP stream (optionally as one hipGraph).
#include "../../include/yolop.h"
#include "common.h"
#include <algorithm>
#include <array>
#include <chrono>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <dlfcn.h>
#include <fcntl.h>
#include <fstream>
#include <map>
#include <sstream>
#include <memory>

using namespace yp;

static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
extern "C" int yp_fail_public(int code, const char* msg) { return fail(code, "%s", msg); }   // for the other translation units
#define HIPCHK(x)                                                                                   \
    do {                                                                                            \
        hipError_t _e = (x);                                                                        \
        if (_e != hipSuccess) return fail(YP_ERR_HIP, "%s failed: %s (%s:%d)", #x, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

struct yp_engine {
    yp_model_desc desc{};
    int device = 0;
    int dtype = DT_BF16;
    std::vector<TensorDesc> tensors;
    std::vector<WeightDesc> weights;
    std::vector<Op> ops;
    std::map<std::string, int> wmap;
    bool finalized = false;
    // plan
    int pB = 0, pH = 0, pW = 0;
    bool planned = false, allocated = false;
    void* arena = nullptr;
    size_t arena_bytes = 0;
    int proto_t = -1;
    void* mask_ws = nullptr;
    void* head_ws = nullptr;
    size_t head_ws_bytes = 0;
    size_t mask_ws_bytes = 0;
    // hipGraph replay
    bool use_graph = false;
    bool graph_auto = false;              // yp_set_graph(3): replay or eager per plan, whichever a one-off timing finds faster
    std::map<std::array<int, 3>, bool> auto_replay;
    bool fuse = true;             // dw->pw fusion (YOLOP_NO_FUSE=1 disables, for A/B)
    bool tail = false;            // conv_dwpw TAIL form (YOLOP_TAIL=1 at yp_create enables; see make_plan)
    bool sparse_head = true;      // v10 head: box / coefficient branches on the stage-1 winners only (YOLOP_DENSE_HEAD=1 at yp_create disables)
    void* sp_ws = nullptr; size_t sp_ws_bytes = 0;   // winners-only head: sel / wlist / wcount / thr / box rows / coefficient rows
    bool tune = true;             // plan-time autotuning of the conv tile configuration
    int tune_source = 0;          // where the current plan's tile configurations came from: 0 the tuner (or heuristics), 1 a YOLOP_TUNE_CACHE file, 2 a packaged table
    hipStream_t own_stream = nullptr;
    hipEvent_t ev_in = nullptr;                      // orders the own-stream replay behind a caller on the legacy NULL stream (forward_replay)
    hipGraphExec_t gexec = nullptr;
    hipGraph_t gsrc = nullptr;                       // the captured graph gexec was instantiated from (kept for yp_debug_graph_info; destroyed with it)
    bool use_lanes = true;
    int n_captures = 0;                              // captures + instantiations so far (yp_debug_graph_info: a steady-state caller must not add to it)
    int g_nodes = 0, g_edges = 0, g_edges_expected = 0;
    hipStream_t last_stream = nullptr; bool have_last = false;   // the stream the previous forward of this engine was enqueued on
    // lane schedule of the current plan (pure host data, build_lane_schedule): what capture_dag turns into graph edges
    struct LaneStep { int op = -1; std::vector<int> waits; bool fork = false; bool record = false; };
    std::vector<LaneStep> lane_steps;
    std::vector<int> lanes_used;             // side lanes that launch at least one op under this plan (only these join lane 0)
    int n_lanes = 1;
    bool warmed = false;                     // one eager pass of this plan has run (modules loaded, attributes set) - required before a capture
    // autotuner results per input shape seen so far: switching between shapes re-plans but does not re-tune
    struct Tuned { int cfg; std::string kernel; };
    std::map<std::array<int, 3>, std::vector<Tuned>> tuned;
    struct Key { int B = 0, H = 0, W = 0; const void* in = nullptr; const void* det = nullptr; const void* idx = nullptr; const void* coeff = nullptr; } gkey;   // what the captured graph is specialised on
    // Output mode of the replay. Direct (the start): the graph writes the caller's buffers, which most callers keep from call to call
    // (bench.py, predictor.py's per-shape output cache) - no copy kernel behind the graph. A caller that brings NEW output buffers would
    // force capture + instantiate per call: the second change of pointers switches this engine to engine-owned results + a 0.3 MB
    // copy-out for good.
    bool direct_out = true;
    int out_changes = 0;
    hipEvent_t ev_done = nullptr;         // recorded behind the last replay on the stream it ran on (an executable is never destroyed under the GPU)
    // engine-owned results of the replayed graph: the graph never references the caller's output buffers (they change from
    // call to call in ordinary use, and every change would mean capture + instantiate + destroy); yp_forward copies the
    // ~0.3 MB out behind the replay
    float* o_det = nullptr; int32_t* o_idx = nullptr; float* o_coeff = nullptr; size_t o_cap = 0;
    // NMS heads (families v8 / 11)
    const uint8_t* tune_input = nullptr;  // the caller's frames of the forward that triggered the tuner (producer ops that read them)
    float nms_conf = 0.25f, nms_iou = 0.7f;
    float* d_nms = nullptr;               // device copy of [conf, iou]
    bool nms_dirty = false;               // the host pair is newer than the device copy: the next forward rewrites it on its stream
    void* nms_ws = nullptr; size_t nms_ws_bytes = 0;
    int es() const { return dtype == DT_BF16 ? 2 : 4; }
};

// =========================================================================================================
// graph builder
// =========================================================================================================
namespace {

struct Scale { double depth, width; int maxc; };
static bool variant_scale(int v, Scale& s) {
    switch (v) {
        case 'n': s = {0.33, 0.25, 1024}; return true;
        case 's': s = {0.33, 0.50, 1024}; return true;
        case 'm': s = {0.67, 0.75, 768}; return true;
        case 'b': s = {0.67, 1.00, 512}; return true;
        case 'l': s = {1.00, 1.00, 512}; return true;
        case 'x': s = {1.00, 1.25, 512}; return true;
    }
    return false;
}
static int make_div8(double x) { return (int)std::ceil(x / 8.0) * 8; }
static int py_round(double x) {   // Python round(): half to even
    double r = std::nearbyint(x);
    return (int)r;
}

struct Builder {
    yp_engine& e;
    int lane = 0;     // capture lane stamped on every op created from here on
    explicit Builder(yp_engine& en) : e(en) {}

    int tensor(const std::string& name, int C, int sdiv, bool f32 = false) {
        TensorDesc t;
        t.name = name; t.C = C; t.sdiv = sdiv; t.f32 = f32;
        e.tensors.push_back(t);
        return (int)e.tensors.size() - 1;
    }
    View full(int t) const { return View{t, 0, e.tensors[t].C}; }
    static View slice(View v, int off, int C) { return View{v.t, v.coff + off, C}; }

    int weight(const std::string& name, int cout, int cin_g, int k, int groups, bool transposed = false, bool stem = false) {
        WeightDesc w;
        w.name = name; w.cout = cout; w.cin_g = cin_g; w.k = k; w.groups = groups; w.transposed = transposed; w.is_stem = stem;
        w.cin_pad = cin_g;
        // Channel counts such as 80 or 400 (v10-X / -M) are not multiples of the 32-deep operand chunk: the LDS-DMA kernels would be out,
        // leaving conv_igemm. Packing the weights with each tap's channels padded to 32 (zeros in the gap) lets every kernel run with
        // Cin = cin_pad: the extra 16 channels it reads per pixel are the NEXT pixel's (or slice's) first ones - finite values (the arena is
        // zero-filled when allocated) times zero weights.
        if (e.dtype == DT_BF16 && !stem && groups == 1 && !transposed && cin_g > 32 && (cin_g % 32) != 0 && (cin_g % 8) == 0) w.cin_pad = (cin_g + 31) / 32 * 32;
        if (!stem && groups == 1) {         // packed GEMM geometry (pack_weight fills exactly this): known from the shape alone, so a plan
            const int K = transposed ? cin_g : k * k * w.cin_pad;   // made before yp_finalize takes the same decisions as one made after
            w.Kpad = (K + 31) / 32 * 32;
            w.mat_bytes = (size_t)((cout + 127) / 128 * 128) * w.Kpad * e.es();
        }
        e.weights.push_back(w);
        e.wmap[name] = (int)e.weights.size() - 1;
        return (int)e.weights.size() - 1;
    }
    // dense conv (+folded BN) (+SiLU) (+residual)
    void conv(const std::string& name, View in, View out, int k, int s, int act, View res = View{}) {
        Op o;
        o.kind = OP_CONV; o.name = name; o.in = in; o.out = out; o.res = res; o.k = k; o.s = s; o.act = act;
        o.widx = weight(name, out.C, in.C, k, 1);
        o.lane = lane;
        e.ops.push_back(o);
    }
    void dwconv(const std::string& name, View in, View out, int k, int s, int act, View res = View{}, int gs = 0, int gstride = 0) {
        Op o;
        o.kind = OP_DWCONV; o.name = name; o.in = in; o.out = out; o.res = res; o.k = k; o.s = s; o.act = act;
        o.gs = gs; o.gstride = gstride;
        o.widx = weight(name, out.C, 1, k, out.C);
        o.lane = lane;
        e.ops.push_back(o);
    }
    int sdiv_of(View v) const { return e.tensors[v.t].sdiv; }

    // C2f / C2fCIB (A.2): cv1 -> chunk(2) -> n x (Bottleneck | CIB) -> cat -> cv2, all inside one concat buffer
    void c2f(const std::string& p, View in, View out, int c2, int n, bool shortcut, bool cib, bool lk) {
        const int c = c2 / 2, sd = sdiv_of(in);
        const int Y = tensor(p + ".cat", (2 + n) * c, sd);
        int m1 = -1;
        conv(p + ".cv1", in, View{Y, 0, 2 * c}, 1, 1, ACT_SILU);
        for (int j = 0; j < n; ++j) {
            const View x{Y, (1 + j) * c, c}, y{Y, (2 + j) * c, c};
            const std::string q = p + ".m." + std::to_string(j);
            const View res = shortcut ? x : View{};
            if (!cib) {
                const int t = tensor(q + ".cv1", c, sd);
                conv(q + ".cv1", x, full(t), 3, 1, ACT_SILU);
                conv(q + ".cv2", full(t), y, 3, 1, ACT_SILU, res);
                if (n == 1) m1 = (int)e.ops.size() - 2;
            } else {
                const int t0 = tensor(q + ".cv1.0", c, sd), t1 = tensor(q + ".cv1.1", 2 * c, sd);
                const int t2 = tensor(q + ".cv1.2", 2 * c, sd), t3 = tensor(q + ".cv1.3", c, sd);
                dwconv(q + ".cv1.0", x, full(t0), 3, 1, ACT_SILU);
                conv(q + ".cv1.1", full(t0), full(t1), 1, 1, ACT_SILU);
                dwconv(q + ".cv1.2", full(t1), full(t2), lk ? 7 : 3, 1, ACT_SILU);   // RepVGGDW pre-merged to one 7x7
                conv(q + ".cv1.3", full(t2), full(t3), 1, 1, ACT_SILU);
                dwconv(q + ".cv1.4", full(t3), y, 3, 1, ACT_SILU, res);
            }
        }
        conv(p + ".cv2", full(Y), out, 1, 1, ACT_SILU);
        if (m1 >= 0) { e.ops.back().c2f_m1 = m1; e.ops.back().c2f_m2 = m1 + 1; }    // (t and the c slice have no other reader)
    }
    // plain Bottleneck with hidden width ch (C3k2's default e = 0.5; C3k's inner ones e = 1.0)
    void bottleneck(const std::string& q, View x, View y, int ch, bool shortcut) {
        const int t = tensor(q + ".cv1", ch, sdiv_of(x));
        conv(q + ".cv1", x, full(t), 3, 1, ACT_SILU);
        conv(q + ".cv2", full(t), y, 3, 1, ACT_SILU, shortcut ? x : View{});
    }
    // C3k2 (ultralytics block.py, YOLO11): C2f whose inner modules are Bottleneck(c, c, e = 0.5) or C3k(c, c, 2)
    void c3k2(const std::string& p, View in, View out, int c2, int n, bool c3k, double e) {
        const int c = (int)(c2 * e), sd = sdiv_of(in);
        const int Y = tensor(p + ".cat", (2 + n) * c, sd);
        conv(p + ".cv1", in, View{Y, 0, 2 * c}, 1, 1, ACT_SILU);
        for (int j = 0; j < n; ++j) {
            const View x{Y, (1 + j) * c, c}, y{Y, (2 + j) * c, c};
            const std::string q = p + ".m." + std::to_string(j);
            const int c_ = c / 2;
            if (!c3k) { bottleneck(q, x, y, c_, true); continue; }
            // C3k: cv3(cat(m(cv1(x)), cv2(x))), m = two Bottleneck(c_, c_, k = 3, e = 1.0)
            const int Z = tensor(q + ".cat", 2 * c_, sd);
            const int a0 = tensor(q + ".cv1", c_, sd), a1 = tensor(q + ".m.0", c_, sd);
            conv(q + ".cv1", x, full(a0), 1, 1, ACT_SILU);
            bottleneck(q + ".m.0", full(a0), full(a1), c_, true);
            bottleneck(q + ".m.1", full(a1), View{Z, 0, c_}, c_, true);
            conv(q + ".cv2", x, View{Z, c_, c_}, 1, 1, ACT_SILU);
            conv(q + ".cv3", full(Z), y, 1, 1, ACT_SILU);
        }
        conv(p + ".cv2", full(Y), out, 1, 1, ACT_SILU);
    }
    // C2PSA (YOLO11): cv1 -> split(a, b) -> n x PSABlock on b -> cv2(cat(a, b))
    void c2psa(const std::string& p, View in, View out, int n) {
        const int c = in.C / 2, sd = sdiv_of(in);
        const int nh = c / 64, hd = c / nh, kd = hd / 2;
        const int P = tensor(p + ".cv1", 2 * c, sd);
        conv(p + ".cv1", in, full(P), 1, 1, ACT_SILU);
        const View b{P, c, c};
        for (int j = 0; j < n; ++j) {
            const std::string q = p + ".m." + std::to_string(j);
            const int Q = tensor(q + ".attn.qkv", c + 2 * kd * nh, sd);
            conv(q + ".attn.qkv", b, full(Q), 1, 1, ACT_NONE);
            const int O = tensor(q + ".attn.o", c, sd);
            {
                Op o;
                o.kind = OP_ATTN; o.name = q + ".attn.o"; o.in = full(Q); o.out = full(O); o.nh = nh; o.kd = kd; o.hd = hd;
                e.ops.push_back(o);
            }
            const int Yp = tensor(q + ".attn.pe", c, sd);
            dwconv(q + ".attn.pe", View{Q, 2 * kd, c}, full(Yp), 3, 1, ACT_NONE, full(O), hd, 2 * kd + hd);
            const int B1 = tensor(q + ".attn.proj", c, sd);
            conv(q + ".attn.proj", full(Yp), full(B1), 1, 1, ACT_NONE, b);
            const int Fh = tensor(q + ".ffn.0", 2 * c, sd);
            conv(q + ".ffn.0", full(B1), full(Fh), 1, 1, ACT_SILU);
            conv(q + ".ffn.1", full(Fh), b, 1, 1, ACT_NONE, full(B1));
        }
        conv(p + ".cv2", full(P), out, 1, 1, ACT_SILU);
    }
    void scdown(const std::string& p, View in, View out) {
        const int t = tensor(p + ".cv1", out.C, sdiv_of(in));
        conv(p + ".cv1", in, full(t), 1, 1, ACT_SILU);
        dwconv(p + ".cv2", full(t), out, 3, 2, ACT_NONE);
        e.ops.back().scd_pre = (int)e.ops.size() - 2;                 // (t has no other reader)
    }
    void sppf(const std::string& p, View in, View out) {
        const int c_ = in.C / 2, sd = sdiv_of(in);
        const int S = tensor(p + ".cat", 4 * c_, sd);
        conv(p + ".cv1", in, View{S, 0, c_}, 1, 1, ACT_SILU);
        {
            Op o;                                   // m(y), m(m(y)), m(m(m(y))) -> slices 1..3 of the concat buffer
            o.kind = OP_POOL3; o.name = p + ".m";
            o.in = View{S, 0, c_}; o.out = View{S, c_, 3 * c_};
            e.ops.push_back(o);
        }
        conv(p + ".cv2", full(S), out, 1, 1, ACT_SILU);
    }
    void psa(const std::string& p, View in, View out) {
        const int c = in.C / 2, sd = sdiv_of(in);
        const int nh = c / 64, hd = c / nh, kd = hd / 2;
        const int P = tensor(p + ".cv1", 2 * c, sd);
        conv(p + ".cv1", in, full(P), 1, 1, ACT_SILU);
        const View a{P, 0, c}, b{P, c, c};
        (void)a;
        const int Q = tensor(p + ".attn.qkv", c + 2 * kd * nh, sd);
        conv(p + ".attn.qkv", b, full(Q), 1, 1, ACT_NONE);
        const int O = tensor(p + ".attn.o", c, sd);
        {
            Op o;
            o.kind = OP_ATTN; o.name = p + ".attn.o"; o.in = full(Q); o.out = full(O); o.nh = nh; o.kd = kd; o.hd = hd;
            e.ops.push_back(o);
        }
        // o + pe(v): depthwise 3x3 over the v rows of qkv (channel h*hd+d lives at h*(2kd+hd)+2kd+d)
        const int Y = tensor(p + ".attn.pe", c, sd);
        dwconv(p + ".attn.pe", View{Q, 2 * kd, c}, full(Y), 3, 1, ACT_NONE, full(O), hd, 2 * kd + hd);
        const int B1 = tensor(p + ".attn.proj", c, sd);
        conv(p + ".attn.proj", full(Y), full(B1), 1, 1, ACT_NONE, b);      // b = b + attn(b)
        const int F = tensor(p + ".ffn.0", 2 * c, sd);
        conv(p + ".ffn.0", full(B1), full(F), 1, 1, ACT_SILU);
        conv(p + ".ffn.1", full(F), b, 1, 1, ACT_NONE, full(B1));           // b = b + ffn(b), written over the dead b
        conv(p + ".cv2", full(P), out, 1, 1, ACT_SILU);
    }
    void upsample(const std::string& name, View in, View out) {
        Op o;
        o.kind = OP_UPSAMPLE; o.name = name; o.in = in; o.out = out;
        e.ops.push_back(o);
    }
};

static int finish_graph_passes(yp_engine& e);

// YOLOv8-seg / YOLO11-seg (ultralytics cfg/models/v8/yolov8-seg.yaml, cfg/models/11/yolo11-seg.yaml [U]; the checkpoints of
// reference yolo_seg/app.py:218-223 and yolo_seg/yolo_with_deva.py:226): backbone + PAN neck + Segment head (box / class /
// mask-coefficient branches per level, Proto on P3), post-process = conf filter + NMS (OP_HEAD with nms)
static int build_graph_seg(yp_engine& e) {
    const int fam = e.desc.family, v = e.desc.variant;
    Scale sc;
    if (fam == YP_FAMILY_V8) {
        switch (v) { case 'n': sc = {0.33, 0.25, 1024}; break; case 's': sc = {0.33, 0.50, 1024}; break; case 'm': sc = {0.67, 0.75, 768}; break;
                     case 'l': sc = {1.00, 1.00, 512}; break; case 'x': sc = {1.00, 1.25, 512}; break; default: return fail(YP_ERR_ARG, "unknown variant '%c'", v); }
    } else {
        switch (v) { case 'n': sc = {0.50, 0.25, 1024}; break; case 's': sc = {0.50, 0.50, 1024}; break; case 'm': sc = {0.50, 1.00, 512}; break;
                     case 'l': sc = {1.00, 1.00, 512}; break; case 'x': sc = {1.00, 1.50, 512}; break; default: return fail(YP_ERR_ARG, "unknown variant '%c'", v); }
    }
    if (e.desc.task != YP_TASK_SEGMENT) return fail(YP_ERR_ARG, "the v8 / 11 families are segmentation models here (task must be YP_TASK_SEGMENT)");
    Builder B(e);
    auto C = [&](int c) { return make_div8(std::min(c, sc.maxc) * sc.width); };
    auto N = [&](int n) { return n > 1 ? std::max(py_round(n * sc.depth), 1) : n; };
    const bool v8 = fam == YP_FAMILY_V8;
    const bool big = !v8 && (v == 'm' || v == 'l' || v == 'x');       // parse_model: C3k2 gets c3k = True for scales m, l, x
    const int c0 = C(64), c1 = C(128), c2 = C(v8 ? 128 : 256), c3 = C(256), c4 = C(v8 ? 256 : 512), c5 = C(512), c6 = C(512), c7 = C(1024),
              c8 = C(1024), c9 = C(1024), c10 = C(1024), cn4 = C(512), cn3 = C(256), cd3 = C(256), cn4b = C(512), cd4 = C(512), cn5 = C(1024);
    const int ctop = v8 ? c9 : c10;                                      // what the neck starts from (SPPF output / C2PSA output)
    const int idx0 = v8 ? 10 : 11;                                       // layer index of the first Upsample
    auto L = [&](int i) { return "model." + std::to_string(i); };
    // concat buffers of the neck, producers write straight into their slice
    const int Ta = B.tensor(L(idx0 + 1), ctop + c6, 16);                // [up(top), L6]
    const int Tb = B.tensor(L(idx0 + 4), cn4 + c4, 8);                  // [up(n4), L4]
    const int Tc = B.tensor(L(idx0 + 7), cd3 + cn4, 16);                // [down(P3), n4]
    const int Td = B.tensor(L(idx0 + 10), cd4 + ctop, 32);              // [down(P4), top]
    const View o4{Tb, cn4, c4}, o6{Ta, ctop, c6}, on4{Tc, cd3, cn4}, otop{Td, cd4, ctop};
    auto block = [&](int i, View in, View out, int cc, int reps, bool shortcut, bool c3k, double ee) {
        if (v8) B.c2f(L(i), in, out, cc, N(reps), shortcut, false, false);
        else B.c3k2(L(i), in, out, cc, N(reps), c3k, ee);
    };
    const int t0 = B.tensor(L(0), c0, 2);
    {
        Op o;
        o.kind = OP_STEM; o.name = L(0); o.out = B.full(t0); o.k = 3; o.s = 2; o.act = ACT_SILU;
        o.widx = B.weight(L(0), c0, 3, 3, 1, false, true);
        e.ops.push_back(o);
    }
    const int t1 = B.tensor(L(1), c1, 4);
    B.conv(L(1), B.full(t0), B.full(t1), 3, 2, ACT_SILU);
    const int t2 = B.tensor(L(2), c2, 4);
    block(2, B.full(t1), B.full(t2), c2, v8 ? 3 : 2, true, big, 0.25);
    const int t3 = B.tensor(L(3), c3, 8);
    B.conv(L(3), B.full(t2), B.full(t3), 3, 2, ACT_SILU);
    block(4, B.full(t3), o4, c4, v8 ? 6 : 2, true, big, 0.25);
    const int t5 = B.tensor(L(5), c5, 16);
    B.conv(L(5), o4, B.full(t5), 3, 2, ACT_SILU);
    block(6, B.full(t5), o6, c6, v8 ? 6 : 2, true, true, 0.5);
    const int t7 = B.tensor(L(7), c7, 32);
    B.conv(L(7), o6, B.full(t7), 3, 2, ACT_SILU);
    const int t8 = B.tensor(L(8), c8, 32);
    block(8, B.full(t7), B.full(t8), c8, v8 ? 3 : 2, true, true, 0.5);
    if (v8) B.sppf(L(9), B.full(t8), otop);
    else {
        const int t9 = B.tensor(L(9), c9, 32);
        B.sppf(L(9), B.full(t8), B.full(t9));
        B.c2psa(L(10), B.full(t9), otop, N(2));
    }
    B.upsample(L(idx0), otop, View{Ta, 0, ctop});
    block(idx0 + 2, B.full(Ta), on4, cn4, v8 ? 3 : 2, false, big, 0.5);
    B.upsample(L(idx0 + 3), on4, View{Tb, 0, cn4});
    const int tp3 = B.tensor(L(idx0 + 5), cn3, 8);
    block(idx0 + 5, B.full(Tb), B.full(tp3), cn3, v8 ? 3 : 2, false, big, 0.5);
    B.conv(L(idx0 + 6), B.full(tp3), View{Tc, 0, cd3}, 3, 2, ACT_SILU);
    const int tp4 = B.tensor(L(idx0 + 8), cn4b, 16);
    block(idx0 + 8, B.full(Tc), B.full(tp4), cn4b, v8 ? 3 : 2, false, big, 0.5);
    B.conv(L(idx0 + 9), B.full(tp4), View{Td, 0, cd4}, 3, 2, ACT_SILU);
    const int tp5 = B.tensor(L(idx0 + 11), cn5, 32);
    block(idx0 + 11, B.full(Td), B.full(tp5), cn5, v8 ? 3 : 2, false, true, 0.5);
    // (C2f shortcut of the v8 neck is False; C3k2 keeps its default shortcut = True everywhere: the yaml passes only c3k)

    // ---- Segment head -----------------------------------------------------------------------------------------------------
    const int nc = e.desc.nc;
    const std::string H = L(idx0 + 12);
    const View feat[3] = {B.full(tp3), B.full(tp4), B.full(tp5)};
    const int hc2 = std::max(std::max(16, feat[0].C / 4), 64);
    const int hc3 = std::max(feat[0].C, std::min(nc, 100));
    const int hc4 = std::max(feat[0].C / 4, YP_NM);
    const int npr = make_div8(std::min(256, sc.maxc) * sc.width);
    Op head;
    head.kind = OP_HEAD; head.name = H + ".postprocess"; head.nlev = 3; head.nms = true;
    for (int l = 0; l < 3; ++l) {
        const std::string Ls = std::to_string(l);
        const int sd = 8 << l;
        const View x = feat[l];
        const std::string pb = H + ".cv2." + Ls, pc = H + ".cv3." + Ls, pm = H + ".cv4." + Ls;
        const int b0 = B.tensor(pb + ".0", hc2, sd), b1 = B.tensor(pb + ".1", hc2, sd), b2 = B.tensor(pb + ".2", 64, sd, true);
        B.lane = 1 + 3 * l;
        B.conv(pb + ".0", x, B.full(b0), 3, 1, ACT_SILU);
        B.conv(pb + ".1", B.full(b0), B.full(b1), 3, 1, ACT_SILU);
        B.conv(pb + ".2", B.full(b1), B.full(b2), 1, 1, ACT_NONE);
        B.lane = 2 + 3 * l;
        const int k4 = B.tensor(pc + ".2", nc, sd, true);
        if (v8) {                                                       // legacy Detect: dense 3x3 -> 3x3 -> 1x1
            const int k0 = B.tensor(pc + ".0", hc3, sd), k1 = B.tensor(pc + ".1", hc3, sd);
            B.conv(pc + ".0", x, B.full(k0), 3, 1, ACT_SILU);
            B.conv(pc + ".1", B.full(k0), B.full(k1), 3, 1, ACT_SILU);
            B.conv(pc + ".2", B.full(k1), B.full(k4), 1, 1, ACT_NONE);
        } else {
            const int k0 = B.tensor(pc + ".0.0", x.C, sd), k1 = B.tensor(pc + ".0.1", hc3, sd), k2 = B.tensor(pc + ".1.0", hc3, sd), k3 = B.tensor(pc + ".1.1", hc3, sd);
            B.dwconv(pc + ".0.0", x, B.full(k0), 3, 1, ACT_SILU);
            B.conv(pc + ".0.1", B.full(k0), B.full(k1), 1, 1, ACT_SILU);
            B.dwconv(pc + ".1.0", B.full(k1), B.full(k2), 3, 1, ACT_SILU);
            B.conv(pc + ".1.1", B.full(k2), B.full(k3), 1, 1, ACT_SILU);
            B.conv(pc + ".2", B.full(k3), B.full(k4), 1, 1, ACT_NONE);
        }
        {
            const int am = B.tensor(H + ".amax." + Ls, 1, sd, true);
            Op o;
            o.lane = B.lane;
            o.kind = OP_AMAX; o.name = H + ".amax." + Ls; o.in = B.full(k4); o.out = B.full(am);
            e.ops.push_back(o);
            head.amax[l] = B.full(am);
        }
        const int m0 = B.tensor(pm + ".0", hc4, sd), m1 = B.tensor(pm + ".1", hc4, sd), m2 = B.tensor(pm + ".2", YP_NM, sd, true);
        B.lane = 3 + 3 * l;
        B.conv(pm + ".0", x, B.full(m0), 3, 1, ACT_SILU);
        B.conv(pm + ".1", B.full(m0), B.full(m1), 3, 1, ACT_SILU);
        B.conv(pm + ".2", B.full(m1), B.full(m2), 1, 1, ACT_NONE);
        head.box[l] = B.full(b2); head.cls[l] = B.full(k4); head.cf[l] = B.full(m2);
    }
    {
        const std::string pp = H + ".proto";
        const int p0 = B.tensor(pp + ".cv1", npr, 8), p1 = B.tensor(pp + ".upsample", npr, 4), p2 = B.tensor(pp + ".cv2", npr, 4), p3 = B.tensor(pp + ".cv3", YP_NM, 4);
        B.lane = 10;
        B.conv(pp + ".cv1", feat[0], B.full(p0), 3, 1, ACT_SILU);
        {
            Op o;
            o.lane = B.lane;
            o.kind = OP_CONVT; o.name = pp + ".upsample"; o.in = B.full(p0); o.out = B.full(p1); o.k = 2; o.s = 2; o.act = ACT_NONE;
            o.widx = B.weight(pp + ".upsample", npr, npr, 2, 1, true);
            e.ops.push_back(o);
        }
        B.conv(pp + ".cv2", B.full(p1), B.full(p2), 3, 1, ACT_SILU);
        B.conv(pp + ".cv3", B.full(p2), B.full(p3), 1, 1, ACT_SILU);
        e.proto_t = p3;
    }
    B.lane = 0;
    head.lane = 0;
    e.ops.push_back(head);
    return finish_graph_passes(e);
}

static int build_graph(yp_engine& e) {
    if (e.desc.family == YP_FAMILY_V8 || e.desc.family == YP_FAMILY_11) return build_graph_seg(e);
    if (e.desc.family != YP_FAMILY_V10) return fail(YP_ERR_ARG, "unknown model family %d", e.desc.family);
    Scale sc;
    if (!variant_scale(e.desc.variant, sc)) return fail(YP_ERR_ARG, "unknown variant '%c'", e.desc.variant);
    const int v = e.desc.variant;
    Builder B(e);
    auto C = [&](int c) { return make_div8(std::min(c, sc.maxc) * sc.width); };
    auto N = [&](int n) { return n > 1 ? std::max(py_round(n * sc.depth), 1) : n; };

    // resolved output channels of layers 0..22 (A.3)
    const int c0 = C(64), c1 = C(128), c2 = C(128), c3 = C(256), c4 = C(256), c5 = C(512), c6 = C(512), c7 = C(1024),
              c8 = C(1024), c9 = C(1024), c10 = C(1024), c13 = C(512), c16 = C(256), c17 = C(256), c19 = C(512),
              c20 = C(512), c22 = C(1024);
    // concat buffers, created up front so producers write straight into their slice
    const int T12 = B.tensor("model.12", c10 + c6, 16);   // [up(L10), L6]
    const int T15 = B.tensor("model.15", c13 + c4, 8);    // [up(L13), L4]
    const int T18 = B.tensor("model.18", c17 + c13, 16);  // [L17, L13]
    const int T21 = B.tensor("model.21", c20 + c10, 32);  // [L20, L10]
    const View o4{T15, c13, c4}, o6{T12, c10, c6}, o13{T18, c17, c13}, o10{T21, c20, c10};

    // L0 stem
    const int t0 = B.tensor("model.0", c0, 2);
    {
        Op o;
        o.kind = OP_STEM; o.name = "model.0"; o.out = B.full(t0); o.k = 3; o.s = 2; o.act = ACT_SILU;
        o.widx = B.weight("model.0", c0, 3, 3, 1, false, true);
        e.ops.push_back(o);
    }
    const int t1 = B.tensor("model.1", c1, 4);
    B.conv("model.1", B.full(t0), B.full(t1), 3, 2, ACT_SILU);
    const int t2 = B.tensor("model.2", c2, 4);
    B.c2f("model.2", B.full(t1), B.full(t2), c2, N(3), true, false, false);
    const int t3 = B.tensor("model.3", c3, 8);
    B.conv("model.3", B.full(t2), B.full(t3), 3, 2, ACT_SILU);
    B.c2f("model.4", B.full(t3), o4, c4, N(6), true, false, false);
    const int t5 = B.tensor("model.5", c5, 16);
    B.scdown("model.5", o4, B.full(t5));
    B.c2f("model.6", B.full(t5), o6, c6, N(6), true, v == 'x', false);
    const int t7 = B.tensor("model.7", c7, 32);
    B.scdown("model.7", o6, B.full(t7));
    const int t8 = B.tensor("model.8", c8, 32);
    B.c2f("model.8", B.full(t7), B.full(t8), c8, N(3), true, v != 'n', v == 's');
    const int t9 = B.tensor("model.9", c9, 32);
    B.sppf("model.9", B.full(t8), B.full(t9));
    B.psa("model.10", B.full(t9), o10);
    B.upsample("model.11", o10, View{T12, 0, c10});
    {
        const bool cib = !(v == 'n' || v == 's' || v == 'm');
        B.c2f("model.13", B.full(T12), o13, c13, N(3), cib, cib, false);
    }
    B.upsample("model.14", o13, View{T15, 0, c13});
    const int t16 = B.tensor("model.16", c16, 8);
    B.c2f("model.16", B.full(T15), B.full(t16), c16, N(3), false, false, false);
    B.conv("model.17", B.full(t16), View{T18, 0, c17}, 3, 2, ACT_SILU);
    const int t19 = B.tensor("model.19", c19, 16);
    {
        const bool cib = !(v == 'n' || v == 's');
        B.c2f("model.19", B.full(T18), B.full(t19), c19, N(3), cib, cib, false);
    }
    B.scdown("model.20", B.full(t19), View{T21, 0, c20});
    const int t22 = B.tensor("model.22", c22, 32);
    B.c2f("model.22", B.full(T21), B.full(t22), c22, N(3), true, true, (v == 'n' || v == 's'));

    // ---- v10Detect one-to-one head (A.4) (+ seg branches, A.7) ---------------------------------------
    const int nc = e.desc.nc;
    const View feat[3] = {B.full(t16), B.full(t19), B.full(t22)};
    const int hc2 = std::max(std::max(16, feat[0].C / 4), 64);
    const int hc3 = std::max(feat[0].C, std::min(nc, 100));
    const int hc4 = std::max(feat[0].C / 4, YP_NM);
    Op head;
    head.kind = OP_HEAD; head.name = "model.23.postprocess"; head.nlev = 3;
    for (int l = 0; l < 3; ++l) {
        const std::string L = std::to_string(l);
        const int sd = 8 << l;
        const View x = feat[l];
        const std::string pb = "model.23.one2one_cv2." + L, pc = "model.23.one2one_cv3." + L;
        // (the class branch comes first in the op list: its depthwise conv then sits right behind the level's last neck conv, whose 1x1 it
        // can join in pwsp_kernel whether the box branch is dense or not)
        const int k0 = B.tensor(pc + ".0.0", x.C, sd), k1 = B.tensor(pc + ".0.1", hc3, sd), k2 = B.tensor(pc + ".1.0", hc3, sd),
                  k3 = B.tensor(pc + ".1.1", hc3, sd), k4 = B.tensor(pc + ".2", nc, sd, true);
        B.lane = 2 + 3 * l;                                           // class branch
        B.dwconv(pc + ".0.0", x, B.full(k0), 3, 1, ACT_SILU);
        B.conv(pc + ".0.1", B.full(k0), B.full(k1), 1, 1, ACT_SILU);
        B.dwconv(pc + ".1.0", B.full(k1), B.full(k2), 3, 1, ACT_SILU);
        B.conv(pc + ".1.1", B.full(k2), B.full(k3), 1, 1, ACT_SILU);
        B.conv(pc + ".2", B.full(k3), B.full(k4), 1, 1, ACT_NONE);
        {   // class-max keys of this level, produced on the class lane right behind the logits
            const int am = B.tensor("model.23.amax." + L, 1, sd, true);
            Op o;
            o.lane = B.lane;
            o.kind = OP_AMAX; o.name = "model.23.amax." + L; o.in = B.full(k4); o.out = B.full(am);
            e.ops.push_back(o);
            head.amax[l] = B.full(am);
        }
        const int b0 = B.tensor(pb + ".0", hc2, sd), b1 = B.tensor(pb + ".1", hc2, sd), b2 = B.tensor(pb + ".2", 64, sd, true);
        B.lane = 1 + 3 * l;                                           // box branch of level l
        B.conv(pb + ".0", x, B.full(b0), 3, 1, ACT_SILU);
        B.conv(pb + ".1", B.full(b0), B.full(b1), 3, 1, ACT_SILU);
        B.conv(pb + ".2", B.full(b1), B.full(b2), 1, 1, ACT_NONE);
        for (int j = 0; j < 3; ++j) { head.hb_box[l][j] = (int)e.ops.size() - 3 + j; head.hb_cf[l][j] = -1; }
        head.box[l] = B.full(b2);
        head.cls[l] = B.full(k4);
        if (e.desc.task == YP_TASK_SEGMENT) {
            const std::string pm = "model.23.cv4." + L;
            const int m0 = B.tensor(pm + ".0", hc4, sd), m1 = B.tensor(pm + ".1", hc4, sd), m2 = B.tensor(pm + ".2", YP_NM, sd, true);
            B.lane = 3 + 3 * l;                                       // mask-coefficient branch
            B.conv(pm + ".0", x, B.full(m0), 3, 1, ACT_SILU);
            B.conv(pm + ".1", B.full(m0), B.full(m1), 3, 1, ACT_SILU);
            B.conv(pm + ".2", B.full(m1), B.full(m2), 1, 1, ACT_NONE);
            for (int j = 0; j < 3; ++j) head.hb_cf[l][j] = (int)e.ops.size() - 3 + j;
            head.cf[l] = B.full(m2);
        }
    }
    if (e.desc.task == YP_TASK_SEGMENT) {
        const int npr = feat[0].C;
        const std::string pp = "model.23.proto";
        const int p0 = B.tensor(pp + ".cv1", npr, 8), p1 = B.tensor(pp + ".upsample", npr, 4), p2 = B.tensor(pp + ".cv2", npr, 4),
                  p3 = B.tensor(pp + ".cv3", YP_NM, 4);
        B.lane = 10;                                                  // prototype branch
        B.conv(pp + ".cv1", feat[0], B.full(p0), 3, 1, ACT_SILU);
        {
            Op o;
            o.lane = B.lane;
            o.kind = OP_CONVT; o.name = pp + ".upsample"; o.in = B.full(p0); o.out = B.full(p1); o.k = 2; o.s = 2; o.act = ACT_NONE;
            o.widx = B.weight(pp + ".upsample", npr, npr, 2, 1, true);
            e.ops.push_back(o);
        }
        B.conv(pp + ".cv2", B.full(p1), B.full(p2), 3, 1, ACT_SILU);
        B.conv(pp + ".cv3", B.full(p2), B.full(p3), 1, 1, ACT_SILU);
        e.proto_t = p3;
    }
    B.lane = 0;
    head.lane = 0;
    e.ops.push_back(head);
    return finish_graph_passes(e);
}

// graph passes shared by every family: they only look at op kinds, shapes and who reads what
static int finish_graph_passes(yp_engine& e) {
    // ---- fusion pass: depthwise 3x3 (s1, SiLU) whose only consumer is the next op, a 1x1 conv ------------------------------
    for (size_t i = 0; i + 1 < e.ops.size(); ++i) {
        const Op& d = e.ops[i];
        Op& c = e.ops[i + 1];
        if (d.kind != OP_DWCONV || d.k != 3 || d.s != 1 || d.res.t >= 0 || d.gs != 0) continue;
        if (c.kind != OP_CONV || c.k != 1 || c.s != 1 || c.res.t >= 0) continue;
        if (c.in.t != d.out.t || c.in.coff != d.out.coff || c.in.C != d.out.C) continue;
        bool other_reader = false;
        for (size_t j = 0; j < e.ops.size(); ++j) {
            if (j == i + 1) continue;
            const Op& q = e.ops[j];
            for (const View* v : {&q.in, &q.res})
                if (v->t == d.out.t && v->coff < d.out.coff + d.out.C && d.out.coff < v->coff + v->C) other_reader = true;
        }
        if (!other_reader) c.fuse_dw = (int)i;
    }
    // ---- fold pass: nearest-x2 upsample written into the leading slice of a concat buffer whose only reader is a 1x1 conv ----
    for (size_t i = 0; i < e.ops.size(); ++i) {
        const Op& u = e.ops[i];
        if (u.kind != OP_UPSAMPLE || (u.out.C % 64) != 0) continue;
        int reader = -1, nread = 0;
        for (size_t j = 0; j < e.ops.size(); ++j) {
            const Op& q = e.ops[j];
            for (const View* v : {&q.in, &q.res})
                if (v->t == u.out.t && v->coff < u.out.coff + u.out.C && u.out.coff < v->coff + v->C) { reader = (int)j; ++nread; }
        }
        if (nread != 1) continue;
        Op& c = e.ops[reader];
        if (c.kind != OP_CONV || c.k != 1 || c.s != 1 || c.in.t != u.out.t || c.in.coff != u.out.coff || c.in.C <= u.out.C) continue;
        c.fold_up = (int)i;
    }
    // ---- 3x3 stride-2 conv whose only consumer is the next op, a 1x1 conv: candidates for conv_halo_s2's fused trailing 1x1 -----
    for (size_t i = 0; i + 1 < e.ops.size(); ++i) {
        const Op& a = e.ops[i];
        Op& c = e.ops[i + 1];
        if (a.kind != OP_CONV || a.k != 3 || a.s != 2 || a.res.t >= 0) continue;
        if (c.kind != OP_CONV || c.k != 1 || c.s != 1 || c.res.t >= 0 || c.fold_up >= 0 || c.fuse_dw >= 0) continue;
        if (c.in.t != a.out.t || c.in.coff != a.out.coff || c.in.C != a.out.C) continue;
        bool other_reader = false;
        for (size_t j = 0; j < e.ops.size(); ++j) {
            if (j == i + 1) continue;
            const Op& q = e.ops[j];
            for (const View* v : {&q.in, &q.res})
                if (v->t == a.out.t && v->coff < a.out.coff + a.out.C && a.out.coff < v->coff + v->C) other_reader = true;
        }
        if (!other_reader) c.fuse_pre = (int)i;
    }
    // ---- 1x1 logit conv (no activation, fp32 output) behind a dw -> pw pair, optionally followed by the class-max op: candidates for the
    //      third stage of conv_dwpw's TAIL form (the class branch of the v10 / YOLO11 heads: dw -> pw -> dw -> [pw -> 1x1 logits -> max]) -----
    for (size_t i = 1; i < e.ops.size(); ++i) {
        Op& c2 = e.ops[i];
        const Op& c1 = e.ops[i - 1];
        if (c2.kind != OP_CONV || c2.k != 1 || c2.s != 1 || c2.res.t >= 0 || c2.act != ACT_NONE || c2.fold_up >= 0 || c2.fuse_dw >= 0 || c2.fuse_pre >= 0) continue;
        if (!e.tensors[c2.out.t].f32 || c2.out.coff != 0 || c2.out.C != e.tensors[c2.out.t].C) continue;
        if (c1.kind != OP_CONV || c1.fuse_dw < 0 || c1.res.t >= 0) continue;
        if (c2.in.t != c1.out.t || c2.in.coff != c1.out.coff || c2.in.C != c1.out.C) continue;
        bool other_reader = false;
        for (size_t j = 0; j < e.ops.size(); ++j) {
            if (j == i) continue;
            const Op& q = e.ops[j];
            for (const View* v : {&q.in, &q.res})
                if (v->t == c1.out.t && v->coff < c1.out.coff + c1.out.C && c1.out.coff < v->coff + v->C) other_reader = true;
        }
        if (other_reader) continue;
        c2.fuse_tail = (int)i - 1;
        if (i + 1 < e.ops.size() && e.ops[i + 1].kind == OP_AMAX && e.ops[i + 1].in.t == c2.out.t) c2.tail_amax = (int)i + 1;
    }
    // ---- the 1x1 that writes a level's fp32 class logits + the class-max op that reads them: candidates for cls_out_kernel -----------------
    for (size_t i = 0; i < e.ops.size(); ++i) {
        const Op& m = e.ops[i];
        if (m.kind != OP_AMAX) continue;
        for (size_t j = 0; j < i; ++j) {
            Op& c = e.ops[j];
            if (c.kind == OP_CONV && c.k == 1 && c.s == 1 && c.act == ACT_NONE && c.res.t < 0 && c.fold_up < 0 && c.out.t == m.in.t && c.out.coff == 0 &&
                c.out.C == e.tensors[c.out.t].C && e.tensors[c.out.t].f32 && m.in.coff == 0 && m.in.C == c.out.C)
                c.amax_post = (int)i;
        }
    }
    // ---- 1x1 conv -> depthwise 3x3 / 7x7 (stride 1) or SPPF's pool chain on (a channel sub-range of) its output: candidates for pwsp_kernel,
    //      one workgroup per (image, channel slice) - the small-map layers (CIB, SPPF, the P5 class branch) --------------------------------
    for (size_t i = 0; i < e.ops.size(); ++i) {
        Op& d = e.ops[i];
        const bool dw = d.kind == OP_DWCONV && (d.k == 3 || d.k == 7) && d.s == 1 && d.gs == 0;
        if (!dw && d.kind != OP_POOL3) continue;
        int c = -1;
        for (size_t j = i; j-- > 0;) {                               // the latest earlier op that writes what d reads
            const Op& q = e.ops[j];
            if (q.out.t == d.in.t && q.out.coff < d.in.coff + d.in.C && d.in.coff < q.out.coff + q.out.C) { c = (int)j; break; }
        }
        if (c < 0) continue;
        const Op& pw = e.ops[c];
        if (pw.kind != OP_CONV || pw.k != 1 || pw.s != 1 || pw.res.t >= 0 || pw.fold_up >= 0) continue;
        if (d.in.coff < pw.out.coff || d.in.coff + d.in.C > pw.out.coff + pw.out.C) continue;      // every channel d reads comes from this conv
        if (e.tensors[pw.out.t].f32 || e.tensors[d.out.t].f32) continue;
        d.pw_pre = c;
    }
    return YP_OK;
}

// ---------------------------------------------------------------------------------------------------------
// plan: resolve shapes for (B,H,W), compute algorithmic flops/bytes
// ---------------------------------------------------------------------------------------------------------
static ConvParams conv_params(const yp_engine& e, const Op& o);
// Releases the replay executable and the graph it was instantiated from. Callers have made sure no replay of it is in flight
// (ev_done synchronised, or a device synchronisation): see the lifetime rule at forward_replay.
static void drop_graph(yp_engine& e) {
    if (e.gexec) { (void)hipGraphExecDestroy(e.gexec); e.gexec = nullptr; }
    if (e.gsrc) { (void)hipGraphDestroy(e.gsrc); e.gsrc = nullptr; }
}
static DwPwParams dwpw_params(const yp_engine& e, const Op& c);
static FrontParams front_params(const yp_engine& e, const Op& o, const uint8_t* img);
static C2fParams c2f_params(const yp_engine& e, const Op& o);
static ScdParams scd_params(const yp_engine& e, const Op& o);
static PwSpParams pwsp_params(const yp_engine& e, const Op& o);
static ClsOutParams cls_out_params(const yp_engine& e, const Op& o);
static bool views_overlap(const View& a, const View& b);
static size_t tensor_elem_bytes(const yp_engine& e, const TensorDesc& t) { return (t.f32 || e.dtype == DT_F32) ? 4 : 2; }

// ---- winners-only head (head_branch.hip): workspace layout and parameter blocks -----------------------------------------------------------------
struct SparseWs {
    size_t sel, wlist, wcount, thr, box, cf, pcount, plist, t0box, t0cf, total;
    int plist_off[3], plist_cap[3];          // per level: first entry / capacity of its position list
    size_t t0_off[3];                        // per level: first (image, pixel) row of the position-addressed maps, in rows
    size_t rows;                             // B * anchors
};
static SparseWs sparse_ws_layout(int B, int max_det, const int (&HWl)[3]) {
    SparseWs w{};
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    w.sel = take((size_t)B * HEAD_MAXK * 4); w.wlist = take((size_t)B * 3 * HEAD_MAXK * 4); w.wcount = take((size_t)B * 3 * 4); w.thr = take((size_t)B * 4);
    w.box = take((size_t)B * max_det * 64 * 4); w.cf = take((size_t)B * max_det * 32 * 4);
    w.pcount = take(8 * 4);                  // [0..3) live counters, [4..7) the counts of the last forward (head stage 2 saves them)
    size_t ents = 0, rows = 0;
    for (int l = 0; l < 3; ++l) {
        w.plist_off[l] = (int)ents; w.plist_cap[l] = B * std::min(HWl[l], 9 * max_det);   // a winner's 3x3 neighbourhood, never more than the level has
        ents += (size_t)(w.plist_cap[l] + 63) & ~(size_t)63;
        w.t0_off[l] = rows; rows += (size_t)B * HWl[l];
    }
    w.rows = rows;
    w.plist = take(ents * 4);
    w.t0box = take(rows * 64 * 2); w.t0cf = take(rows * 32 * 2);
    w.total = off;
    return w;
}
static SparseWs sparse_ws_layout(const yp_engine& e) {
    const int hw[3] = {(e.pH / 8) * (e.pW / 8), (e.pH / 16) * (e.pW / 16), (e.pH / 32) * (e.pW / 32)};
    return sparse_ws_layout(e.pB, e.desc.max_det, hw);
}
// which: 0 = box branch (one2one_cv2), 1 = mask-coefficient branch (cv4)
static HeadBranchParams head_branch_params(const yp_engine& e, const Op& h, int which) {
    HeadBranchParams p{};
    const SparseWs ws = sparse_ws_layout(e);
    char* base = (char*)e.sp_ws;
    for (int l = 0; l < 3; ++l) {
        const int* ids = which == 0 ? h.hb_box[l] : h.hb_cf[l];
        if (ids[0] < 0) { p.cmid = 0; return p; }
        const Op &c0 = e.ops[ids[0]], &c1 = e.ops[ids[1]], &c2 = e.ops[ids[2]];
        const WeightDesc &w0 = e.weights[c0.widx], &w1 = e.weights[c1.widx], &w2 = e.weights[c2.widx];
        const TensorDesc& ti = e.tensors[c0.in.t];
        p.x[l] = ti.ptr; p.x_stride[l] = ti.C; p.x_coff[l] = c0.in.coff; p.H[l] = ti.H; p.W[l] = ti.W; p.Cin[l] = c0.in.C; p.x_bytes[l] = ti.bytes;
        p.w0[l] = w0.d_w; p.Kpad0[l] = w0.Kpad; p.b0[l] = w0.d_b;
        p.w1[l] = w1.d_w; p.Kpad1[l] = w1.Kpad; p.b1[l] = w1.d_b;
        p.w2[l] = w2.d_w; p.Kpad2[l] = w2.Kpad; p.b2[l] = w2.d_b;
        if (l == 0) { p.cmid = c0.out.C; p.cout = c2.out.C; p.act0 = c0.act; p.act1 = c1.act; }
        // the shape the kernel is written for: 3x3 s1 -> 3x3 s1 -> 1x1 without activation, no residuals, equal widths on every level
        if (c0.k != 3 || c1.k != 3 || c2.k != 1 || c0.s != 1 || c1.s != 1 || c2.s != 1 || c0.res.t >= 0 || c1.res.t >= 0 || c2.res.t >= 0 || c2.act != ACT_NONE ||
            c0.out.C != p.cmid || c1.out.C != p.cmid || c1.in.C != p.cmid || c2.in.C != p.cmid || c2.out.C != p.cout || c0.act != p.act0 || c1.act != p.act1 ||
            w0.cin_pad != c0.in.C || w1.cin_pad != p.cmid) { p.cmid = 0; return p; }
    }
    p.B = e.pB; p.max_det = e.desc.max_det; p.maxk = HEAD_MAXK;
    p.A0 = p.H[0] * p.W[0]; p.A1 = p.H[1] * p.W[1];
    auto at = [&](size_t off) -> char* { return base ? base + off : nullptr; };      // (plan time: the workspace does not exist yet)
    p.sel = (const int*)at(ws.sel); p.wlist = (const int*)at(ws.wlist); p.wcount = (const int*)at(ws.wcount);
    p.out = (float*)at(which == 0 ? ws.box : ws.cf);
    p.plist = (const int*)at(ws.plist); p.pcount = (const int*)at(ws.pcount);
    p.t0 = at(which == 0 ? ws.t0box : ws.t0cf); p.t0_bytes = ws.rows * (size_t)p.cmid * 2;
    for (int l = 0; l < 3; ++l) { p.plist_off[l] = ws.plist_off[l]; p.plist_cap[l] = ws.plist_cap[l]; p.t0_off[l] = ws.t0_off[l] * (size_t)p.cmid; }
    p.pos_grid = 256;                        // one workgroup per CU (its three plane slots fill most of a CU's LDS)
    return p;
}

static int make_plan(yp_engine& e, int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0 || (H % 32) || (W % 32)) return fail(YP_ERR_ARG, "input must be [B,H,W,3] with H,W multiples of 32 (got %d,%d,%d)", B, H, W);
    if (e.planned && e.pB == B && e.pH == H && e.pW == W) return YP_OK;
    size_t A = 0;
    for (int l = 0; l < 3; ++l) A += (size_t)(H / (8 << l)) * (W / (8 << l));
    if (A > 12288) return fail(YP_ERR_ARG, "input %dx%d has %zu anchors; the LDS top-k supports at most 12288", H, W, A);
    for (auto& t : e.tensors) {
        t.H = H / t.sdiv; t.W = W / t.sdiv;
        t.bytes = (size_t)B * t.H * t.W * t.C * tensor_elem_bytes(e, t);
    }
    for (auto& o : e.ops) {
        const double es = e.es();
        auto vbytes = [&](const View& v) {
            if (v.t < 0) return 0.0;
            const TensorDesc& t = e.tensors[v.t];
            return (double)B * t.H * t.W * v.C * (double)tensor_elem_bytes(e, t);
        };
        o.flops = 0; o.bytes = vbytes(o.in) + vbytes(o.out) + vbytes(o.res);
        if (o.kind == OP_CONV || o.kind == OP_DWCONV || o.kind == OP_STEM || o.kind == OP_CONVT) {
            const TensorDesc& to = e.tensors[o.out.t];
            const WeightDesc& w = e.weights[o.widx];
            const double px = (double)B * to.H * to.W;
            if (o.kind == OP_CONVT) o.flops = 2.0 * px * w.cout * w.cin_g;   // one tap per output pixel
            else o.flops = 2.0 * px * w.cout * w.cin_g * w.k * w.k;
            o.bytes += (double)w.cout * w.cin_g * w.k * w.k * es;
            if (o.kind == OP_STEM) o.bytes += (double)B * H * W * 3;
        } else if (o.kind == OP_ATTN) {
            const TensorDesc& ti = e.tensors[o.in.t];
            const double Nn = (double)ti.H * ti.W;
            o.flops = 2.0 * B * o.nh * Nn * Nn * (o.kd + o.hd);
        } else if (o.kind == OP_HEAD) {
            o.bytes = 0;                       // class-max keys + the winners' class / box rows + the outputs
            for (int l = 0; l < 3; ++l) o.bytes += (o.amax[l].t >= 0) ? vbytes(o.amax[l]) : vbytes(o.cls[l]);
            o.bytes += (double)B * e.desc.max_det * (e.desc.nc + 64 + 6 + 1) * 4;
        }
    }
    e.pB = B; e.pH = H; e.pW = W; e.planned = true; e.allocated = false; e.warmed = false;
    for (auto& o : e.ops) o.cfg = -1;
    static const char* kn[] = {"stem_kernel", "", "dwconv_kernel", "pool5_kernel", "upsample2_kernel", "attention_kernel", "head_select_kernel", "", "sppf_pool3_kernel", "anchor_max_level_kernel"};
    for (auto& o : e.ops) { o.fused = false; o.skip = false; o.folded = false; o.fused2 = false; o.fused3 = false; o.fused4 = false; o.fused5 = false; o.fused6 = false; o.fused7 = false; o.fused8 = false; o.pw_store = false; o.sparse_box = false; o.sparse_cf = false; }
    static const bool no_fold = [] { const char* v = std::getenv("YOLOP_NO_FOLD"); return v && *v == '1'; }();   // A/B switch
    for (auto& o : e.ops) {
        if (o.kind != OP_CONV || o.fold_up < 0 || e.dtype != DT_BF16 || no_fold) continue;
        o.folded = true;                                   // tentatively, so that conv_params describes the folded form
        const ConvParams q = conv_params(e, o);
        bool any = false;
        for (int c = 0; c < conv_dma_p_num_cfgs() && !any; ++c) any = conv_dma_p_cfg_valid(q, c);
        if (any) e.ops[o.fold_up].skip = true;
        else o.folded = false;
    }
    for (auto& o : e.ops) {
        if (o.kind == OP_CONV && o.fuse_pre >= 0 && e.dtype == DT_BF16 && e.fuse) {
            o.fused2 = true;
            const int c = conv_halo_s2_pw_cfg(conv_params(e, o));
            if (c >= 0) {
                e.ops[o.fuse_pre].skip = true; o.cfg = 500 + c; o.kernel = conv_halo_s2_pw_kernel_name(c);
                // ... and with the stem in front of it, when the stem's output has no other reader
                static const bool no_front = [] { const char* v = std::getenv("YOLOP_NO_FRONT"); return v && *v == '1'; }();   // A/B switch
                const Op& c1 = e.ops[o.fuse_pre];
                int stem = -1, readers = 0;
                for (size_t j = 0; j < e.ops.size(); ++j) {
                    if (e.ops[j].kind == OP_STEM && e.ops[j].out.t == c1.in.t) stem = (int)j;
                    for (const View* v : {&e.ops[j].in, &e.ops[j].res}) if (v->t == c1.in.t) ++readers;
                }
                if (stem >= 0 && readers == 1 && !no_front && e.weights[e.ops[stem].widx].d_w2 != nullptr) {
                    o.stem_op = stem;
                    if (frontend_valid(front_params(e, o, nullptr))) { o.fused3 = true; e.ops[stem].skip = true; o.kernel = "frontend_kernel"; }
                }
                continue;
            }
            o.fused2 = false;
        }
        static const bool no_c2f = [] { const char* v = std::getenv("YOLOP_NO_C2F"); return v && *v == '1'; }();   // A/B switch
        if (o.kind == OP_CONV && o.c2f_m1 >= 0 && e.dtype == DT_BF16 && e.fuse && !no_c2f && c2f_fused_valid(c2f_params(e, o))) {
            o.fused4 = true; e.ops[o.c2f_m1].skip = true; e.ops[o.c2f_m2].skip = true; o.kernel = "c2f_fused_kernel";
            continue;
        }
        if (o.kind == OP_CONV && o.fuse_dw >= 0 && e.dtype == DT_BF16 && e.fuse) {
            const DwPwParams q = dwpw_params(e, o);
            if (conv_dwpw_valid(q)) { o.fused = true; e.ops[o.fuse_dw].skip = true; o.kernel = conv_dwpw_kernel_name(q); continue; }
        }
        // TAIL form (logit conv + class-max keys as a third stage of the last dw -> pw pair): opt-in, YOLOP_TAIL=1. Alone it takes 33 us less
        // than the three launches it replaces (213 -> 180 us over the P3 / P4 class branches, -105 MB of HBM traffic), but in the replayed
        // graph the step is 0.7 % SLOWER with it (1.936 vs 1.922 ms, same box, three alternating runs): the 1x1 conv and the max pass it
        // removes were HBM-bound and ran beside the VALU-bound kernels of the other head lanes for free, while the longer fused kernel holds
        // its statically assigned CUs for longer.
        if (o.kind == OP_CONV && o.fuse_tail >= 0 && e.dtype == DT_BF16 && e.fuse && e.tail && e.ops[o.fuse_tail].fused) {
            // (ops are visited in order: the pointwise conv in front has already been decided)
            o.fused6 = true;
            const DwPwParams q = dwpw_params(e, o);
            if (conv_dwpw_valid(q)) {
                e.ops[o.fuse_tail].skip = true;
                if (o.tail_amax >= 0) e.ops[o.tail_amax].skip = true;
                o.kernel = conv_dwpw_kernel_name(q);
                continue;
            }
            o.fused6 = false;
        }
        if (o.kind == OP_CONV) o.kernel = conv_kernel_name(conv_params(e, o), e.dtype);
        else if (o.kind == OP_CONVT) {
            ConvParams p{};
            p.Cout = o.out.C; p.M = B * e.tensors[o.in.t].H * e.tensors[o.in.t].W; p.Cin = o.in.C; p.ks = 1;
            p.Kpad = (o.in.C + 31) / 32 * 32; p.cfg = o.cfg;
            o.kernel = conv_kernel_name(p, e.dtype);
        } else if (o.kind == OP_DWCONV) {
            static const bool no_scd = [] { const char* v = std::getenv("YOLOP_NO_SCD"); return v && *v == '1'; }();   // A/B switch
            if (o.scd_pre >= 0 && e.dtype == DT_BF16 && e.fuse && !no_scd && scdown_fused_valid(scd_params(e, o))) {
                o.fused5 = true; e.ops[o.scd_pre].skip = true; o.kernel = scdown_fused_kernel_name(scd_params(e, o));
                continue;
            }
            const char* t = e.dtype == DT_BF16 ? "bf16" : "f32";
            char buf[64];
            DwParams q{};
            q.H = q.Ho = e.tensors[o.in.t].H; q.W = q.Wo = e.tensors[o.in.t].W; q.C = o.out.C; q.ks = o.k; q.stride = o.s; q.gs = o.gs;
            q.x_stride = e.tensors[o.in.t].C; q.x_coff = o.in.coff; q.y_stride = e.tensors[o.out.t].C; q.y_coff = o.out.coff;
            q.res = o.res.t >= 0 ? (const void*)1 : nullptr;
            q.x_bytes = (size_t)B * q.H * q.W * q.x_stride * 2;
            if (dwconv_mfma_valid(q, e.dtype)) snprintf(buf, sizeof(buf), "dwconv_mfma_kernel<%d,%s>", o.k, B * (q.C / 32) >= 256 ? "false" : "true");
            else if (o.k == 3 && o.s == 1) snprintf(buf, sizeof(buf), "dwconv_row_kernel<%s,3,1,4>", t);
            else if (o.k == 3 && o.s == 2) snprintf(buf, sizeof(buf), "dwconv_row_kernel<%s,3,2,2>", t);
            else if (o.k == 7 && o.s == 1) snprintf(buf, sizeof(buf), "dwconv_row_kernel<%s,7,1,2>", t);
            else snprintf(buf, sizeof(buf), "dwconv_kernel<%s>", t);
            o.kernel = buf;
        } else if (o.kind == OP_STEM) {
            char buf[64];
            if (e.dtype == DT_BF16) snprintf(buf, sizeof(buf), "stem_mfma_kernel<%d>", o.out.C / 16);
            else snprintf(buf, sizeof(buf), "stem_kernel<f32>");
            o.kernel = buf;
        } else if (o.kind == OP_POOL3 && e.dtype == DT_BF16 && (o.in.C & 31) == 0) o.kernel = "sppf_pool3_bf16_kernel";
        else o.kernel = (o.kind == OP_HEAD && o.nms) ? "head_nms_kernel" : kn[o.kind];
    }
    // winners-only head (v10 top-k head, bf16): the box / coefficient branches run on the stage-1 winners inside the head op; their dense
    // convolutions stay in the op list (yp_run_op steps them, the fp32 parity mode and YOLOP_DENSE_HEAD=1 run them) but launch nothing here
    for (auto& o : e.ops) {
        if (o.kind != OP_HEAD || o.nms || e.dtype != DT_BF16 || !e.sparse_head || e.desc.max_det > HEAD_MAXK || o.amax[0].t < 0) continue;
        for (int which = 0; which < 2; ++which) {
            const HeadBranchParams q = head_branch_params(e, o, which);
            if (q.cmid == 0 || !head_branch_valid(q)) continue;
            if (which == 1 && !o.sparse_box) continue;              // (stage 2 reads the coefficient rows by rank only beside winners-only box rows)
            // the branch's tensors must have no reader outside the branch and the head op (they are never written in this mode)
            bool outside = false;
            for (int l = 0; l < 3 && !outside; ++l)
                for (int j = 0; j < 3 && !outside; ++j) {
                    const int oi = (which == 0 ? o.hb_box : o.hb_cf)[l][j];
                    const View& w = e.ops[oi].out;
                    for (size_t r = 0; r < e.ops.size() && !outside; ++r) {
                        const Op& q2 = e.ops[r];
                        if (&q2 == &o) continue;
                        bool member = false;
                        for (int l2 = 0; l2 < 3; ++l2) for (int j2 = 0; j2 < 3; ++j2) member |= (int)r == (which == 0 ? o.hb_box : o.hb_cf)[l2][j2];
                        if (member) continue;
                        for (const View* v : {&q2.in, &q2.res}) outside |= v->t >= 0 && v->t == w.t && v->coff < w.coff + w.C && w.coff < v->coff + v->C;
                    }
                }
            if (outside) continue;
            (which == 0 ? o.sparse_box : o.sparse_cf) = true;
            for (int l = 0; l < 3; ++l)
                for (int j = 0; j < 3; ++j) {
                    Op& c = e.ops[(which == 0 ? o.hb_box : o.hb_cf)[l][j]];
                    c.skip = true;
                }
        }
    }
    // 1x1 -> depthwise / pool chain as pwsp_kernel (decided last: it looks at which ops still launch). The 1x1 conv is skipped, the spatial
    // op launches the kernel; the conv's own output is written as well when anything else reads it. Not when an op that still launches
    // sits between the two and reads the conv's output (the kernel runs at the SPATIAL op's place in the order).
    {
        static const bool no_pwsp = [] { const char* v = std::getenv("YOLOP_NO_PWSP"); return v && *v == '1'; }();   // A/B switch
        for (size_t i = 0; i < e.ops.size(); ++i) {
            Op& d = e.ops[i];
            if (d.pw_pre < 0 || e.dtype != DT_BF16 || !e.fuse || no_pwsp || d.skip || d.fused5) continue;
            Op& c = e.ops[d.pw_pre];
            if (c.skip || c.fused || c.fused2 || c.fused3 || c.fused4 || c.fused6 || c.folded) continue;
            bool between = false, other = false;
            for (size_t j = 0; j < e.ops.size(); ++j) {
                const Op& q = e.ops[j];
                if (j == i || (int)j == d.pw_pre || q.skip) continue;
                bool reads = false;
                for (const View* v : {&q.in, &q.res}) reads |= views_overlap(*v, c.out);
                if (q.kind == OP_HEAD)
                    for (int l = 0; l < 3; ++l) {
                        for (const View* v : {&q.box[l], &q.cls[l], &q.cf[l], &q.amax[l]}) reads |= views_overlap(*v, c.out);
                        for (int which = 0; which < 2; ++which) {
                            const int b0 = (which == 0 ? q.hb_box : q.hb_cf)[l][0];
                            if (b0 >= 0 && (which == 0 ? q.sparse_box : q.sparse_cf)) reads |= views_overlap(e.ops[b0].in, c.out);
                        }
                    }
                other |= reads;
                if ((int)j > d.pw_pre && j < i) {
                    between |= reads || views_overlap(q.out, c.out) || views_overlap(q.out, d.out) || views_overlap(q.out, d.res) || views_overlap(q.in, d.out) ||
                               views_overlap(q.res, d.out);
                }
            }
            if (between) continue;
            d.fused7 = true; d.pw_store = other;
            if (!pwsp_valid(pwsp_params(e, d))) { d.fused7 = false; d.pw_store = false; continue; }
            c.skip = true;
            d.kernel = pwsp_kernel_name(pwsp_params(e, d));
        }
    }
    // class logits + class-max keys in one launch (cls_out_kernel): the OP_AMAX op is skipped
    {
        static const bool no_co = [] { const char* v = std::getenv("YOLOP_NO_CLSOUT"); return v && *v == '1'; }();   // A/B switch
        for (auto& o : e.ops) {
            if (o.kind != OP_CONV || o.amax_post < 0 || e.dtype != DT_BF16 || !e.fuse || no_co) continue;
            if (o.skip || o.fused || o.fused2 || o.fused3 || o.fused4 || o.fused6 || o.folded || e.ops[o.amax_post].skip) continue;
            if (!cls_out_valid(cls_out_params(e, o))) continue;
            o.fused8 = true; e.ops[o.amax_post].skip = true;
            o.kernel = cls_out_kernel_name(cls_out_params(e, o));
        }
    }
    // algorithmic work of the graph as it runs: an op whose work moved into a fused consumer reports nothing and launches
    // nothing; the consumer reports the FLOPs of all its stages and the bytes of what it reads and writes (the intermediates
    // never reach HBM), a conv with a folded upsample reads the low-resolution tensor instead of its upsampled copy
    {
        const double es = e.es();
        auto vb = [&](const View& v) {
            if (v.t < 0) return 0.0;
            const TensorDesc& t = e.tensors[v.t];
            return (double)B * t.H * t.W * v.C * (double)tensor_elem_bytes(e, t);
        };
        auto wb = [&](const Op& q) { const WeightDesc& w = e.weights[q.widx]; return (double)w.cout * w.cin_g * w.k * w.k * es; };
        for (auto& o : e.ops) {
            if (o.kind == OP_DWCONV && o.fused5) { const Op& c1 = e.ops[o.scd_pre]; o.flops += c1.flops; o.bytes = vb(c1.in) + vb(o.out) + wb(c1) + wb(o); }
            if (o.fused7) {
                const Op& c1 = e.ops[o.pw_pre];
                o.flops += c1.flops;
                o.bytes = vb(c1.in) + (o.pw_store ? vb(c1.out) : 0.0) + vb(o.out) + vb(o.res) + wb(c1) + (o.kind == OP_DWCONV ? wb(o) : 0.0);
            }
            if (o.kind != OP_CONV) continue;
            if (o.fused) { const Op& d = e.ops[o.fuse_dw]; o.flops += d.flops; o.bytes = vb(d.in) + vb(d.res) + vb(o.out) + wb(d) + wb(o); }
            else if (o.fused6) {                    // (its pointwise conv, visited before, already carries the depthwise stage's FLOPs)
                const Op& c1 = e.ops[o.fuse_tail];
                const Op& d = e.ops[c1.fuse_dw];
                o.flops += c1.flops;
                o.bytes = vb(d.in) + vb(o.out) + wb(d) + wb(c1) + wb(o) + (o.tail_amax >= 0 ? vb(e.ops[o.tail_amax].out) : 0.0);
            }
            else if (o.fused3) {
                const Op &c1 = e.ops[o.fuse_pre], &st = e.ops[o.stem_op];
                o.flops += c1.flops + st.flops; o.bytes = (double)B * H * W * 3 + vb(o.out) + wb(st) + wb(c1) + wb(o);
            } else if (o.fused2) { const Op& c1 = e.ops[o.fuse_pre]; o.flops += c1.flops; o.bytes = vb(c1.in) + vb(o.out) + wb(c1) + wb(o); }
            else if (o.fused4) {
                const Op &m1 = e.ops[o.c2f_m1], &m2 = e.ops[o.c2f_m2];
                o.flops += m1.flops + m2.flops;
                o.bytes = vb(View{o.in.t, o.in.coff, 2 * m1.in.C}) + vb(o.out) + wb(m1) + wb(m2) + wb(o);
            }
            if (o.folded) { const Op& u = e.ops[o.fold_up]; o.bytes += vb(u.in) - vb(u.out); }
            if (o.fused8) o.bytes += vb(e.ops[o.amax_post].out);
        }
        for (auto& o : e.ops) {
            if (o.kind != OP_HEAD) continue;
            for (int which = 0; which < 2; ++which) {
                if (!(which == 0 ? o.sparse_box : o.sparse_cf)) continue;
                const HeadBranchParams q = head_branch_params(e, o, which);
                for (int l = 0; l < 3; ++l) {        // (which level a winner lies on is data: a third each, neighbourhoods disjoint unless the level is full)
                    const double nw = (double)B * e.desc.max_det / 3.0;
                    const double npos = std::min(9.0 * nw, (double)B * q.H[l] * q.W[l]);
                    o.flops += 2.0 * (npos * 9.0 * q.Cin[l] * q.cmid + nw * (9.0 * q.cmid * q.cmid + (double)q.cmid * q.cout));
                    o.bytes += npos * (9.0 * q.Cin[l] * 2 + 2.0 * q.cmid * 2) + nw * q.cout * 4 + (9.0 * q.Cin[l] * q.cmid + 9.0 * q.cmid * q.cmid + q.cmid * q.cout) * 2;
                }
            }
            if (o.sparse_box || o.sparse_cf) o.kernel = "head_select_kernel<1> + head_pos_kernel + head_win_kernel + head_select_kernel<2>";
        }
        for (auto& o : e.ops)
            if (o.skip) { o.flops = 0; o.bytes = 0; o.kernel = "-"; }
    }
    return YP_OK;
}

static int allocate_plan(yp_engine& e) {
    if (e.allocated) return YP_OK;
    // A dense conv packed with padded taps (WeightDesc::cin_pad > Cin, see conv_params) reads cin_pad channels per pixel and multiplies the
    // surplus by zero weights: the surplus bytes are the next channels of the same pixel (in a concat buffer: a slice a LATER op writes), the
    // next pixel, or - at the last pixel of the last image - up to (cin_pad - Cin) * 2 < 64 bytes past the end of the tensor. Every one of
    // those bytes must be a finite bf16 pattern (NaN * 0 = NaN on the matrix cores; a NaN output then lands in the very slice the next
    // forward over-reads, i.e. it would never heal). So (i) a tensor read by such a conv owns a 64-byte tail inside its slot, and (ii) the
    // whole region of a layout is zeroed on EVERY layout, not only when the arena grows: a re-plan into the kept arena puts bf16 tensors
    // over stale fp32 logits / u32 keys, whose low halves read as bf16 NaN once in 256.
    std::vector<char> tail(e.tensors.size(), 0);
    for (const auto& o : e.ops)
        if (o.kind == OP_CONV && o.widx >= 0 && o.in.t >= 0 && e.weights[o.widx].cin_pad > o.in.C) tail[o.in.t] = 1;
    auto slot = [&](size_t i) { return (e.tensors[i].bytes + (tail[i] ? 64 : 0) + 255) & ~(size_t)255; };
    size_t total = 0;
    for (size_t i = 0; i < e.tensors.size(); ++i) total += slot(i);
    HIPCHK(hipSetDevice(e.device));
    total += 4096;                                   // (slack behind the last tensor: padded-tap reads of conv_igemm's plain loads)
    if (total > e.arena_bytes) {
        if (e.arena) HIPCHK(hipFree(e.arena));
        e.arena = nullptr;
        HIPCHK(hipMalloc(&e.arena, total));
        e.arena_bytes = total;
    }
    HIPCHK(hipMemset(e.arena, 0, total));            // (2.9 GB at S / bs 32: ~1 ms, once per plan)
    size_t off = 0;
    for (size_t i = 0; i < e.tensors.size(); ++i) {
        e.tensors[i].ptr = (char*)e.arena + off;
        off += slot(i);
    }
    {
        size_t A = 0;
        for (int l = 0; l < 3; ++l) A += (size_t)(e.pH / (8 << l)) * (e.pW / (8 << l));
        const size_t need = head_scratch_bytes(e.pB, (int)A);
        if (need > e.head_ws_bytes) {
            if (e.head_ws) HIPCHK(hipFree(e.head_ws));
            e.head_ws = nullptr;
            HIPCHK(hipMalloc(&e.head_ws, need));
            e.head_ws_bytes = need;
        }
    }
    {
        bool sparse = false;
        for (const auto& o : e.ops) sparse |= o.kind == OP_HEAD && (o.sparse_box || o.sparse_cf);
        const size_t need = sparse ? sparse_ws_layout(e).total : 0;
        if (need > e.sp_ws_bytes) {
            if (e.sp_ws) HIPCHK(hipFree(e.sp_ws));
            e.sp_ws = nullptr; e.sp_ws_bytes = 0;
            HIPCHK(hipMalloc(&e.sp_ws, need));
            HIPCHK(hipMemset(e.sp_ws, 0, need));
            e.sp_ws_bytes = need;
        }
    }
    if (e.desc.family != YP_FAMILY_V10) {
        size_t A = 0;
        for (int l = 0; l < 3; ++l) A += (size_t)(e.pH / (8 << l)) * (e.pW / (8 << l));
        const size_t need = head_nms_scratch_bytes(e.pB, (int)A);
        if (need > e.nms_ws_bytes) {
            if (e.nms_ws) HIPCHK(hipFree(e.nms_ws));
            e.nms_ws = nullptr; e.nms_ws_bytes = 0;
            HIPCHK(hipMalloc(&e.nms_ws, need));
            e.nms_ws_bytes = need;
        }
        if (!e.d_nms) {
            HIPCHK(hipMalloc((void**)&e.d_nms, 2 * sizeof(float)));
            const float v[2] = {e.nms_conf, e.nms_iou};
            HIPCHK(hipMemcpy(e.d_nms, v, sizeof(v), hipMemcpyHostToDevice));
        }
    }
    if ((size_t)e.pB > e.o_cap) {
        if (e.o_det) { HIPCHK(hipFree(e.o_det)); HIPCHK(hipFree(e.o_idx)); HIPCHK(hipFree(e.o_coeff)); }
        const size_t rows = (size_t)e.pB * e.desc.max_det;
        HIPCHK(hipMalloc(&e.o_det, rows * 6 * sizeof(float)));
        HIPCHK(hipMalloc(&e.o_idx, rows * sizeof(int32_t)));
        HIPCHK(hipMalloc(&e.o_coeff, rows * 32 * sizeof(float)));
        e.o_cap = (size_t)e.pB;
    }
    if (e.gexec) { (void)hipDeviceSynchronize(); drop_graph(e); }   // (never under a running replay)
    e.allocated = true;
    return YP_OK;
}

// ---------------------------------------------------------------------------------------------------------
// op launch
// ---------------------------------------------------------------------------------------------------------
struct RunArgs { const uint8_t* in; float* det; int32_t* idx; float* coeff; };

static ConvParams conv_params(const yp_engine& e, const Op& o) {
    if (o.fused2) {            // the 3x3 s2 producer's parameters with this 1x1 as the trailing stage
        const Op& a = e.ops[o.fuse_pre];
        Op a1 = a;
        a1.cfg = o.cfg;
        ConvParams p = conv_params(e, a1);
        const WeightDesc& w2 = e.weights[o.widx];
        const TensorDesc& to = e.tensors[o.out.t];
        p.y = to.ptr; p.y_stride = to.C; p.y_coff = o.out.coff; p.y_bytes = to.bytes;
        p.w2 = w2.d_w; p.bias2 = w2.d_b; p.C2 = o.out.C; p.act2 = o.act; p.Kpad2 = w2.Kpad; p.w2_bytes = w2.mat_bytes;
        return p;
    }
    const WeightDesc& w = e.weights[o.widx];
    const TensorDesc &ti = e.tensors[o.in.t], &to = e.tensors[o.out.t];
    ConvParams p{};
    p.x = ti.ptr; p.x_stride = ti.C; p.x_coff = o.in.coff; p.H = ti.H; p.W = ti.W; p.Cin = o.in.C;
    if (w.cin_pad > o.in.C) p.Cin = w.cin_pad;      // (packed with padded taps: the kernels read cin_pad channels per pixel)
    p.w = w.d_w; p.Kpad = w.Kpad; p.bias = w.d_b;
    p.y = to.ptr; p.y_stride = to.C; p.y_coff = o.out.coff; p.Ho = to.H; p.Wo = to.W; p.Cout = o.out.C;
    if (o.res.t >= 0) { p.res = e.tensors[o.res.t].ptr; p.res_stride = e.tensors[o.res.t].C; p.res_coff = o.res.coff; }
    p.M = e.pB * to.H * to.W; p.ks = o.k; p.stride = o.s; p.pad = o.k / 2; p.act = o.act;
    p.out_f32 = (to.f32 && e.dtype == DT_BF16) ? 1 : 0;
    p.up = 1; p.oy = 0; p.ox = 0;
    p.x_bytes = ti.bytes; p.w_bytes = w.mat_bytes; p.y_bytes = to.bytes; p.cfg = o.cfg; p.dbg = conv_debug_ablation();
    if (o.folded) {
        const Op& u = e.ops[o.fold_up];
        const TensorDesc& t2 = e.tensors[u.in.t];
        p.x2 = t2.ptr; p.x2_bytes = t2.bytes; p.x2_stride = t2.C; p.x2_coff = u.in.coff; p.x2_C = u.out.C; p.x2_H = t2.H; p.x2_W = t2.W;
    }
    return p;
}

static FrontParams front_params(const yp_engine& e, const Op& o, const uint8_t* img) {
    const Op& c1 = e.ops[o.fuse_pre];
    const Op& st = e.ops[o.stem_op];
    const WeightDesc &w0 = e.weights[st.widx], &w1 = e.weights[c1.widx], &w2 = e.weights[o.widx];
    const TensorDesc &t0 = e.tensors[st.out.t], &to = e.tensors[o.out.t];
    FrontParams p{};
    p.img = img; p.imgH = e.pH; p.imgW = e.pW; p.B = e.pB;
    p.w0 = w0.d_w2; p.bias0 = w0.d_b; p.act0 = st.act; p.C0 = st.out.C; p.H1 = t0.H; p.W1 = t0.W;
    p.w1 = w1.d_w; p.bias1 = w1.d_b; p.act1 = c1.act; p.C1 = c1.out.C; p.Kpad1 = w1.Kpad; p.w1_bytes = w1.mat_bytes;
    p.w2 = w2.d_w; p.bias2 = w2.d_b; p.act2 = o.act; p.C2 = o.out.C; p.Kpad2 = w2.Kpad; p.w2_bytes = w2.mat_bytes;
    p.y = to.ptr; p.y_stride = to.C; p.y_coff = o.out.coff; p.y_bytes = to.bytes; p.Ho = to.H; p.Wo = to.W;
    return p;
}

static C2fParams c2f_params(const yp_engine& e, const Op& o) {
    const Op &m1 = e.ops[o.c2f_m1], &m2 = e.ops[o.c2f_m2];
    const WeightDesc &w1 = e.weights[m1.widx], &w2 = e.weights[m2.widx], &w3 = e.weights[o.widx];
    const TensorDesc &ti = e.tensors[o.in.t], &to = e.tensors[o.out.t];
    C2fParams p{};
    p.x = ti.ptr; p.x_stride = ti.C; p.x_coff = o.in.coff; p.x_bytes = ti.bytes; p.B = e.pB; p.H = ti.H; p.W = ti.W; p.C = m1.in.C;
    p.w1 = w1.d_w; p.bias1 = w1.d_b; p.act1 = m1.act; p.Kpad1 = w1.Kpad; p.w1_bytes = w1.mat_bytes;
    p.w2 = w2.d_w; p.bias2 = w2.d_b; p.act2 = m2.act; p.Kpad2 = w2.Kpad; p.w2_bytes = w2.mat_bytes;
    p.shortcut = m2.res.t >= 0 ? 1 : 0;
    p.w3 = w3.d_w; p.bias3 = w3.d_b; p.act3 = o.act; p.Kpad3 = w3.Kpad; p.Cout = o.out.C; p.w3_bytes = w3.mat_bytes;
    p.y = to.ptr; p.y_stride = to.C; p.y_coff = o.out.coff; p.y_bytes = to.bytes;
    // the kernel's channel map: a = [coff, coff+C), b = the next C (the bottleneck's input and residual), c = the C after that
    if (m1.in.t != o.in.t || m1.in.coff != o.in.coff + p.C || m2.out.t != o.in.t || m2.out.coff != o.in.coff + 2 * p.C || o.in.C != 3 * p.C ||
        m1.k != 3 || m2.k != 3 || m1.s != 1 || m2.s != 1 || m1.out.C != p.C || m2.out.C != p.C || m1.res.t >= 0 ||
        (m2.res.t >= 0 && (m2.res.t != m1.in.t || m2.res.coff != m1.in.coff)) || (to.f32 && e.dtype == DT_BF16)) p.C = 0;
    return p;
}

static ScdParams scd_params(const yp_engine& e, const Op& o) {
    const Op& c1 = e.ops[o.scd_pre];
    const WeightDesc &w1 = e.weights[c1.widx], &wd = e.weights[o.widx];
    const TensorDesc &ti = e.tensors[c1.in.t], &to = e.tensors[o.out.t];
    ScdParams p{};
    p.x = ti.ptr; p.x_stride = ti.C; p.x_coff = c1.in.coff; p.x_bytes = ti.bytes; p.B = e.pB; p.H = ti.H; p.W = ti.W; p.K = c1.in.C;
    p.w1 = w1.d_w; p.bias1 = w1.d_b; p.act1 = c1.act; p.Kpad1 = w1.Kpad; p.C = c1.out.C; p.w1_bytes = w1.mat_bytes;
    p.wd = wd.d_w; p.biasd = wd.d_b; p.actd = o.act;
    p.y = to.ptr; p.y_stride = to.C; p.y_coff = o.out.coff; p.y_bytes = to.bytes; p.Ho = to.H; p.Wo = to.W;
    // the kernel's assumptions beyond the sizes: plain 1x1 s1 -> 3x3 s2 over the 1x1's whole output, no residuals, no gather
    if (c1.k != 1 || c1.s != 1 || c1.res.t >= 0 || o.k != 3 || o.s != 2 || o.res.t >= 0 || o.gs != 0 || o.in.t != c1.out.t ||
        o.in.coff != c1.out.coff || o.in.C != c1.out.C || o.out.C != c1.out.C || c1.fold_up >= 0 || to.f32) p.K = 0;
    return p;
}

static ClsOutParams cls_out_params(const yp_engine& e, const Op& o) {
    ClsOutParams p{};
    const WeightDesc& w = e.weights[o.widx];
    const TensorDesc &ti = e.tensors[o.in.t], &to = e.tensors[o.out.t], &tk = e.tensors[e.ops[o.amax_post].out.t];
    p.x = ti.ptr; p.x_stride = ti.C; p.x_coff = o.in.coff; p.x_bytes = ti.bytes;
    p.M = e.pB * to.H * to.W; p.K = o.in.C; p.nc = o.out.C;
    p.w = w.d_w; p.Kpad = w.Kpad; p.w_bytes = w.mat_bytes; p.bias = w.d_b;
    p.y = (float*)to.ptr; p.y_stride = to.C; p.y_coff = o.out.coff;
    p.keys = (unsigned*)tk.ptr;
    if (w.cin_pad != o.in.C) p.K = 0;
    return p;
}

// pwsp_kernel: `o` is the spatial op of a fused pair (o.fused7) or a plain 1x1 conv that runs in the same decomposition (cfg PWSP_CFG)
static PwSpParams pwsp_params(const yp_engine& e, const Op& o) {
    PwSpParams p{};
    const bool pair = o.kind != OP_CONV;
    const Op& c1 = pair ? e.ops[o.pw_pre] : o;
    const WeightDesc& w1 = e.weights[c1.widx];
    const TensorDesc &ti = e.tensors[c1.in.t], &t1 = e.tensors[c1.out.t];
    p.x = ti.ptr; p.x_stride = ti.C; p.x_coff = c1.in.coff; p.x_bytes = ti.bytes; p.B = e.pB; p.H = ti.H; p.W = ti.W; p.K = c1.in.C;
    p.w1 = w1.d_w; p.bias1 = w1.d_b; p.act1 = c1.act; p.Kpad1 = w1.Kpad; p.C1 = c1.out.C; p.w1_bytes = w1.mat_bytes;
    p.dbg = conv_debug_ablation();
    if (!pair || o.pw_store) { p.y1 = t1.ptr ? t1.ptr : (void*)1; p.y1_stride = t1.C; p.y1_coff = c1.out.coff; }
    if (!pair) {
        if (c1.res.t >= 0) { const TensorDesc& tr = e.tensors[c1.res.t]; p.res1 = tr.ptr ? tr.ptr : (const void*)1; p.res1_stride = tr.C; p.res1_coff = c1.res.coff; }
        if (w1.cin_pad != c1.in.C || t1.f32 || c1.k != 1 || c1.s != 1 || c1.fold_up >= 0) p.sp = -1;      // (not this kernel's shape)
        return p;
    }
    const TensorDesc& t2 = e.tensors[o.out.t];
    p.sp = o.kind == OP_POOL3 ? 3 : (o.k == 3 ? 1 : 2);
    p.sp_c0 = o.in.coff - c1.out.coff; p.Csp = o.in.C;
    if (o.kind == OP_DWCONV) {
        const WeightDesc& wd = e.weights[o.widx];
        p.wd = wd.d_w; p.biasd = wd.d_b; p.actd = o.act;
        if (o.res.t >= 0) { const TensorDesc& tr = e.tensors[o.res.t]; p.res = tr.ptr ? tr.ptr : (const void*)1; p.res_stride = tr.C; p.res_coff = o.res.coff; }
        if (o.out.C != o.in.C) p.sp = -1;
    } else if (o.out.C != 3 * o.in.C) p.sp = -1;
    p.y2 = t2.ptr ? t2.ptr : (void*)1; p.y2_stride = t2.C; p.y2_coff = o.out.coff;
    if (w1.cin_pad != c1.in.C || c1.in.t < 0) p.sp = -1;
    return p;
}

static hipError_t run_op(yp_engine& e, const Op& o, const RunArgs& a, hipStream_t st) {
    auto T = [&](const View& v) -> const TensorDesc& { return e.tensors[v.t]; };
    const int B = e.pB;
    switch (o.kind) {
        case OP_STEM: {
            const WeightDesc& w = e.weights[o.widx];
            const TensorDesc& to = T(o.out);
            StemParams p{};
            p.x = a.in; p.H = e.pH; p.W = e.pW; p.w = (const float*)w.d_w; p.wpk = w.d_w2; p.bias = w.d_b;
            p.y = to.ptr; p.y_stride = to.C; p.y_coff = o.out.coff; p.Ho = to.H; p.Wo = to.W; p.C0 = o.out.C; p.B = B; p.act = o.act;
            return launch_stem(p, e.dtype, st);
        }
        case OP_CONV:
            if (o.fused || o.fused6) return launch_conv_dwpw(dwpw_params(e, o), st);
            if (o.fused3) return launch_frontend(front_params(e, o, a.in), st);
            if (o.fused4) return launch_c2f_fused(c2f_params(e, o), st);
            if (o.fused2) { const ConvParams q = conv_params(e, o); return launch_conv_halo_s2(q, o.cfg - 500, st); }
            if (o.fused8) return launch_cls_out(cls_out_params(e, o), st);
            if (o.cfg == PWSP_CFG) return launch_pwsp(pwsp_params(e, o), st);
            if (conv_dma_forced_cfg() == PWSP_CFG && e.dtype == DT_BF16 && !o.folded) {          // test hook (yp_debug_force_conv_cfg): every 1x1 that admits it
                const PwSpParams q = pwsp_params(e, o);
                if (q.sp == 0 && pwsp_valid(q)) return launch_pwsp(q, st);
            }
            return launch_conv(conv_params(e, o), e.dtype, st);
        case OP_CONVT: {
            const WeightDesc& w = e.weights[o.widx];
            const TensorDesc &ti = T(o.in), &to = T(o.out);
            for (int dy = 0; dy < 2; ++dy)
                for (int dx = 0; dx < 2; ++dx) {
                    ConvParams p{};
                    p.x = ti.ptr; p.x_stride = ti.C; p.x_coff = o.in.coff; p.H = ti.H; p.W = ti.W; p.Cin = o.in.C;
                    const size_t sub = (size_t)((w.cout + 127) / 128 * 128) * w.Kpad * e.es();
                    p.w = (const char*)w.d_w + sub * (dy * 2 + dx); p.Kpad = w.Kpad; p.bias = w.d_b;
                    p.y = to.ptr; p.y_stride = to.C; p.y_coff = o.out.coff; p.Ho = ti.H; p.Wo = ti.W; p.Cout = o.out.C;
                    p.M = B * ti.H * ti.W; p.ks = 1; p.stride = 1; p.pad = 0; p.act = o.act; p.out_f32 = 0;
                    p.up = 2; p.oy = dy; p.ox = dx;
                    p.x_bytes = ti.bytes; p.w_bytes = w.mat_bytes; p.y_bytes = to.bytes; p.cfg = o.cfg;
                    hipError_t err = launch_conv(p, e.dtype, st);
                    if (err != hipSuccess) return err;
                }
            return hipSuccess;
        }
        case OP_DWCONV: {
            if (o.fused5) return launch_scdown_fused(scd_params(e, o), st);
            if (o.fused7) return launch_pwsp(pwsp_params(e, o), st);
            const WeightDesc& w = e.weights[o.widx];
            const TensorDesc &ti = T(o.in), &to = T(o.out);
            DwParams p{};
            p.x = ti.ptr; p.x_stride = ti.C; p.x_coff = o.in.coff; p.H = ti.H; p.W = ti.W; p.C = o.out.C;
            p.w = w.d_w; p.bias = w.d_b;
            p.y = to.ptr; p.y_stride = to.C; p.y_coff = o.out.coff; p.Ho = to.H; p.Wo = to.W;
            if (o.res.t >= 0) { p.res = T(o.res).ptr; p.res_stride = T(o.res).C; p.res_coff = o.res.coff; }
            p.B = B; p.ks = o.k; p.stride = o.s; p.pad = o.k / 2; p.act = o.act; p.gs = o.gs; p.gstride = o.gstride;
            p.x_bytes = ti.bytes;
            return launch_dwconv(p, e.dtype, st);
        }
        case OP_POOL5: {
            const TensorDesc &ti = T(o.in), &to = T(o.out);
            PoolParams p{};
            p.x = ti.ptr; p.x_stride = ti.C; p.x_coff = o.in.coff; p.y = to.ptr; p.y_stride = to.C; p.y_coff = o.out.coff;
            p.B = B; p.H = ti.H; p.W = ti.W; p.C = o.in.C;
            return launch_pool5(p, e.dtype, st);
        }
        case OP_POOL3: {
            const TensorDesc &ti = T(o.in), &to = T(o.out);
            if (o.fused7) return launch_pwsp(pwsp_params(e, o), st);
            PoolParams p{};
            p.x = ti.ptr; p.x_stride = ti.C; p.x_coff = o.in.coff; p.y = to.ptr; p.y_stride = to.C; p.y_coff = o.out.coff;
            p.B = B; p.H = ti.H; p.W = ti.W; p.C = o.in.C;
            if (sppf_pool3_fits(p, e.dtype)) return launch_sppf_pool3(p, e.dtype, st);
            for (int i = 0; i < 3; ++i) {           // large maps: three chained launches
                PoolParams q = p;
                q.x_coff = o.in.coff + i * o.in.C; q.y_coff = o.out.coff + i * o.in.C;
                hipError_t err = launch_pool5(q, e.dtype, st);
                if (err != hipSuccess) return err;
            }
            return hipSuccess;
        }
        case OP_UPSAMPLE: {
            const TensorDesc &ti = T(o.in), &to = T(o.out);
            UpParams p{};
            p.x = ti.ptr; p.x_stride = ti.C; p.x_coff = o.in.coff; p.y = to.ptr; p.y_stride = to.C; p.y_coff = o.out.coff;
            p.B = B; p.H = ti.H; p.W = ti.W; p.C = o.in.C;
            return launch_upsample(p, e.dtype, st);
        }
        case OP_ATTN: {
            const TensorDesc &ti = T(o.in), &to = T(o.out);
            AttnParams p{};
            p.qkv = ti.ptr; p.q_stride = ti.C; p.q_coff = o.in.coff; p.o = to.ptr; p.o_stride = to.C; p.o_coff = o.out.coff;
            p.B = B; p.N = ti.H * ti.W; p.nh = o.nh; p.kd = o.kd; p.hd = o.hd; p.scale = 1.0f / std::sqrt((float)o.kd);
            return launch_attention(p, e.dtype, st);
        }
        case OP_AMAX: {
            const TensorDesc &ti = T(o.in), &to = T(o.out);
            return launch_anchor_max_level((const float*)ti.ptr, B, ti.H * ti.W, o.in.C, (unsigned*)to.ptr, st);
        }
        case OP_HEAD: {
            HeadParams p{};
            p.nlev = 3; p.A = 0;
            for (int l = 0; l < 3; ++l) {
                const TensorDesc& tb = T(o.box[l]);
                p.box[l] = (const float*)tb.ptr; p.cls[l] = (const float*)T(o.cls[l]).ptr;
                p.cf[l] = (o.cf[l].t >= 0) ? (const float*)T(o.cf[l]).ptr : nullptr;
                p.hw[l][0] = tb.H; p.hw[l][1] = tb.W; p.A += tb.H * tb.W;
            }
            p.B = B; p.nc = e.desc.nc; p.max_det = e.desc.max_det;
            p.det = a.det; p.idx = a.idx; p.coeff = (o.cf[0].t >= 0) ? a.coeff : nullptr; p.scratch = e.head_ws;
            for (int l = 0; l < 3; ++l) p.mk[l] = (o.amax[l].t >= 0) ? (const unsigned*)T(o.amax[l]).ptr : nullptr;
            if (o.nms) { p.nms_params = e.d_nms; p.nms_ws = (float*)e.nms_ws; return launch_head_nms(p, st); }
            if (o.sparse_box || o.sparse_cf) {
                // winners-only head: stage 1 -> the branch(es) on the winners -> stage 2 + decode. A branch that stays dense (an unsupported
                // width) is read from its dense map as before.
                const SparseWs ws = sparse_ws_layout(e);
                char* base = (char*)e.sp_ws;
                p.sp_sel = (int*)(base + ws.sel); p.sp_wlist = (int*)(base + ws.wlist); p.sp_wcount = (int*)(base + ws.wcount); p.sp_thr = (unsigned*)(base + ws.thr);
                p.sp_box = o.sparse_box ? (float*)(base + ws.box) : nullptr;
                p.sp_cf = (o.sparse_cf && p.coeff) ? (float*)(base + ws.cf) : nullptr;
                p.sp_plist = (int*)(base + ws.plist); p.sp_pcount = (int*)(base + ws.pcount);
                for (int l = 0; l < 3; ++l) { p.sp_plist_off[l] = ws.plist_off[l]; p.sp_plist_cap[l] = ws.plist_cap[l]; }
                hipError_t err = launch_head_stage1(p, st);
                if (err != hipSuccess) return err;
                if (o.sparse_box && (err = launch_head_branch(head_branch_params(e, o, 0), st)) != hipSuccess) return err;
                if (p.sp_cf && (err = launch_head_branch(head_branch_params(e, o, 1), st)) != hipSuccess) return err;
                return launch_head_stage2(p, st);
            }
            return launch_head(p, st);
        }
    }
    return hipErrorInvalidValue;
}

// Plan-time autotuner: for every dense conv that the LDS-DMA kernel supports, time each valid tile configuration on
// the real tensors (weights are loaded, activations hold whatever the arena holds - timing does not depend on values
// up to DVFS) and keep the fastest. Runs once per (B,H,W) plan, outside any graph capture.
static bool views_overlap(const View& a, const View& b);
static void op_views(const yp_engine& e, const Op& o, std::vector<View>& rd, std::vector<View>& wr);

static int autotune(yp_engine& e) {
    if (e.dtype != DT_BF16) return YP_OK;
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    RunArgs none{nullptr, nullptr, nullptr, nullptr};
    // Cold timing: inside the replayed graph a layer's input was written by the previous kernel and (for the large maps) is
    // no longer in L2 / Infinity Cache, while back-to-back repetitions of one op find it there. Writing a buffer larger
    // than the Infinity Cache before every timed repetition makes the tuner rank configurations as the graph will see them.
    // YOLOP_TUNE_COLD: 0 = back-to-back repetitions (input in L2), 1 = everything evicted before each repetition (cold),
    // 2 = "as in the graph": the op that produces this op's input runs right before every timed repetition, so the input sits where the
    // replayed graph leaves it (fresh in the Infinity Cache, partly in the producing XCDs' L2) - neither as warm as mode 0 nor as cold
    // as mode 1. Measured end to end (same box, two runs each): mode 1 15.37 / 15.43 k img/s, mode 2 15.27 / 15.21 k - the cold ranking
    // stays the default
    static const int cold_mode = [] { const char* v = std::getenv("YOLOP_TUNE_COLD"); return v ? atoi(v) : 1; }();
    void* flush = nullptr;
    const size_t flush_bytes = (size_t)320 << 20;
    if (cold_mode) HIPCHK(hipMalloc(&flush, flush_bytes));
    std::vector<std::vector<View>> rdv(e.ops.size()), wrv(e.ops.size());
    for (size_t i = 0; i < e.ops.size(); ++i) op_views(e, e.ops[i], rdv[i], wrv[i]);
    auto producer_of = [&](const Op& o) -> const Op* {          // the latest earlier op that writes something this op reads
        const size_t i = (size_t)(&o - e.ops.data());
        for (size_t j = i; j-- > 0;) {
            if (e.ops[j].skip || e.ops[j].kind == OP_HEAD) continue;
            for (const View& w : wrv[j])
                for (const View& r : rdv[i])
                    if (views_overlap(w, r)) return &e.ops[j];
        }
        return nullptr;
    };
    const uint8_t* tune_in = e.tune_input;
    RunArgs warm{tune_in, nullptr, nullptr, nullptr};
    auto time_cfg = [&](Op& o, float& tmin) -> hipError_t {
        tmin = 1e30f;
        const Op* prod = (cold_mode == 2) ? producer_of(o) : nullptr;
        if (prod && (prod->kind == OP_STEM || prod->fused3) && !tune_in) prod = nullptr;      // (needs the caller's frames)
        for (int rep = 0; rep < 4; ++rep) {
            if (flush && rep > 0 && (cold_mode == 1 || !prod)) { hipError_t fe = hipMemsetAsync(flush, rep, flush_bytes, nullptr); if (fe != hipSuccess) return fe; }
            if (prod) {
                if (rep == 1) { hipError_t fe = hipMemsetAsync(flush, rep, flush_bytes, nullptr); if (fe != hipSuccess) return fe; }   // once: nothing older than the producer stays warm
                hipError_t pe = run_op(e, *prod, warm, nullptr);
                if (pe != hipSuccess) return pe;
            }
            hipError_t ee;
            if ((ee = hipEventRecord(e0, nullptr)) != hipSuccess) return ee;
            hipError_t err = run_op(e, o, none, nullptr);
            if (err != hipSuccess) return err;
            if ((ee = hipEventRecord(e1, nullptr)) != hipSuccess) return ee;
            if ((ee = hipEventSynchronize(e1)) != hipSuccess) return ee;
            float ms = 0;
            if ((ee = hipEventElapsedTime(&ms, e0, e1)) != hipSuccess) return ee;
            if (rep > 0) tmin = std::min(tmin, ms);
        }
        return hipSuccess;
    };
    for (Op& o : e.ops) {
        if (o.kind != OP_CONV && o.kind != OP_CONVT) continue;
        if (o.fused || o.fused2 || o.fused4 || o.fused6 || o.fused8 || o.skip) continue;
        ConvParams p{};
        if (o.kind == OP_CONV) p = conv_params(e, o);
        else { p.Cin = o.in.C; p.Cout = o.out.C; p.ks = 1; p.Kpad = e.weights[o.widx].Kpad; p.M = e.pB * e.tensors[o.in.t].H * e.tensors[o.in.t].W;
               p.x_bytes = e.tensors[o.in.t].bytes; p.w_bytes = e.weights[o.widx].mat_bytes; }
        if (!conv_dma_supported(p)) continue;
        float best = 1e30f;
        int bestc = -1;
        std::vector<std::pair<float, int>> timed;                           // (first-pass time, cfg) of every candidate
        for (int c = 0; c < conv_dma_num_cfgs(); ++c) {
            if (p.x2_C > 0 || !conv_dma_cfg_valid(p, c)) continue;          // (the one-tile-per-workgroup family has no folded-upsample gather)
            o.cfg = c;
            float tmin;
            hipError_t err = time_cfg(o, tmin);
            if (err != hipSuccess) return fail(YP_ERR_HIP, "autotune op %s cfg %d: %s", o.name.c_str(), c, hipGetErrorString(err));
            timed.emplace_back(tmin, c);
            if (tmin < best) { best = tmin; bestc = c; }
        }
        if (o.kind == OP_CONV) {
            std::vector<int> cands;
            for (int c = 0; c < conv_halo_num_cfgs(); ++c) if (conv_halo_cfg_valid(p, c)) cands.push_back(100 + c);
            for (int c = 0; c < conv_halo_p_num_cfgs(); ++c) if (conv_halo_p_cfg_valid(p, c)) cands.push_back(200 + c);
            for (int c = 0; c < conv_dma_p_num_cfgs(); ++c) if (conv_dma_p_cfg_valid(p, c)) cands.push_back(300 + c);
            static const bool no_s2 = [] { const char* v = std::getenv("YOLOP_NO_S2"); return v && *v == '1'; }();   // A/B switch
            for (int c = 0; !no_s2 && c < conv_halo_s2_num_cfgs(); ++c) if (conv_halo_s2_cfg_valid(p, c)) cands.push_back(500 + c);
            static const bool no_t1 = [] { const char* v = std::getenv("YOLOP_NO_T1"); return v && *v == '1'; }();   // A/B switch
            for (int c = 0; !no_t1 && c < conv_tile1_num_cfgs(); ++c) if (conv_tile1_cfg_valid(p, c)) cands.push_back(600 + c);
            static const bool no_lc = [] { const char* v = std::getenv("YOLOP_NO_LC"); return v && *v == '1'; }();   // A/B switch
            for (int c = 0; !no_lc && c < conv_dma_lc_num_cfgs(); ++c) if (conv_dma_lc_cfg_valid(p, c)) cands.push_back(400 + c);
            static const bool no_wr = [] { const char* v = std::getenv("YOLOP_NO_WREG"); return v && *v == '1'; }();   // A/B switch
            for (int c = 0; !no_wr && c < conv_wreg_num_cfgs(); ++c) if (conv_wreg_cfg_valid(p, c)) cands.push_back(700 + c);
            static const bool no_px = [] { const char* v = std::getenv("YOLOP_NO_PXD"); return v && *v == '1'; }();     // A/B switch
            for (int c = 0; !no_px && c < conv_pxd_num_cfgs(); ++c) if (conv_pxd_cfg_valid(p, c)) cands.push_back(800 + c);
            static const bool no_ks = [] { const char* v = std::getenv("YOLOP_NO_KS"); return v && *v == '1'; }();     // A/B switch
            for (int c = 0; !no_ks && c < conv_ks_num_cfgs(); ++c) if (conv_ks_cfg_valid(p, c)) cands.push_back(900 + c);
            static const bool no_wres = [] { const char* v = std::getenv("YOLOP_NO_WRES"); return v && *v == '1'; }();   // A/B switch
            for (int c = 0; !no_wres && c < conv_wres_num_cfgs(); ++c) if (conv_wres_cfg_valid(p, c)) cands.push_back(1100 + c);
            // (opt-in: stand-alone - the tuner's protocol - the weights-in-registers form wins several 40x40 layers by 1-3 us; with it among the
            // candidates the step is 1.6977 against 1.7003 ms over three tunings each: two more configurations to time for nothing; DESIGN.md "Round 4")
            static const bool use_wrs = [] { const char* v = std::getenv("YOLOP_WRS"); return v && *v == '1'; }();
            for (int c = 0; use_wrs && c < conv_wrs_num_cfgs(); ++c) if (conv_wrs_cfg_valid(p, c)) cands.push_back(1200 + c);
            static const bool no_ps = [] { const char* v = std::getenv("YOLOP_NO_PWSP"); return v && *v == '1'; }();     // A/B switch
            if (!no_ps && !o.folded && p.x2_C == 0 && pwsp_valid(pwsp_params(e, o))) cands.push_back(PWSP_CFG);
            for (int cc : cands) {
                o.cfg = cc;
                float tmin;
                hipError_t err = time_cfg(o, tmin);
                if (err != hipSuccess) return fail(YP_ERR_HIP, "autotune op %s cfg %d: %s", o.name.c_str(), cc, hipGetErrorString(err));
                timed.emplace_back(tmin, cc);
                if (tmin < best) { best = tmin; bestc = cc; }
            }
        }
        // second pass over the three fastest: a minimum of three samples is noisy enough that a 5 % slower configuration sometimes
        // wins the first pass, and one bad pick on a 50-us layer costs the whole step 1-2 %
        if (timed.size() > 1) {
            std::sort(timed.begin(), timed.end());
            best = 1e30f;
            for (size_t k = 0; k < std::min<size_t>(3, timed.size()); ++k) {
                o.cfg = timed[k].second;
                float t = timed[k].first;
                for (int again = 0; again < 2; ++again) {
                    float tmin;
                    hipError_t err = time_cfg(o, tmin);
                    if (err != hipSuccess) return fail(YP_ERR_HIP, "autotune op %s cfg %d: %s", o.name.c_str(), o.cfg, hipGetErrorString(err));
                    t = std::min(t, tmin);
                }
                if (t < best) { best = t; bestc = timed[k].second; }
            }
        }
        o.cfg = bestc;
        if (o.cfg == PWSP_CFG) o.kernel = pwsp_kernel_name(pwsp_params(e, o));
        else if (o.kind == OP_CONV) o.kernel = conv_kernel_name(conv_params(e, o), e.dtype);
        else { p.cfg = o.cfg; o.kernel = conv_kernel_name(p, e.dtype); }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (flush) (void)hipFree(flush);
    return YP_OK;
}

// Optional on-disk cache of the autotuner's choices (env YOLOP_TUNE_CACHE=<path prefix>): one file per
// (variant, task, dtype, B, H, W), lines "<op name> <cfg>". Lets a profiled run skip the tuning launches.
static const int TUNE_TABLE_VERSION = 5;
// Packaged tables: with no YOLOP_TUNE_CACHE in the environment the same files are looked up, read-only, as
// <directory of this library>/tune_tables/tt_<key>.txt - the best of several tunings of the shapes the package was measured on
// (tools/make_tune_table.py; two tunings of one build differ by +-15 us per step, see DESIGN.md "Round 4"). A shape without a table,
// or a table this build cannot launch, falls through to the tuner as before. YOLOP_NO_TUNE_TABLES=1 ignores them.
static std::string packaged_table_prefix() {
    static const std::string pre = [] {
        const char* off = std::getenv("YOLOP_NO_TUNE_TABLES");
        if (off && *off == '1') return std::string();
        Dl_info info;
        if (!dladdr((const void*)&packaged_table_prefix, &info) || !info.dli_fname) return std::string();
        std::string dir(info.dli_fname);
        const size_t k = dir.find_last_of('/');
        dir = (k == std::string::npos) ? std::string(".") : dir.substr(0, k);
        return dir + "/tune_tables/tt";
    }();
    return pre;
}
static std::string tune_cache_path(const yp_engine& e, bool packaged = false) {
    const char* env = std::getenv("YOLOP_TUNE_CACHE");
    std::string pre;
    if (packaged) { if (env && *env) return ""; pre = packaged_table_prefix(); }
    else if (env && *env) pre = env;
    if (pre.empty()) return "";
    std::ostringstream os;
    // "t<N>": bump TUNE_TABLE_VERSION whenever configuration ids are added, removed or renumbered - older files are then simply not found
    os << pre << "_f" << e.desc.family << (char)e.desc.variant << (e.desc.task ? "seg" : "det") << "_nc" << e.desc.nc << "_dt" << e.dtype << "_" << e.pB << "x" << e.pH << "x" << e.pW
       << "_t" << TUNE_TABLE_VERSION << ".txt";
    return os.str();
}
// Install tile configurations that did not come from this process's tuner (cache file, yp_tuning_import): one id per op of the current
// plan (ignored for ops that are not tunable convs). Every id is checked with the predicates the tuner itself uses; all or nothing.
static ConvParams tune_params(const yp_engine& e, const Op& o) {
    if (o.kind == OP_CONV) return conv_params(e, o);
    ConvParams p{};
    p.Cin = o.in.C; p.Cout = o.out.C; p.ks = 1; p.Kpad = e.weights[o.widx].Kpad; p.M = e.pB * e.tensors[o.in.t].H * e.tensors[o.in.t].W;
    p.x_bytes = e.tensors[o.in.t].bytes; p.w_bytes = e.weights[o.widx].mat_bytes;
    return p;
}
static bool apply_tuning(yp_engine& e, const int* cfgs, int n) {
    if (n != (int)e.ops.size()) return false;
    for (size_t i = 0; i < e.ops.size(); ++i) {
        const Op& o = e.ops[i];
        if ((o.kind != OP_CONV && o.kind != OP_CONVT) || o.fused || o.fused2 || o.fused4 || o.fused6 || o.fused8 || o.skip) continue;
        ConvParams p = tune_params(e, o);
        p.cfg = -1;
        if (cfgs[i] == PWSP_CFG) { if (o.kind != OP_CONV || o.folded || !pwsp_valid(pwsp_params(e, o))) return false; continue; }
        if (!conv_cfg_usable(p, e.dtype, cfgs[i])) return false;
    }
    for (size_t i = 0; i < e.ops.size(); ++i) {
        Op& o = e.ops[i];
        if ((o.kind != OP_CONV && o.kind != OP_CONVT) || o.fused || o.fused2 || o.fused4 || o.fused6 || o.fused8 || o.skip) continue;   // a fused op keeps its own symbol / id
        o.cfg = cfgs[i];
        if (o.cfg == PWSP_CFG) { o.kernel = pwsp_kernel_name(pwsp_params(e, o)); continue; }
        ConvParams p = tune_params(e, o);
        p.cfg = o.cfg;
        o.kernel = conv_kernel_name(p, e.dtype);
    }
    return true;
}
static bool load_tune_cache(yp_engine& e, bool packaged = false) {
    const std::string path = tune_cache_path(e, packaged);
    if (path.empty()) return false;
    std::ifstream f(path);
    if (!f) return false;
    std::map<std::string, int> m;
    std::string name;
    int cfg;
    while (f >> name >> cfg) m[name] = cfg;
    std::vector<int> cfgs(e.ops.size(), -1);
    for (size_t i = 0; i < e.ops.size(); ++i) {
        const Op& o = e.ops[i];
        if (o.kind != OP_CONV && o.kind != OP_CONVT) continue;
        auto it = m.find(o.name);
        if (it == m.end()) return false;
        cfgs[i] = it->second;
    }
    return apply_tuning(e, cfgs.data(), (int)cfgs.size());     // an id this build cannot launch for its layer = a cache miss
}
static void save_tune_cache(const yp_engine& e) {
    const std::string path = tune_cache_path(e);
    if (path.empty()) return;
    std::ofstream f(path);
    for (const Op& o : e.ops)
        if (o.kind == OP_CONV || o.kind == OP_CONVT) f << o.name << " " << o.cfg << "\n";
}

static DwPwParams dwpw_params(const yp_engine& e, const Op& c) {
    if (c.fused6) {            // the dw -> pw pair in front with this 1x1 (and the class-max keys) as the third stage
        Op c1 = e.ops[c.fuse_tail];
        c1.fused6 = false;
        DwPwParams p = dwpw_params(e, c1);
        const WeightDesc& w3 = e.weights[c.widx];
        const TensorDesc& to = e.tensors[c.out.t];
        p.w3 = w3.d_w ? w3.d_w : (const void*)1; p.Kpad3 = w3.Kpad; p.w3_bytes = w3.mat_bytes; p.b3 = w3.d_b ? w3.d_b : (const float*)1; p.C3 = c.out.C;
        p.y3 = to.ptr ? (float*)to.ptr : (float*)1; p.y3_stride = to.C; p.y3_coff = c.out.coff; p.y3_bytes = to.bytes;
        p.keys = c.tail_amax >= 0 ? (unsigned*)e.tensors[e.ops[c.tail_amax].out.t].ptr : nullptr;
        p.out_f32 = 0;
        return p;
    }
    const Op& d = e.ops[c.fuse_dw];
    const WeightDesc &wd = e.weights[d.widx], &wp = e.weights[c.widx];
    const TensorDesc &ti = e.tensors[d.in.t], &to = e.tensors[c.out.t];
    DwPwParams p{};
    p.x = ti.ptr; p.x_stride = ti.C; p.x_coff = d.in.coff; p.B = e.pB; p.H = ti.H; p.W = ti.W; p.C = d.in.C; p.x_bytes = ti.bytes;
    p.w_dw = wd.d_w; p.b_dw = wd.d_b; p.act_dw = d.act;
    p.w_pw = wp.d_w; p.Kpad = wp.Kpad; p.wpw_bytes = wp.mat_bytes; p.b_pw = wp.d_b; p.act_pw = c.act;
    p.y = to.ptr; p.y_stride = to.C; p.y_coff = c.out.coff; p.y_bytes = to.bytes; p.Cout = c.out.C;
    p.out_f32 = (to.f32 && e.dtype == DT_BF16) ? 1 : 0;
    return p;
}

// the persistent conv kernels are additionally templated on <HAS_RES, OUT_F32>: make the reported symbol exact
static void finish_kernel_names(yp_engine& e) {
    for (Op& o : e.ops) {
        const bool lc = o.kernel.find("_lc_kernel<") != std::string::npos || o.kernel.find("_s2_kernel<") != std::string::npos || o.kernel.find("_tile1_kernel<") != std::string::npos || o.kernel.find("_tile1w_kernel<") != std::string::npos || o.kernel.find("_wreg_kernel<") != std::string::npos || o.kernel.find("_pxd_kernel<") != std::string::npos || o.kernel.find("_ks_kernel<") != std::string::npos;
        if ((o.kernel.find("_p_kernel<") == std::string::npos && !lc) || o.kernel.find(",false>") != std::string::npos || o.kernel.find(",true>") != std::string::npos) continue;
        const bool f32 = (o.kind == OP_CONV) && e.tensors[o.out.t].f32 && e.dtype == DT_BF16;
        const bool res = o.res.t >= 0;
        bool wres = false, pipe = false, pp = false;
        if (o.kernel.size() > 3 && o.kernel.compare(o.kernel.size() - 3, 3, ",P>") == 0) { pipe = true; o.kernel.erase(o.kernel.size() - 3); o.kernel += ">"; }
        if (o.kernel.size() > 3 && o.kernel.compare(o.kernel.size() - 3, 3, ",Q>") == 0) { pp = true; o.kernel.erase(o.kernel.size() - 3); o.kernel += ">"; }
        if (o.kernel.size() > 3 && o.kernel.compare(o.kernel.size() - 3, 3, ",W>") == 0) { wres = true; o.kernel.erase(o.kernel.size() - 3); }
        else o.kernel.pop_back();
        o.kernel += f32 ? ",false,true" : (res ? ",true,false" : ",false,false");
        if (o.kernel.find("conv_dma_p_kernel") != std::string::npos) { o.kernel += wres ? ",true" : ",false"; o.kernel += pipe ? ",true" : ",false"; o.kernel += pp ? ",true" : ",false"; }
        o.kernel += ">";
    }
}

// ---------------------------------------------------------------------------------------------------------
// Multi-lane launch for graph capture: ops carry a lane; every lane is a stream. Dependencies are derived from the
// tensor views (RAW / WAR / WAW on overlapping channel ranges); a dependency that crosses lanes becomes an event
// record on the producer's stream + a wait on the consumer's. Captured from lane 0's stream this yields a hipGraph
// whose independent head branches (box / class / coefficient per level, prototypes) overlap with the rest of the neck.
// ---------------------------------------------------------------------------------------------------------
static bool views_overlap(const View& a, const View& b) {
    return a.t >= 0 && a.t == b.t && a.coff < b.coff + b.C && b.coff < a.coff + a.C;
}
static void op_views(const yp_engine& e, const Op& o, std::vector<View>& rd, std::vector<View>& wr) {
    rd.clear(); wr.clear();
    if (o.skip) return;
    if (o.fused) rd.push_back(e.ops[o.fuse_dw].in);
    else if (o.fused6) rd.push_back(e.ops[e.ops[o.fuse_tail].fuse_dw].in);
    else if (o.fused5) rd.push_back(e.ops[o.scd_pre].in);
    else if (o.fused7) rd.push_back(e.ops[o.pw_pre].in);
    else if (o.fused4) rd.push_back(View{o.in.t, o.in.coff, 2 * e.ops[o.c2f_m1].in.C});
    else if (o.fused3) { /* reads the caller's frames only */ }
    else if (o.fused2) rd.push_back(e.ops[o.fuse_pre].in);
    else if (o.in.t >= 0) rd.push_back(o.in);
    if (o.folded) rd.push_back(e.ops[o.fold_up].in);      // (besides the concat buffer, whose skip part it still reads)
    if (o.res.t >= 0) rd.push_back(o.res);
    if (o.out.t >= 0) wr.push_back(o.out);
    if (o.fused7 && o.pw_store) wr.push_back(e.ops[o.pw_pre].out);
    if (o.fused8) wr.push_back(e.ops[o.amax_post].out);
    if (o.fused6 && o.tail_amax >= 0) wr.push_back(e.ops[o.tail_amax].out);
    if (o.kind == OP_HEAD)
        for (int l = 0; l < 3; ++l) {
            if (o.sparse_box) rd.push_back(e.ops[o.hb_box[l][0]].in);              // the level's feature map instead of the dense box map
            else if (o.box[l].t >= 0) rd.push_back(o.box[l]);
            if (o.cls[l].t >= 0) rd.push_back(o.cls[l]);
            if (o.sparse_cf) rd.push_back(e.ops[o.hb_cf[l][0]].in);
            else if (o.cf[l].t >= 0) rd.push_back(o.cf[l]);
            if (o.amax[l].t >= 0) rd.push_back(o.amax[l]);
        }
}

// Pure host step: the launch order with its cross-lane waits / records for the current plan. Kept apart from the HIP calls so
// that (i) it runs once per plan instead of once per capture and (ii) the CPU-only sanitizer build can exercise it.
static int build_lane_schedule(yp_engine& e) {
    const size_t n = e.ops.size();
    int nl = 1;
    for (const Op& o : e.ops) nl = std::max(nl, o.lane + 1);
    e.n_lanes = nl;
    e.lane_steps.clear();
    e.lanes_used.clear();
    std::vector<std::vector<View>> rds(n), wrs(n);
    for (size_t i = 0; i < n; ++i) op_views(e, e.ops[i], rds[i], wrs[i]);
    std::vector<char> recorded(n, 0), forked(nl, 0);
    forked[0] = 1;
    for (size_t i = 0; i < n; ++i) {
        const Op& o = e.ops[i];
        if (o.skip) continue;
        yp_engine::LaneStep stp;
        stp.op = (int)i;
        // cross-lane dependencies: latest conflicting op of every other lane
        std::vector<int> need(nl, -1);
        for (size_t j = 0; j < i; ++j) {
            const Op& q = e.ops[j];
            if (q.lane == o.lane || q.skip) continue;
            bool dep = false;
            for (const View& w : wrs[j]) {
                for (const View& r : rds[i]) dep |= views_overlap(w, r);     // RAW
                for (const View& w2 : wrs[i]) dep |= views_overlap(w, w2);   // WAW
            }
            for (const View& r : rds[j])
                for (const View& w2 : wrs[i]) dep |= views_overlap(r, w2);   // WAR
            if (dep) need[q.lane] = (int)j;
        }
        for (int l = 0; l < nl; ++l) {
            if (need[l] < 0) continue;
            if (!recorded[need[l]]) return fail(YP_ERR_STATE, "internal: dependency %s -> %s was not recorded", e.ops[need[l]].name.c_str(), o.name.c_str());
            stp.waits.push_back(need[l]);
        }
        // A side lane joins the capture through its first wait on an event of a lane that is already part of it. A first op
        // without any such dependency (it reads the caller's frames only, say) would otherwise run OUTSIDE the capture: fork it
        // explicitly from lane 0.
        if (!forked[o.lane]) {
            bool via_wait = false;
            for (int j : stp.waits) via_wait |= forked[e.ops[j].lane] != 0;
            stp.fork = !via_wait;
            forked[o.lane] = 1;
            e.lanes_used.push_back(o.lane);
        }
        // record after this op if a later op of another lane conflicts with it
        bool later = false;
        for (size_t k = i + 1; k < n && !later; ++k) {
            const Op& q = e.ops[k];
            if (q.lane == o.lane || q.skip) continue;
            for (const View& w : wrs[i]) {
                for (const View& r : rds[k]) later |= views_overlap(w, r);
                for (const View& w2 : wrs[k]) later |= views_overlap(w, w2);
            }
            for (const View& r : rds[i])
                for (const View& w2 : wrs[k]) later |= views_overlap(r, w2);
        }
        stp.record = later;
        if (later) recorded[i] = 1;
        e.lane_steps.push_back(std::move(stp));
    }
    return YP_OK;
}

// Capture of the multi-lane forward as a DAG, on ONE stream and without a single event.
//
// Rounds 1-3 captured the lanes as streams: a cross-lane dependency was hipEventRecord on the producer's stream + hipStreamWaitEvent on the
// consumer's, with one set of events kept for the engine's life and re-used by every capture. A replay of a RE-captured graph of that
// kind died with a host SIGSEGV inside hipGraphLaunch (round 3, tests/test_gpu_fullsize.py: always with the caller on the legacy NULL
// stream, one run in three otherwise; Python-level stack only, no native backtrace was ever obtained). What the crashing runs share and
// no passing configuration had is cross-stream capture state that outlives a capture: events recorded inside capture #1 (and side
// streams that were pulled into it) entering capture #2. The hypothesis - an event whose record finds nothing new to attach keeps the
// node handles of the destroyed graph #1 - was put to a stand-alone probe (tools/micro/recapture_probe.hip: 6 lanes, 20 re-captures per
// mode with re-used events, fresh events, an early fork, NULL-stream replays): every mode passes on ROCm 7.2, so the fault is NOT
// reproduced in isolation and its cause inside the runtime remains unestablished. What can be done is to take the whole mechanism away:
//   RULE: no event and no second stream ever takes part in a capture. The op list is captured on own_stream alone; before each op the
//   stream's dependency set is REPLACED (hipStreamUpdateCaptureDependencies, hipStreamSetCaptureDependencies) by the graph nodes the lane
//   schedule names - the tail of the op's own lane and the tails of the ops it waits for - and after the op the new tail is read back
//   (hipStreamGetCaptureInfo_v2). Node handles belong to the graph under construction and die with it; nothing outlives a capture,
//   so capture #2 cannot see anything of capture #1.
// The resulting graph has the same edges the stream form produced (yp_debug_graph_info reports nodes / edges against the schedule's count;
// tests/test_gpu_ring.py holds replays of re-captured graphs, on alternating streams and on the NULL stream, to the eager results).
// The walk both capture_dag and the host selftest use: `launch(step, deps, tail)` gets the dependency set of the step (sorted, unique) and
// returns the tail its launches leave behind; `finish(all_tails)` gets the union of the lanes' tails.
template <class Node, class Launch, class Finish>
static int walk_lane_dag(const yp_engine& e, Launch launch, Finish finish, int* edges_out) {
    const size_t n = e.ops.size();
    const int nl = e.n_lanes;
    std::vector<std::vector<Node>> lane_tail(nl), op_tail(n);
    std::vector<char> started(nl, 0);
    int edges = 0;
    for (const yp_engine::LaneStep& stp : e.lane_steps) {
        const Op& o = e.ops[stp.op];
        std::vector<Node> deps;
        if (started[o.lane]) deps = lane_tail[o.lane];
        else if (stp.fork) deps = lane_tail[0];                       // a side lane whose first op depends on nothing: behind lane 0 as it stands
        for (int j : stp.waits) deps.insert(deps.end(), op_tail[j].begin(), op_tail[j].end());
        std::sort(deps.begin(), deps.end());
        deps.erase(std::unique(deps.begin(), deps.end()), deps.end());
        int rc = launch(stp, deps, lane_tail[o.lane]);
        if (rc != YP_OK) return rc;
        started[o.lane] = 1;
        if (stp.record) op_tail[stp.op] = lane_tail[o.lane];
        edges += (int)deps.size();
    }
    std::vector<Node> all;
    for (int l = 0; l < nl; ++l) if (started[l]) all.insert(all.end(), lane_tail[l].begin(), lane_tail[l].end());
    std::sort(all.begin(), all.end());
    all.erase(std::unique(all.begin(), all.end()), all.end());
    if (edges_out) *edges_out = edges;                                // edges INTO the first node of every op (an op of several kernels adds its inner chain)
    return finish(all);
}

static int capture_dag(yp_engine& e, const RunArgs& a) {
    hipStream_t cs = e.own_stream;
    bool first = true;
    auto launch = [&](const yp_engine::LaneStep& stp, std::vector<hipGraphNode_t>& deps, std::vector<hipGraphNode_t>& tail) -> int {
        const Op& o = e.ops[stp.op];
        // (the very first op starts from the empty set a fresh capture has)
        if (!first) HIPCHK(hipStreamUpdateCaptureDependencies(cs, deps.empty() ? nullptr : deps.data(), deps.size(), hipStreamSetCaptureDependencies));
        first = false;
        hipError_t err = run_op(e, o, a, cs);
        if (err != hipSuccess) return fail(YP_ERR_HIP, "launch of op '%s' failed: %s", o.name.c_str(), hipGetErrorString(err));
        hipStreamCaptureStatus stt = hipStreamCaptureStatusNone;
        unsigned long long id = 0;
        hipGraph_t g = nullptr;
        const hipGraphNode_t* dn = nullptr;
        size_t nd = 0;
        HIPCHK(hipStreamGetCaptureInfo_v2(cs, &stt, &id, &g, &dn, &nd));
        if (stt != hipStreamCaptureStatusActive) return fail(YP_ERR_HIP, "capture invalidated at op '%s'", o.name.c_str());
        tail.assign(dn, dn + nd);
        return YP_OK;
    };
    // the capture ends behind every lane's tail (nothing to join: there is only this stream)
    auto finish = [&](std::vector<hipGraphNode_t>& all) -> int {
        HIPCHK(hipStreamUpdateCaptureDependencies(cs, all.empty() ? nullptr : all.data(), all.size(), hipStreamSetCaptureDependencies));
        return YP_OK;
    };
    return walk_lane_dag<hipGraphNode_t>(e, launch, finish, &e.g_edges_expected);
}

static int run_all(yp_engine& e, const RunArgs& a, hipStream_t st) {
    for (const Op& o : e.ops) {
        if (o.skip) continue;
        hipError_t err = run_op(e, o, a, st);
        if (err != hipSuccess) return fail(YP_ERR_HIP, "launch of op '%s' failed: %s", o.name.c_str(), hipGetErrorString(err));
    }
    return YP_OK;
}

// ---------------------------------------------------------------------------------------------------------
// weight packing
// ---------------------------------------------------------------------------------------------------------
static void put(std::vector<unsigned char>& buf, size_t idx, float v, int dtype) {
    if (dtype == DT_BF16) { uint16_t h = f2bf(v); memcpy(&buf[idx * 2], &h, 2); }
    else memcpy(&buf[idx * 4], &v, 4);
}

// Host half of the weight hand-over: repack one folded fp32 parameter into the kernel layout (pure host arithmetic, covered by the
// CPU sanitizer build). main = the packed matrix / tap table, aux = the stem's second (GEMM) layout, else empty.
static void pack_weight(const yp_engine& e, WeightDesc& w, std::vector<unsigned char>& main, std::vector<unsigned char>& aux) {
    const int es = e.es();
    main.clear(); aux.clear();
    if (w.is_stem) {
        // [ky][kx][c_bgr][co] fp32, values rounded to the engine dtype; BGR memory order <- RGB weight order
        std::vector<float> f((size_t)27 * w.cout);
        for (int co = 0; co < w.cout; ++co)
            for (int ci = 0; ci < 3; ++ci)
                for (int ky = 0; ky < 3; ++ky)
                    for (int kx = 0; kx < 3; ++kx) {
                        float v = w.w[((size_t)(co * 3 + ci) * 3 + ky) * 3 + kx];
                        if (e.dtype == DT_BF16) v = bf2f(f2bf(v));
                        f[((size_t)(ky * 3 + kx) * 3 + (2 - ci)) * w.cout + co] = v;
                    }
        main.resize(f.size() * 4);
        memcpy(main.data(), f.data(), main.size());
        if (e.dtype == DT_BF16) {   // GEMM layout for the MFMA stem: [co][k=(ky,kx,c_bgr)] bf16, K padded 27 -> 32
            std::vector<uint16_t> g((size_t)w.cout * 32, 0);
            for (int co = 0; co < w.cout; ++co)
                for (int k = 0; k < 27; ++k) g[(size_t)co * 32 + k] = f2bf(f[(size_t)k * w.cout + co]);
            aux.resize(g.size() * 2);
            memcpy(aux.data(), g.data(), aux.size());
        }
    } else if (w.groups > 1) {
        // depthwise: [k*k][C]
        const int C = w.cout, kk = w.k * w.k;
        main.assign((size_t)kk * C * es, 0);
        for (int c = 0; c < C; ++c)
            for (int t = 0; t < kk; ++t) put(main, (size_t)t * C + c, w.w[(size_t)c * kk + t], e.dtype);
    } else if (w.transposed) {
        // ConvTranspose2d k2 s2: weight [Cin][Cout][2][2] -> 4 GEMM matrices [CoutPad][Kpad], K = Cin
        const int cin = w.cin_g, cout = w.cout;
        w.Kpad = (cin + 31) / 32 * 32;
        const size_t rows = (size_t)(cout + 127) / 128 * 128, sub = rows * w.Kpad;
        w.mat_bytes = sub * es;
        main.assign(sub * 4 * es, 0);
        for (int dy = 0; dy < 2; ++dy)
            for (int dx = 0; dx < 2; ++dx)
                for (int co = 0; co < cout; ++co)
                    for (int ci = 0; ci < cin; ++ci)
                        put(main, sub * (dy * 2 + dx) + (size_t)co * w.Kpad + ci, w.w[(((size_t)ci * cout + co) * 2 + dy) * 2 + dx], e.dtype);
    } else {
        // dense: [Cout][Cin][k][k] -> [CoutPad128][Kpad], k order (ky,kx,ci)
        const int cin = w.cin_g, cout = w.cout, k = w.k, cp = w.cin_pad > 0 ? w.cin_pad : cin;
        const int K = k * k * cp;
        w.Kpad = (K + 31) / 32 * 32;
        const size_t rows = (size_t)(cout + 127) / 128 * 128;
        w.mat_bytes = rows * w.Kpad * es;
        main.assign(rows * w.Kpad * es, 0);
        for (int co = 0; co < cout; ++co)
            for (int ci = 0; ci < cin; ++ci)
                for (int ky = 0; ky < k; ++ky)
                    for (int kx = 0; kx < k; ++kx)
                        put(main, (size_t)co * w.Kpad + (size_t)(ky * k + kx) * cp + ci, w.w[(((size_t)co * cin + ci) * k + ky) * k + kx], e.dtype);
    }
}

static int upload_weight(yp_engine& e, WeightDesc& w) {
    std::vector<unsigned char> main, aux;
    pack_weight(e, w, main, aux);
    HIPCHK(hipMalloc(&w.d_w, main.size()));
    HIPCHK(hipMemcpy(w.d_w, main.data(), main.size(), hipMemcpyHostToDevice));
    if (!aux.empty()) {
        HIPCHK(hipMalloc(&w.d_w2, aux.size()));
        HIPCHK(hipMemcpy(w.d_w2, aux.data(), aux.size(), hipMemcpyHostToDevice));
    }
    HIPCHK(hipMalloc((void**)&w.d_b, (size_t)w.cout * 4));
    HIPCHK(hipMemcpy(w.d_b, w.b.data(), (size_t)w.cout * 4, hipMemcpyHostToDevice));
    std::vector<float>().swap(w.w);
    return YP_OK;
}

static void weight_shape(const WeightDesc& w, int64_t s[4]) {
    if (w.transposed) { s[0] = w.cin_g; s[1] = w.cout; }
    else { s[0] = w.cout; s[1] = w.cin_g; }
    s[2] = w.k; s[3] = w.k;
}

}  // namespace

// =========================================================================================================
// C-ABI
// =========================================================================================================
extern "C" {

const char* yp_last_error(void) { return g_err; }

// Fatal-signal aid, on by default (YOLOP_SEGV_TRACE=0 turns it off): a native backtrace on stderr when the process takes a
// SIGSEGV / SIGABRT / SIGBUS (glibc's heap checks end in abort()), then the default action. Async-signal-safe calls only.
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
static struct sigaction g_prev_sa[3];
static const int g_fatal_sigs[3] = {SIGSEGV, SIGABRT, SIGBUS};
static int g_fatal_fd = -1;                      // YOLOP_SEGV_FILE=<path>: the native backtrace is appended there as well
static void yp_fatal_handler(int sig) {
    void* frames[64];
    const int n = backtrace(frames, 64);
    const char* msg = sig == SIGSEGV ? "\n[yolop] SIGSEGV - native backtrace:\n" : sig == SIGABRT ? "\n[yolop] SIGABRT - native backtrace:\n" : "\n[yolop] SIGBUS - native backtrace:\n";
    (void)!write(2, msg, strlen(msg));
    backtrace_symbols_fd(frames, n, 2);
    if (g_fatal_fd >= 0) {                        // (a test runner that captures fd 2 loses the lines above with the process)
        (void)!write(g_fatal_fd, msg, strlen(msg));
        backtrace_symbols_fd(frames, n, g_fatal_fd);
    }
    // hand over to whoever was installed before us (Python's faulthandler prints the interpreter stack), else the default action
    for (int i = 0; i < 3; ++i)
        if (g_fatal_sigs[i] == sig) (void)sigaction(sig, &g_prev_sa[i], nullptr);
    raise(sig);
}
static void install_fatal_handlers() {
    static bool done = false;
    if (done) return;
    done = true;
    const char* tr = std::getenv("YOLOP_SEGV_TRACE");
    if (tr && *tr == '0') return;
    if (const char* f = std::getenv("YOLOP_SEGV_FILE")) g_fatal_fd = open(f, O_WRONLY | O_CREAT | O_APPEND, 0644);
    void* warm[2];
    (void)backtrace(warm, 2);                    // loads libgcc now: the first backtrace() call allocates, which a handler must not
    for (int i = 0; i < 3; ++i) {
        struct sigaction sa;
        memset(&sa, 0, sizeof(sa));
        sa.sa_handler = yp_fatal_handler;
        sigemptyset(&sa.sa_mask);
        sa.sa_flags = SA_NODEFER | SA_ONSTACK;
        if (sigaction(g_fatal_sigs[i], &sa, &g_prev_sa[i]) != 0) g_prev_sa[i].sa_handler = SIG_DFL;
    }
}

int yp_create(const yp_model_desc* desc, int device, yp_engine** out) {
    if (!desc || !out) return fail(YP_ERR_ARG, "null argument");
    if (desc->nc <= 0 || desc->max_det <= 0 || desc->max_det > 1024) return fail(YP_ERR_ARG, "bad nc/max_det");
    if (desc->task == YP_TASK_SEGMENT && desc->max_det > YP_MAX_MASKS)
        return fail(YP_ERR_ARG, "segmentation engines take max_det <= %d (the mask tail keeps one frame's coefficients in LDS; ultralytics' default is 300)", YP_MAX_MASKS);
    if (desc->dtype != YP_BF16 && desc->dtype != YP_F32) return fail(YP_ERR_ARG, "bad dtype");
    if (desc->task != YP_TASK_DETECT && desc->task != YP_TASK_SEGMENT) return fail(YP_ERR_ARG, "bad task");
    install_fatal_handlers();
    std::unique_ptr<yp_engine> e(new yp_engine());
    e->desc = *desc; e->device = device; e->dtype = desc->dtype;
    { const char* nf = std::getenv("YOLOP_NO_FUSE"); e->fuse = !(nf && *nf == '1'); }
    { const char* tf = std::getenv("YOLOP_TAIL"); e->tail = tf && *tf == '1'; }
    { const char* dh = std::getenv("YOLOP_DENSE_HEAD"); e->sparse_head = !(dh && *dh == '1'); }
    int rc = build_graph(*e);
    if (rc != YP_OK) return rc;
    for (const Op& o : e->ops)
        for (const View* v : {&o.in, &o.out, &o.res})
            if (v->t >= 0 && o.kind != OP_STEM && ((v->C & 7) || (v->coff & 7)) && !e->tensors[v->t].f32)
                return fail(YP_ERR_ARG, "op %s: channel slice (%d,%d) is not a multiple of 8", o.name.c_str(), v->coff, v->C);
    *out = e.release();
    return YP_OK;
}

int yp_destroy(yp_engine* e) {
    if (!e) return YP_OK;
    if (e->finalized || e->arena) { (void)hipSetDevice(e->device); (void)hipDeviceSynchronize(); }   // nothing of this engine may still be running
    for (auto& w : e->weights) { if (w.d_w) (void)hipFree(w.d_w); if (w.d_w2) (void)hipFree(w.d_w2); if (w.d_b) (void)hipFree(w.d_b); }
    if (e->arena) (void)hipFree(e->arena);
    if (e->mask_ws) (void)hipFree(e->mask_ws);
    if (e->head_ws) (void)hipFree(e->head_ws);
    if (e->nms_ws) (void)hipFree(e->nms_ws);
    if (e->d_nms) (void)hipFree(e->d_nms);
    if (e->sp_ws) (void)hipFree(e->sp_ws);
    if (e->o_det) { (void)hipFree(e->o_det); (void)hipFree(e->o_idx); (void)hipFree(e->o_coeff); }
    drop_graph(*e);
    if (e->ev_in) (void)hipEventDestroy(e->ev_in);
    if (e->ev_done) (void)hipEventDestroy(e->ev_done);
    if (e->own_stream) (void)hipStreamDestroy(e->own_stream);
    delete e;
    return YP_OK;
}

int yp_weight_count(const yp_engine* e) { return e ? (int)e->weights.size() * 2 : fail(YP_ERR_ARG, "null engine"); }

int yp_weight_info(const yp_engine* e, int i, char* name, int cap, int64_t shape[4], int* ndim) {
    if (!e || i < 0 || i >= (int)e->weights.size() * 2) return fail(YP_ERR_ARG, "bad weight index");
    const WeightDesc& w = e->weights[i / 2];
    const bool is_bias = i & 1;
    if (name && cap > 0) snprintf(name, cap, "%s.%s", w.name.c_str(), is_bias ? "bias" : "weight");
    if (is_bias) { if (shape) { shape[0] = w.cout; shape[1] = shape[2] = shape[3] = 1; } if (ndim) *ndim = 1; }
    else { if (shape) weight_shape(w, shape); if (ndim) *ndim = 4; }
    return YP_OK;
}

int yp_set_weight(yp_engine* e, const char* name, const float* host, const int64_t* shape, int ndim) {
    if (!e || !name || !host || !shape) return fail(YP_ERR_ARG, "null argument");
    if (e->finalized) return fail(YP_ERR_STATE, "engine already finalized");
    std::string n(name);
    const bool is_bias = n.size() > 5 && n.compare(n.size() - 5, 5, ".bias") == 0;
    const bool is_w = n.size() > 7 && n.compare(n.size() - 7, 7, ".weight") == 0;
    if (!is_bias && !is_w) return fail(YP_ERR_WEIGHT, "parameter name '%s' must end in .weight or .bias", name);
    const std::string base = n.substr(0, n.size() - (is_bias ? 5 : 7));
    auto it = e->wmap.find(base);
    if (it == e->wmap.end()) return fail(YP_ERR_WEIGHT, "unknown parameter '%s'", name);
    WeightDesc& w = e->weights[it->second];
    if (is_bias) {
        if (ndim != 1 || shape[0] != w.cout) return fail(YP_ERR_WEIGHT, "'%s': expected shape [%d]", name, w.cout);
        w.b.assign(host, host + w.cout);
        w.have_b = true;
    } else {
        int64_t s[4];
        weight_shape(w, s);
        if (ndim != 4 || shape[0] != s[0] || shape[1] != s[1] || shape[2] != s[2] || shape[3] != s[3])
            return fail(YP_ERR_WEIGHT, "'%s': expected shape [%lld,%lld,%lld,%lld]", name, (long long)s[0], (long long)s[1], (long long)s[2], (long long)s[3]);
        w.w.assign(host, host + (size_t)(s[0] * s[1] * s[2] * s[3]));
        w.have_w = true;
    }
    return YP_OK;
}

int yp_finalize(yp_engine* e) {
    if (!e) return fail(YP_ERR_ARG, "null engine");
    if (e->finalized) return YP_OK;
    for (const auto& w : e->weights)
        if (!w.have_w || !w.have_b) return fail(YP_ERR_WEIGHT, "parameter '%s.%s' was never set", w.name.c_str(), w.have_w ? "bias" : "weight");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(YP_ERR_HIP, "no HIP device: the MI355X kernels cannot run here (no CPU fallback exists)");
    if (e->device < 0 || e->device >= ndev) return fail(YP_ERR_ARG, "device %d out of range (%d devices)", e->device, ndev);
    HIPCHK(hipSetDevice(e->device));
    for (auto& w : e->weights) {
        int rc = upload_weight(*e, w);
        if (rc != YP_OK) return rc;
    }
    HIPCHK(hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&e->ev_in, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&e->ev_done, hipEventDisableTiming));
    e->finalized = true;
    return YP_OK;
}

int yp_plan(yp_engine* e, int B, int H, int W) {
    if (!e) return fail(YP_ERR_ARG, "null engine");
    int rc = make_plan(*e, B, H, W);
    return rc == YP_OK ? (int)e->ops.size() : rc;
}

int yp_op_info(const yp_engine* e, int i, char* name, int cap, int* kind, double* flops, double* bytes) {
    if (!e || i < 0 || i >= (int)e->ops.size()) return fail(YP_ERR_ARG, "bad op index");
    const Op& o = e->ops[i];
    if (name && cap > 0) snprintf(name, cap, "%s", o.name.c_str());
    if (kind) *kind = o.kind;
    if (flops) *flops = o.flops;
    if (bytes) *bytes = o.bytes;
    return YP_OK;
}

int yp_op_kernel(const yp_engine* e, int i, char* name, int cap) {
    if (!e || i < 0 || i >= (int)e->ops.size() || !name || cap <= 0) return fail(YP_ERR_ARG, "bad argument");
    snprintf(name, cap, "%s", e->ops[i].kernel.c_str());
    return YP_OK;
}

int yp_op_fusion(const yp_engine* e, int i, int* pre, int* pre_stored) {
    if (!e || i < 0 || i >= (int)e->ops.size()) return fail(YP_ERR_ARG, "bad op index");
    const Op& o = e->ops[i];
    if (pre) *pre = o.fused7 ? o.pw_pre : -1;
    if (pre_stored) *pre_stored = (o.fused7 && o.pw_store) ? 1 : 0;
    return YP_OK;
}

int yp_op_output(const yp_engine* e, int i, int* tensor, int* coff, int* C) {
    if (!e || i < 0 || i >= (int)e->ops.size()) return fail(YP_ERR_ARG, "bad op index");
    const Op& o = e->ops[i];
    if (tensor) *tensor = o.out.t;
    if (coff) *coff = o.out.coff;
    if (C) *C = o.out.C;
    return YP_OK;
}

static int push_nms_params(yp_engine* e, hipStream_t st);
int yp_op_input(const yp_engine* e, int i, int* tensor, int* coff, int* C, int* c_read) {
    if (!e || i < 0 || i >= (int)e->ops.size()) return fail(YP_ERR_ARG, "bad op index");
    const Op& o = e->ops[i];
    if (tensor) *tensor = o.in.t;
    if (coff) *coff = o.in.coff;
    if (C) *C = o.in.C;
    if (c_read) *c_read = (o.kind == OP_CONV && o.widx >= 0 && e->weights[o.widx].cin_pad > o.in.C) ? e->weights[o.widx].cin_pad : o.in.C;
    return YP_OK;
}

int yp_run_op(yp_engine* e, int i, const uint8_t* in_dev, float* det_out, int32_t* idx_out, float* coeff_out, void* stream) {
    if (!e || i < 0 || i >= (int)e->ops.size()) return fail(YP_ERR_ARG, "bad op index");
    if (!e->allocated) return fail(YP_ERR_STATE, "no forward has run yet");
    HIPCHK(hipSetDevice(e->device));
    RunArgs a{in_dev, det_out, idx_out, coeff_out};
    if (push_nms_params(e, (hipStream_t)stream) != YP_OK) return YP_ERR_HIP;
    hipError_t err = run_op(*e, e->ops[i], a, (hipStream_t)stream);
    if (err != hipSuccess) return fail(YP_ERR_HIP, "op %s: %s", e->ops[i].name.c_str(), hipGetErrorString(err));
    return YP_OK;
}

int yp_tensor_write(yp_engine* e, int ti, int coff, int C, const float* host) {
    if (!e || ti < 0 || ti >= (int)e->tensors.size() || !host) return fail(YP_ERR_ARG, "bad argument");
    if (!e->allocated) return fail(YP_ERR_STATE, "no forward has run yet");
    const TensorDesc& t = e->tensors[ti];
    if (coff < 0 || C <= 0 || coff + C > t.C) return fail(YP_ERR_ARG, "bad channel slice");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipDeviceSynchronize());
    const size_t npix = (size_t)e->pB * t.H * t.W;
    const size_t es = tensor_elem_bytes(*e, t);
    std::vector<unsigned char> buf(npix * t.C * es);
    HIPCHK(hipMemcpy(buf.data(), t.ptr, buf.size(), hipMemcpyDeviceToHost));
    for (size_t p = 0; p < npix; ++p)
        for (int c = 0; c < C; ++c) {
            const float v = host[p * C + c];
            if (es == 4) memcpy(&buf[(p * t.C + coff + c) * 4], &v, 4);
            else { const uint16_t h = f2bf(v); memcpy(&buf[(p * t.C + coff + c) * 2], &h, 2); }
        }
    HIPCHK(hipMemcpy(t.ptr, buf.data(), buf.size(), hipMemcpyHostToDevice));
    return YP_OK;
}

int yp_tensor_count(const yp_engine* e) { return e ? (int)e->tensors.size() : fail(YP_ERR_ARG, "null engine"); }

int yp_tensor_info(const yp_engine* e, int i, char* name, int cap, int dims[4], int* is_f32) {
    if (!e || i < 0 || i >= (int)e->tensors.size()) return fail(YP_ERR_ARG, "bad tensor index");
    const TensorDesc& t = e->tensors[i];
    if (name && cap > 0) snprintf(name, cap, "%s", t.name.c_str());
    if (dims) { dims[0] = e->pB; dims[1] = t.H; dims[2] = t.W; dims[3] = t.C; }
    if (is_f32) *is_f32 = (t.f32 || e->dtype == DT_F32) ? 1 : 0;
    return YP_OK;
}

int yp_tensor_read(yp_engine* e, int i, float* host_out) {
    if (!e || i < 0 || i >= (int)e->tensors.size() || !host_out) return fail(YP_ERR_ARG, "bad argument");
    if (!e->allocated) return fail(YP_ERR_STATE, "no forward has run yet");
    const TensorDesc& t = e->tensors[i];
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipDeviceSynchronize());
    const size_t n = (size_t)e->pB * t.H * t.W * t.C;
    if (t.f32 || e->dtype == DT_F32) {
        HIPCHK(hipMemcpy(host_out, t.ptr, n * 4, hipMemcpyDeviceToHost));
    } else {
        std::vector<uint16_t> tmp(n);
        HIPCHK(hipMemcpy(tmp.data(), t.ptr, n * 2, hipMemcpyDeviceToHost));
        for (size_t j = 0; j < n; ++j) host_out[j] = bf2f(tmp[j]);
    }
    return YP_OK;
}

static void remember_tuning(yp_engine& e) {
    std::vector<yp_engine::Tuned> v;
    v.reserve(e.ops.size());
    for (const Op& o : e.ops) v.push_back({o.cfg, o.kernel});
    e.tuned[{e.pB, e.pH, e.pW}] = std::move(v);
}
static bool recall_tuning(yp_engine& e) {
    auto it = e.tuned.find({e.pB, e.pH, e.pW});
    if (it == e.tuned.end() || it->second.size() != e.ops.size()) return false;
    for (size_t i = 0; i < e.ops.size(); ++i) { e.ops[i].cfg = it->second[i].cfg; e.ops[i].kernel = it->second[i].kernel; }
    return true;
}

// Everything a forward needs that is NOT a launch on the caller's stream: plan, arena, tile configurations, lane schedule and
// resources, and - once per plan - one eager pass on the engine's own stream. That pass is what makes a later capture safe: the
// first launch of a kernel loads its code object and sets its LDS attribute, and none of that may happen while a stream is
// capturing (round-1 bench abort: the only caller whose FIRST forward was already in graph mode).
static int prepare(yp_engine* e, int B, int H, int W, const uint8_t* in, float* det) {
    if (!e || !in || !det) return fail(YP_ERR_ARG, "null argument");
    if (!e->finalized) return fail(YP_ERR_STATE, "yp_finalize has not been called");
    int rc = make_plan(*e, B, H, W);
    if (rc != YP_OK) return rc;
    if (e->allocated && e->warmed) return YP_OK;
    const bool fresh = !e->allocated;
    if (fresh) {
        HIPCHK(hipSetDevice(e->device));
        HIPCHK(hipDeviceSynchronize());          // the arena may move: nothing of the previous plan (or of the caller's input) may be in flight
        rc = allocate_plan(*e);
        if (rc != YP_OK) return rc;
        if (!recall_tuning(*e)) {
            e->tune_source = 0;
            if (e->tune && load_tune_cache(*e)) e->tune_source = 1;
            else if (e->tune && load_tune_cache(*e, true)) e->tune_source = 2;
            else if (e->tune) {
                e->tune_input = in;
                rc = autotune(*e);
                e->tune_input = nullptr;
                HIPCHK(hipDeviceSynchronize());
                if (rc != YP_OK) return rc;
                save_tune_cache(*e);
            }
            finish_kernel_names(*e);
            remember_tuning(*e);
        }
        rc = build_lane_schedule(*e);
        if (rc != YP_OK) return rc;
        e->warmed = false;
    }
    if (!e->warmed) {
        const bool seg = e->desc.task == YP_TASK_SEGMENT;
        RunArgs aw{in, e->o_det, e->o_idx, seg ? e->o_coeff : nullptr};
        HIPCHK(hipDeviceSynchronize());
        rc = run_all(*e, aw, e->own_stream);
        if (rc != YP_OK) return rc;
        HIPCHK(hipStreamSynchronize(e->own_stream));
        e->warmed = true;
    }
    return YP_OK;
}

__global__ void set_pair_kernel(float* dst, float a, float b) { dst[0] = a; dst[1] = b; }
static int push_nms_params(yp_engine* e, hipStream_t st) {
    if (!e->nms_dirty || !e->d_nms) return YP_OK;
    hipLaunchKernelGGL(set_pair_kernel, dim3(1), dim3(1), 0, st, e->d_nms, e->nms_conf, e->nms_iou);
    HIPCHK(hipGetLastError());
    e->nms_dirty = false;
    return YP_OK;
}

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// Stream contract of yp_forward. Forwards of ONE engine share its arena, workspaces and device-side parameter blocks, so they must execute
// one after the other. Calls on the same stream are ordered by the stream. A call on a DIFFERENT stream than the previous one first makes
// its stream wait for ev_done - the event every forward (eager or replay) records behind its last launch - so a warm-up on one stream
// followed by steps on another, or predict() under changing torch streams, cannot overlap two forwards of one engine. Cost: one
// hipEventRecord per forward, one hipStreamWaitEvent per stream change.
static int order_after_previous(yp_engine* e, hipStream_t st) {
    if (e->have_last && e->last_stream != st) HIPCHK(hipStreamWaitEvent(st, e->ev_done, 0));
    return YP_OK;
}
static int mark_done(yp_engine* e, hipStream_t ran_on, hipStream_t caller) {
    HIPCHK(hipEventRecord(e->ev_done, ran_on));
    e->last_stream = caller; e->have_last = true;
    return YP_OK;
}

static int forward_eager(yp_engine* e, const RunArgs& a, hipStream_t st) {
    int rc = order_after_previous(e, st);
    if (rc == YP_OK) rc = push_nms_params(e, st);
    if (rc == YP_OK) rc = run_all(*e, a, st);
    return rc != YP_OK ? rc : mark_done(e, st, st);
}

// hipGraph replay ON THE CALLER'S STREAM when that is not the legacy null stream (a replay on a stream of the engine's own, behind an event pair,
// left the GPU idle for ~20 us between steps and needed a copy kernel for the results). The graph is captured on the engine's own stream
// (capture needs a stream nothing else uses; capture_dag) but an executable graph launches on any stream. It is specialised on the plan,
// the INPUT pointer and - in direct mode - the output pointers.
// Lifetime rule: an executable is destroyed only (i) here, after hipEventSynchronize(ev_done) - ev_done was recorded behind the last launch
// on the very stream that launch ran on, whichever stream that was -, (ii) in allocate_plan / yp_tuning_import / yp_destroy after a device
// synchronisation, (iii) in yp_set_graph after the same hipEventSynchronize.
// The legacy NULL stream: hipGraphLaunch of a multi-branch graph on stream 0 faulted inside the runtime (ROCm 7.2, deterministic in
// tests/test_gpu_fullsize.py with the event-built graphs of round 3). A caller on stream 0 gets the engine's own stream in between: its
// stream-0 work -> ev_in -> replay on own_stream -> ev_done -> stream 0.
static int forward_replay(yp_engine* e, const uint8_t* in_dev, int B, int H, int W, float* det_out, int32_t* idx_out, float* coeff_out, hipStream_t st) {
    int rc;
    const bool seg = e->desc.task == YP_TASK_SEGMENT;
    float* const cf = (coeff_out && seg) ? coeff_out : nullptr;
    const bool same_plan = e->gexec && e->gkey.B == B && e->gkey.H == H && e->gkey.W == W && e->gkey.in == (const void*)in_dev;
    if (e->direct_out && same_plan && (e->gkey.det != det_out || e->gkey.idx != idx_out || e->gkey.coeff != cf) && ++e->out_changes >= 2)
        e->direct_out = false;                        // this caller rotates its output buffers: engine-owned results + copy-out from now on
    static const int own_mode = [] { const char* v = std::getenv("YOLOP_REPLAY_OWN_STREAM"); return v ? atoi(v) : 0; }();   // A/B switch: 1 = round 2's path, 2 = own stream + direct outputs
    const bool own = own_mode != 0 || st == nullptr;
    const bool direct = own_mode != 1 && e->direct_out && det_out && idx_out && (!seg || cf);
    RunArgs ag = direct ? RunArgs{in_dev, det_out, idx_out, cf} : RunArgs{in_dev, e->o_det, e->o_idx, seg ? e->o_coeff : nullptr};
    if (!same_plan || e->gkey.det != ag.det || e->gkey.idx != ag.idx || e->gkey.coeff != ag.coeff) {
        if (e->gexec) {
            // the previous executable may still be running (replays are asynchronous): never destroy it under the GPU
            if (e->have_last) HIPCHK(hipEventSynchronize(e->ev_done));
            drop_graph(*e);
        }
        hipGraph_t g = nullptr;
        HIPCHK(hipStreamBeginCapture(e->own_stream, hipStreamCaptureModeThreadLocal));
        rc = e->use_lanes ? capture_dag(*e, ag) : run_all(*e, ag, e->own_stream);
        hipError_t ce = hipStreamEndCapture(e->own_stream, &g);
        if (rc != YP_OK) { if (g) (void)hipGraphDestroy(g); return rc; }
        if (ce != hipSuccess) return fail(YP_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(ce));
        e->gsrc = g;
        {
            size_t nn = 0, ne = 0;
            HIPCHK(hipGraphGetNodes(g, nullptr, &nn));
            HIPCHK(hipGraphGetEdges(g, nullptr, nullptr, &ne));
            e->g_nodes = (int)nn; e->g_edges = (int)ne;
            if (!e->use_lanes) e->g_edges_expected = (int)nn - 1;
        }
        HIPCHK(hipGraphInstantiate(&e->gexec, g, nullptr, nullptr, 0));
        ++e->n_captures;
        e->gkey.B = B; e->gkey.H = H; e->gkey.W = W; e->gkey.in = in_dev; e->gkey.det = ag.det; e->gkey.idx = ag.idx; e->gkey.coeff = ag.coeff;
    }
    rc = order_after_previous(e, st);
    if (rc != YP_OK) return rc;
    hipStream_t rs = st;
    if (own) {
        rs = e->own_stream;
        HIPCHK(hipEventRecord(e->ev_in, st));
        HIPCHK(hipStreamWaitEvent(rs, e->ev_in, 0));
    }
    rc = push_nms_params(e, rs);
    if (rc != YP_OK) return rc;
    HIPCHK(hipGraphLaunch(e->gexec, rs));
    if (!direct) {
        const size_t rows = (size_t)B * e->desc.max_det;
        HIPCHK(launch_copy_out(e->o_det, det_out, e->o_idx, idx_out, e->o_coeff, cf, rows, rs));
    }
    rc = mark_done(e, rs, st);
    if (rc != YP_OK) return rc;
    if (own) HIPCHK(hipStreamWaitEvent(st, e->ev_done, 0));
    return YP_OK;
}

int yp_forward(yp_engine* e, const uint8_t* in_dev, int B, int H, int W, float* det_out, int32_t* idx_out,
               float* coeff_out, void* stream) {
    int rc = prepare(e, B, H, W, in_dev, det_out);
    if (rc != YP_OK) return rc;
    HIPCHK(hipSetDevice(e->device));
    hipStream_t st = (hipStream_t)stream;
    RunArgs a{in_dev, det_out, idx_out, coeff_out};
    if (!e->use_graph) return forward_eager(e, a, st);
    if (!e->graph_auto) return forward_replay(e, in_dev, B, H, W, det_out, idx_out, coeff_out, st);
    // auto: per plan, whichever of the two launch modes is faster ON THIS BOX for this shape. A replay pays a hop to the engine's stream,
    // a copy-out and the graph executor's per-node cost (measured ~8 us per node against ~6 us per eager launch on ROCm 7.2): with one
    // frame per call - the reference's call shape, yolo_seg/app.py:85-91 - its ~80 kernels are a few microseconds each and eager launches
    // win; at 32 frames the replay wins. Timed once per plan: 6 calls of each (same inputs, same outputs), host clock around a sync.
    auto it = e->auto_replay.find({B, H, W});
    if (it == e->auto_replay.end()) {
        double ms[2] = {0, 0};
        for (int mode = 0; mode < 2; ++mode) {
            for (int rep = 0; rep < 8; ++rep) {
                if (rep == 2) { HIPCHK(hipStreamSynchronize(st)); ms[mode] = -now_ms(); }
                rc = mode ? forward_replay(e, in_dev, B, H, W, det_out, idx_out, coeff_out, st) : forward_eager(e, a, st);
                if (rc != YP_OK) return rc;
            }
            HIPCHK(hipStreamSynchronize(st));
            ms[mode] += now_ms();
        }
        it = e->auto_replay.emplace(std::array<int, 3>{B, H, W}, ms[1] < ms[0]).first;
        static const bool say = [] { const char* v = std::getenv("YOLOP_VERBOSE"); return v && *v == '1'; }();
        if (say) fprintf(stderr, "[yolop] %dx%dx%d: eager %.3f ms, replay %.3f ms per call -> %s\n", B, H, W, ms[0] / 6, ms[1] / 6, it->second ? "replay" : "eager");
        return YP_OK;                                   // (the timed calls produced this call's outputs)
    }
    return it->second ? forward_replay(e, in_dev, B, H, W, det_out, idx_out, coeff_out, st) : forward_eager(e, a, st);
}

int yp_mask_contours(const uint8_t* masks_dev, int n, int H, int W, int strategy, int max_pts, int32_t* pts_out, int32_t* count_out, int32_t* parts_out,
                     int parts_cap, double* rect_out, void* stream) {
    if (n < 0 || H <= 0 || W <= 0 || max_pts < 2) return fail(YP_ERR_ARG, "yp_mask_contours: bad sizes (max_pts >= 2)");
    if (strategy != YP_CONTOURS_LARGEST && strategy != YP_CONTOURS_ALL) return fail(YP_ERR_ARG, "yp_mask_contours: strategy must be YP_CONTOURS_LARGEST or YP_CONTOURS_ALL");
    if (n > 0 && (!masks_dev || !pts_out || !count_out)) return fail(YP_ERR_ARG, "yp_mask_contours: null buffer");
    if (parts_out && parts_cap < 2) return fail(YP_ERR_ARG, "yp_mask_contours: parts_cap >= 2 with a parts buffer");
    if ((long)H * W >= (1l << 31)) return fail(YP_ERR_ARG, "yp_mask_contours: image too large");
    HIPCHK(launch_contours(masks_dev, n, H, W, strategy, max_pts, pts_out, count_out, parts_out, parts_cap, rect_out, (hipStream_t)stream));
    return YP_OK;
}

int yp_letterbox(const uint8_t* src_dev, int h0, int w0, uint8_t* dst_dev, int out_h, int out_w, int new_h, int new_w, int top, int left,
                 int pad_value, void* stream) {
    if (!src_dev || !dst_dev) return fail(YP_ERR_ARG, "yp_letterbox: null buffer");
    if (h0 <= 0 || w0 <= 0 || new_h <= 0 || new_w <= 0 || out_h <= 0 || out_w <= 0 || top < 0 || left < 0 || top + new_h > out_h ||
        left + new_w > out_w || pad_value < 0 || pad_value > 255)
        return fail(YP_ERR_ARG, "yp_letterbox: bad geometry %dx%d -> %dx%d at (%d,%d) in %dx%d", h0, w0, new_h, new_w, top, left, out_h, out_w);
    if ((long)h0 * w0 * 3 >= (1l << 31) || (long)out_h * out_w * 3 >= (1l << 31)) return fail(YP_ERR_ARG, "yp_letterbox: image too large");
    HIPCHK(launch_letterbox(src_dev, h0, w0, dst_dev, out_h, out_w, new_h, new_w, top, left, pad_value, (hipStream_t)stream));
    return YP_OK;
}

int yp_debug_head_clocks(uint64_t* out8) {
    if (!out8) return fail(YP_ERR_ARG, "yp_debug_head_clocks: null output");
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(head_read_clocks((unsigned long long*)out8));
    return YP_OK;
}

int yp_debug_head_branch_clocks(uint64_t* out8) {
    if (!out8) return fail(YP_ERR_ARG, "yp_debug_head_branch_clocks: null output");
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(head_branch_read_clocks((unsigned long long*)out8));
    return YP_OK;
}

int yp_debug_head_winners(yp_engine* e, int32_t* sel_host, float* box_host, float* coeff_host) {
    if (!e) return fail(YP_ERR_ARG, "null engine");
    if (!e->allocated) return fail(YP_ERR_STATE, "no forward has run yet");
    const Op* h = nullptr;
    for (const Op& o : e->ops) if (o.kind == OP_HEAD) h = &o;
    if (!h || !(h->sparse_box || h->sparse_cf) || !e->sp_ws) return 0;           // dense head: nothing to show
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipDeviceSynchronize());
    const SparseWs ws = sparse_ws_layout(*e);
    const char* base = (const char*)e->sp_ws;
    if (sel_host) HIPCHK(hipMemcpy(sel_host, base + ws.sel, (size_t)e->pB * HEAD_MAXK * 4, hipMemcpyDeviceToHost));
    if (box_host && h->sparse_box) HIPCHK(hipMemcpy(box_host, base + ws.box, (size_t)e->pB * e->desc.max_det * 64 * 4, hipMemcpyDeviceToHost));
    if (coeff_host && h->sparse_cf) HIPCHK(hipMemcpy(coeff_host, base + ws.cf, (size_t)e->pB * e->desc.max_det * 32 * 4, hipMemcpyDeviceToHost));
    return (h->sparse_box ? 1 : 0) | (h->sparse_cf ? 2 : 0);
}

int yp_tuning_source(const yp_engine* e) {
    if (!e) return YP_ERR_ARG;
    return e->tune_source;
}

int yp_debug_graph_info(const yp_engine* e, int64_t* out6) {
    if (!e || !out6) return fail(YP_ERR_ARG, "yp_debug_graph_info: null argument");
    out6[0] = e->n_captures; out6[1] = e->g_nodes; out6[2] = e->g_edges; out6[3] = e->g_edges_expected;
    out6[4] = (int64_t)e->lanes_used.size() + 1; out6[5] = e->gexec ? 1 : 0;
    return YP_OK;
}

int yp_debug_head_positions(yp_engine* e, int64_t* out6) {
    if (!e || !out6) return fail(YP_ERR_ARG, "yp_debug_head_positions: null argument");
    for (int i = 0; i < 6; ++i) out6[i] = 0;
    if (!e->allocated) return fail(YP_ERR_STATE, "no forward has run yet");
    const Op* h = nullptr;
    for (const Op& o : e->ops) if (o.kind == OP_HEAD) h = &o;
    if (!h || !(h->sparse_box || h->sparse_cf) || !e->sp_ws) return 0;
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipDeviceSynchronize());
    const SparseWs ws = sparse_ws_layout(*e);
    const char* base = (const char*)e->sp_ws;
    int pc[8];
    HIPCHK(hipMemcpy(pc, base + ws.pcount, sizeof(pc), hipMemcpyDeviceToHost));     // [4..7): the counts stage 2 saved before emptying the lists
    std::vector<int> wc((size_t)e->pB * 3);
    HIPCHK(hipMemcpy(wc.data(), base + ws.wcount, wc.size() * 4, hipMemcpyDeviceToHost));
    for (int l = 0; l < 3; ++l) {
        out6[l] = std::min(pc[4 + l], ws.plist_cap[l]);
        for (int b = 0; b < e->pB; ++b) out6[3 + l] += wc[(size_t)b * 3 + l];
    }
    return 1;
}

__global__ void op_marker_kernel(int* sink) { if (sink) *sink = 0; }
int yp_debug_marker(void* stream) {
    hipLaunchKernelGGL(op_marker_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (int*)nullptr);
    HIPCHK(hipGetLastError());
    return YP_OK;
}

int yp_debug_pwsp_clocks(uint64_t* out32) {
    if (!out32) return fail(YP_ERR_ARG, "yp_debug_pwsp_clocks: null output");
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(pwsp_read_clocks((unsigned long long*)out32));
    return YP_OK;
}

int yp_debug_contour_clocks(uint64_t* out12) {
    if (!out12) return fail(YP_ERR_ARG, "yp_debug_contour_clocks: null output");
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(contour_read_clocks((unsigned long long*)out12));
    return YP_OK;
}

int yp_debug_ablation(int v) {
    conv_set_debug_ablation(v);
    return 0;
}

int yp_debug_force_conv_cfg(int cfg) {
    conv_dma_force_cfg(cfg);
    return conv_dma_num_cfgs();
}

// Host-only walk over everything the executor computes for the current plan short of launching: parameter blocks of every op,
// exact kernel symbols, tune-cache round trip (when YOLOP_TUNE_CACHE is set), per-shape tuning memo, lane schedule. Needs no GPU;
// it exists so that the CPU sanitizer build (`make asan`, tools/asan_host.cpp) covers the executor's host code.
int yp_debug_host_selftest(yp_engine* e) {
    if (!e) return fail(YP_ERR_ARG, "null engine");
    if (!e->planned) return fail(YP_ERR_STATE, "yp_plan has not been called");
    size_t acc = 0;
    if (!e->finalized)
        for (WeightDesc& w : e->weights) {
            if (!w.have_w || !w.have_b || w.w.empty()) continue;
            std::vector<unsigned char> main, aux;
            pack_weight(*e, w, main, aux);
            acc += main.size() + aux.size();
        }
    for (const Op& o : e->ops) {
        if (o.skip) continue;
        if (o.kind == OP_CONV) {
            if (o.fused || o.fused6) acc += (size_t)dwpw_params(*e, o).Cout;
            else if (o.fused3) acc += (size_t)front_params(*e, o, nullptr).C2;
            else if (o.fused4) acc += (size_t)c2f_params(*e, o).Cout;
            else acc += (size_t)conv_params(*e, o).Cout;
        } else if (o.kind == OP_DWCONV && o.fused5) acc += (size_t)scd_params(*e, o).C;
        else if (o.fused7) acc += (size_t)pwsp_params(*e, o).C1;
    }
    finish_kernel_names(*e);
    save_tune_cache(*e);
    (void)load_tune_cache(*e);
    {   // a packaged table for this plan's key, if one is shipped, must be launchable by this build (every id passes the tuner's own
        // predicates for its layer); the plan keeps the configurations it had
        const std::string tp = tune_cache_path(*e, true);
        if (!tp.empty() && std::ifstream(tp).good()) {
            std::vector<std::pair<int, std::string>> keep;
            for (const Op& o : e->ops) keep.emplace_back(o.cfg, o.kernel);
            const bool ok = load_tune_cache(*e, true);
            for (size_t i = 0; i < e->ops.size(); ++i) { e->ops[i].cfg = keep[i].first; e->ops[i].kernel = keep[i].second; }
            if (!ok) return fail(YP_ERR_STATE, "packaged tune table %s does not apply to this build's plan", tp.c_str());
        }
    }
    finish_kernel_names(*e);
    remember_tuning(*e);
    if (!recall_tuning(*e)) return fail(YP_ERR_STATE, "internal: tuning memo lost");
    int rc = build_lane_schedule(*e);
    if (rc != YP_OK) return rc;
    // the schedule's own invariants: every wait names an op that records, every side lane that launches is joined exactly once
    std::vector<char> rec(e->ops.size(), 0);
    std::vector<char> seen(e->n_lanes, 0);
    seen[0] = 1;
    for (const auto& stp : e->lane_steps) {
        const Op& o = e->ops[stp.op];
        for (int j : stp.waits)
            if (!rec[j]) return fail(YP_ERR_STATE, "internal: %s waits on %s, which never records", o.name.c_str(), e->ops[j].name.c_str());
        if (!seen[o.lane]) {
            bool via = stp.fork;
            for (int j : stp.waits) via |= seen[e->ops[j].lane] != 0;
            if (!via) return fail(YP_ERR_STATE, "internal: lane %d would start outside the capture at %s", o.lane, o.name.c_str());
            seen[o.lane] = 1;
        }
        if (stp.record) rec[stp.op] = 1;
    }
    int used = 0;
    for (int l = 1; l < e->n_lanes; ++l) used += seen[l];
    if (used != (int)e->lanes_used.size()) return fail(YP_ERR_STATE, "internal: %d side lanes launch but %zu are joined", used, e->lanes_used.size());
    // the graph capture_dag builds from this schedule, walked with integer node ids: every dependency is an EARLIER node (so the graph is
    // acyclic and no handle of another capture can appear), an op without dependencies exists only before lane 0 has launched anything
    // (it reads the caller's frames alone), and the capture ends behind the tail of every lane that launched
    {
        int next_id = 0, lane0_nodes = 0, bad = 0, nsteps = 0;
        auto launch = [&](const yp_engine::LaneStep& stp, std::vector<int>& deps, std::vector<int>& tail) -> int {
            for (int d : deps) if (d < 0 || d >= next_id) ++bad;
            if (deps.empty() && lane0_nodes > 0) ++bad;
            if (e->ops[stp.op].lane == 0) ++lane0_nodes;
            tail.assign(1, next_id++);
            ++nsteps;
            return YP_OK;
        };
        size_t joined = 0;
        auto finish = [&](std::vector<int>& all) -> int { joined = all.size(); return YP_OK; };
        int edges = 0;
        rc = walk_lane_dag<int>(*e, launch, finish, &edges);
        if (rc != YP_OK) return rc;
        if (bad) return fail(YP_ERR_STATE, "internal: the capture DAG has %d dangling or forward dependencies", bad);
        if (joined != e->lanes_used.size() + 1 || nsteps != (int)e->lane_steps.size()) return fail(YP_ERR_STATE, "internal: the capture ends behind %zu lane tails, %zu lanes launched", joined, e->lanes_used.size() + 1);
    }
    return (int)e->lane_steps.size() + (acc == 0 ? 0 : 0);
}

int yp_set_nms(yp_engine* e, float conf, float iou) {
    if (!e) return fail(YP_ERR_ARG, "null engine");
    if (!(conf >= 0.f && conf < 1.f) || !(iou > 0.f && iou <= 1.f)) return fail(YP_ERR_ARG, "yp_set_nms: conf in [0,1), iou in (0,1]");
    if (conf == e->nms_conf && iou == e->nms_iou) return YP_OK;
    // The pair lives in device memory (a captured graph reads it through a pointer). It is rewritten by a one-thread kernel that the NEXT
    // forward enqueues in front of its own launches on the stream it runs on (yp_forward: push_nms_params): stream order puts it behind
    // every forward enqueued before and costs no synchronisation - predict(conf=0.25) and auto_segment's conf=0.9 on one engine
    // (yolo_seg/app.py:91, yolo_seg/yolo_with_deva.py:51) used to pay a device sync + blocking copy per call.
    e->nms_conf = conf; e->nms_iou = iou;
    e->nms_dirty = true;
    return YP_OK;
}

int yp_set_autotune(yp_engine* e, int enable) {
    if (!e) return fail(YP_ERR_ARG, "null engine");
    e->tune = enable != 0;
    e->tuned.clear();                    // (configurations remembered per shape were chosen under the other setting)
    return YP_OK;
}

int yp_tuning_export(const yp_engine* e, int32_t* cfg_out, int cap) {
    if (!e) return fail(YP_ERR_ARG, "null engine");
    if (!e->allocated) return fail(YP_ERR_STATE, "yp_tuning_export: no forward has run on the current plan");
    const int n = (int)e->ops.size();
    if (cfg_out) {
        if (cap < n) return fail(YP_ERR_ARG, "yp_tuning_export: %d ops, buffer holds %d", n, cap);
        for (int i = 0; i < n; ++i) cfg_out[i] = e->ops[i].cfg;
    }
    return n;
}

int yp_tuning_import(yp_engine* e, int B, int H, int W, const int32_t* cfg, int n) {
    if (!e || !cfg) return fail(YP_ERR_ARG, "null argument");
    if (!e->finalized) return fail(YP_ERR_STATE, "yp_finalize has not been called");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipDeviceSynchronize());                 // the plan (and with it every launch parameter) changes under nothing in flight
    e->planned = false;                             // re-derive the plan's fusion decisions from scratch for (B,H,W)
    int rc = make_plan(*e, B, H, W);
    if (rc != YP_OK) return rc;
    if (n != (int)e->ops.size()) return fail(YP_ERR_ARG, "yp_tuning_import: %d ids for a plan of %zu ops", n, e->ops.size());
    std::vector<int> v(cfg, cfg + n);
    if (!apply_tuning(*e, v.data(), n)) return fail(YP_ERR_ARG, "yp_tuning_import: a configuration id is not launchable for its layer in this build");
    finish_kernel_names(*e);
    remember_tuning(*e);                            // prepare() recalls it instead of tuning
    return YP_OK;
}

int yp_set_graph(yp_engine* e, int enable) {
    if (!e) return fail(YP_ERR_ARG, "null engine");
    if (enable < 0 || enable > 3) return fail(YP_ERR_ARG, "yp_set_graph: 0 eager, 1 replay with lanes, 2 replay without lanes, 3 auto");
    const bool lanes = enable != 2;      // 2 = graph without concurrent lanes (A/B measurements)
    if (enable != 0 && e->use_lanes != lanes && e->gexec) {
        // only a change of the LANE mode invalidates the captured executable; switching between eager launches and replay keeps it, so a
        // caller that alternates one-frame calls (eager) with batches (replay) pays neither a sync nor a re-capture
        HIPCHK(hipSetDevice(e->device));
        if (e->have_last) HIPCHK(hipEventSynchronize(e->ev_done));
        drop_graph(*e);
        e->auto_replay.clear();
    }
    e->use_graph = enable != 0;
    e->graph_auto = enable == 3;
    if (enable != 0) e->use_lanes = lanes;
    return YP_OK;
}

int yp_profile(yp_engine* e, const uint8_t* in_dev, int B, int H, int W, float* det_out, int32_t* idx_out,
               float* coeff_out, float* ms_out, int iters, void* stream) {
    int rc = prepare(e, B, H, W, in_dev, det_out);
    if (rc != YP_OK) return rc;
    if (!ms_out || iters <= 0) return fail(YP_ERR_ARG, "bad profile arguments");
    HIPCHK(hipSetDevice(e->device));
    hipStream_t st = (hipStream_t)stream;
    RunArgs a{in_dev, det_out, idx_out, coeff_out};
    const size_t n = e->ops.size();
    std::vector<hipEvent_t> ev(2 * n);
    for (auto& x : ev) HIPCHK(hipEventCreate(&x));
    std::vector<double> acc(n, 0.0);
    for (int it = 0; it < iters; ++it) {
        for (size_t i = 0; i < n; ++i) {
            HIPCHK(hipEventRecord(ev[2 * i], st));
            hipError_t err = e->ops[i].skip ? hipSuccess : run_op(*e, e->ops[i], a, st);
            if (err != hipSuccess) return fail(YP_ERR_HIP, "op %s: %s", e->ops[i].name.c_str(), hipGetErrorString(err));
            HIPCHK(hipEventRecord(ev[2 * i + 1], st));
        }
        HIPCHK(hipStreamSynchronize(st));
        for (size_t i = 0; i < n; ++i) {
            float ms = 0;
            HIPCHK(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]));
            acc[i] += ms;
        }
    }
    for (size_t i = 0; i < n; ++i) ms_out[i] = (float)(acc[i] / iters);
    for (auto& x : ev) (void)hipEventDestroy(x);
    return YP_OK;
}

int yp_proto(const yp_engine* e, const void** proto_dev, int* Hp, int* Wp) {
    if (!e || !proto_dev) return fail(YP_ERR_ARG, "null argument");
    if (e->proto_t < 0) return fail(YP_ERR_STATE, "engine was not created with YP_TASK_SEGMENT");
    if (!e->allocated) return fail(YP_ERR_STATE, "no forward has run yet");
    const TensorDesc& t = e->tensors[e->proto_t];
    *proto_dev = t.ptr;
    if (Hp) *Hp = t.H;
    if (Wp) *Wp = t.W;
    return YP_OK;
}

static int masks_common(yp_engine* e, int b, const float* coeff_dev, const float* boxes_dev, int n, int oh, int ow, int retina, int rh, int rw,
                        uint8_t* masks_out, int64_t* id_out, int32_t* kept_out, int suppress_small, int min_area, void* stream) {
    if (!e) return fail(YP_ERR_ARG, "null engine");
    if (e->proto_t < 0) return fail(YP_ERR_STATE, "engine was not created with YP_TASK_SEGMENT");
    if (!e->allocated) return fail(YP_ERR_STATE, "no forward has run yet");
    if (b < 0 || b >= e->pB || n < 0 || oh <= 0 || ow <= 0) return fail(YP_ERR_ARG, "bad mask arguments");
    if (n > 0 && (!coeff_dev || !boxes_dev)) return fail(YP_ERR_ARG, "null coefficient / box buffer");
    if (!retina && (oh != e->pH || ow != e->pW)) return fail(YP_ERR_ARG, "retina=0 masks are produced at the letterboxed input size %dx%d", e->pH, e->pW);
    if (kept_out && !id_out) return fail(YP_ERR_ARG, "kept_out needs id_out");
    if (n > YP_MAX_MASKS) return fail(YP_ERR_ARG, "at most %d masks per call (coefficients are LDS-resident)", YP_MAX_MASKS);
    HIPCHK(hipSetDevice(e->device));
    const TensorDesc& t = e->tensors[e->proto_t];
    MaskParams p{};
    p.proto = (const char*)t.ptr + (size_t)b * t.H * t.W * t.C * tensor_elem_bytes(*e, t);
    p.Hp = t.H; p.Wp = t.W; p.coeff = coeff_dev; p.boxes = boxes_dev; p.n = n; p.oh = oh; p.ow = ow;
    if (retina) {
        // scale_masks (A.7): gain=min(mh/oh,mw/ow); pad=((mw-ow*gain)/2,(mh-oh*gain)/2); crop [int(pad):int(m-pad)]
        const double gain = std::min((double)t.H / oh, (double)t.W / ow);
        const double padw = (t.W - ow * gain) / 2, padh = (t.H - oh * gain) / 2;
        p.t = (int)padh; p.l = (int)padw;
        p.ch = (int)(t.H - padh) - p.t; p.cw = (int)(t.W - padw) - p.l;
        p.bsx = 1.f; p.bsy = 1.f; p.crop_before = 0;
    } else {
        p.t = 0; p.l = 0; p.ch = t.H; p.cw = t.W;
        p.bsx = (float)t.W / (float)e->pW; p.bsy = (float)t.H / (float)e->pH; p.crop_before = 1;
    }
    p.masks = masks_out; p.ids = id_out; p.kept = kept_out; p.suppress_small = suppress_small; p.min_area = min_area;
    if (rh > 0 && (rh != oh || rw != ow)) {
        if (rw <= 0 || !id_out) return fail(YP_ERR_ARG, "the second resize needs a target size and id_out");
        if ((double)oh / rh > 16.0 || (double)ow / rw > 16.0) return fail(YP_ERR_ARG, "second resize shrinks by more than 16x");
        p.rh = rh; p.rw = rw;
    }
    {
        const size_t need = masks_workspace_bytes(p);
        if (need > e->mask_ws_bytes) {
            HIPCHK(hipStreamSynchronize((hipStream_t)stream));
            if (e->mask_ws) HIPCHK(hipFree(e->mask_ws));
            e->mask_ws = nullptr; e->mask_ws_bytes = 0;
            HIPCHK(hipMalloc(&e->mask_ws, need));
            e->mask_ws_bytes = need;
        }
        p.area = (int32_t*)e->mask_ws;
    }
    hipError_t err = launch_masks(p, e->dtype, (hipStream_t)stream);
    if (err != hipSuccess) return fail(YP_ERR_HIP, "mask kernels: %s", hipGetErrorString(err));
    return YP_OK;
}

int yp_masks(yp_engine* e, int b, const float* coeff_dev, const float* boxes_dev, int n, int oh, int ow, int retina,
             uint8_t* masks_out, int64_t* id_out, int32_t* kept_out, int suppress_small, int min_area, void* stream) {
    return masks_common(e, b, coeff_dev, boxes_dev, n, oh, ow, retina, 0, 0, masks_out, id_out, kept_out, suppress_small, min_area, stream);
}

int yp_id_mask_resized(yp_engine* e, int b, const float* coeff_dev, const float* boxes_dev, int n, int oh, int ow, int rh, int rw,
                       int64_t* id_out, int32_t* kept_out, int suppress_small, int min_area, void* stream) {
    if (rh <= 0 || rw <= 0) return fail(YP_ERR_ARG, "bad target size");
    return masks_common(e, b, coeff_dev, boxes_dev, n, oh, ow, 1, rh, rw, nullptr, id_out, kept_out, suppress_small, min_area, stream);
}

}  // extern "C"
