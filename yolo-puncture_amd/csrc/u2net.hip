// U^2-Net / U^2-Net-P behind the C-ABI (yp_u2net_*): the second per-frame network of the reference's video loop
// (`unet_predict(unet_model, cropped_frame)`, yolo_seg/app.py:184; model yolo_seg/tasks/models/U2Net.py:424-526, loader and
// post-process yolo_seg/tasks/unet_segment.py:32-73).
//
// Graph builder (host): RSU-7/6/5/4/4F blocks from the (in, mid, out) table of U2NETP / U2NET; every `torch.cat` is a channel-slice
// view of a buffer allocated up front (the bilinear up-sample and the skip's producer write straight into their halves), the
// block residual `hx1d + hxin` rides in the last convolution's epilogue (added after the ReLU, as in the reference).
// Kernels: dilated 3x3 conv + folded BatchNorm + ReLU = conv_igemm (matrix cores; fp32 mode = exact fp32 FMA chains), plus the
// small ones below - u8 BGR -> RGB/255 (channel-padded to 8), ceil-mode 2x2 max-pool, PyTorch-exact bilinear resize-to-size, and
// the tail: six side maps -> bilinear to full size -> 1x1 fusion -> sigmoid -> min/max -> normPRED -> mask.
#include "../../include/yolop.h"
#include "common.h"
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

using namespace yp;

extern "C" int yp_fail_public(int code, const char* msg);
static int u2fail(int code, const char* fmt, ...) {
    char buf[400];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    return yp_fail_public(code, buf);
}
#define U2HIP(x)                                                                                                   \
    do {                                                                                                           \
        hipError_t _e = (x);                                                                                       \
        if (_e != hipSuccess) return u2fail(YP_ERR_HIP, "%s failed: %s (%s:%d)", #x, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

namespace {

enum U2Kind { U2_INPUT = 0, U2_CONV = 1, U2_POOL = 2, U2_UP = 3 };

struct U2Tensor {
    std::string name;
    int C = 0, lvl = 0;
    bool f32 = false;      // fp32 regardless of the engine dtype (side maps)
    int H = 0, W = 0;
    size_t bytes = 0;
    void* ptr = nullptr;
};
struct U2Weight {
    std::string name;
    int cout = 0, cin = 0, cin_pad = 0, k = 3;
    bool have_w = false, have_b = false;
    std::vector<float> w, b;
    void* d_w = nullptr;
    float* d_b = nullptr;
    int Kpad = 0;
    size_t mat_bytes = 0;
};
struct U2Op {
    int kind = U2_CONV;
    std::string name;
    View in, out, res;
    int widx = -1, dil = 1, act = ACT_RELU;
    int impl = -1;         // convs: -1 = not yet chosen for this plan, 0 = conv_igemm, 1 = conv_small, 2 = conv_small taking the max pool /
                           // bilinear up-sample in front of it while loading, 3 = conv_halo_f32 (tuned in the plan's first pass)
    int pool_op = -1;      // convs: index of the pool op that produces this conv's input and feeds nothing else (graph pass)
    int up_op = -1;        // convs: index of the bilinear up-sample op that produces the first channels of this conv's input (graph pass)
    int consumer = -1;     // pools / up-samples: index of that conv; the op does not launch while the conv runs with impl 2
};

}  // namespace

struct yp_u2net {
    int variant = 'p', dtype = DT_F32, device = 0;
    std::vector<U2Tensor> tensors;
    std::vector<U2Weight> weights;
    std::vector<U2Op> ops;
    std::map<std::string, int> wmap;
    int side_t[6] = {-1, -1, -1, -1, -1, -1};
    int outconv_w = -1;
    bool finalized = false;
    // conv_small.hip vs conv_igemm.hip per convolution: small_max < 0 = timed per layer in a plan's first pass (default), otherwise
    // conv_small takes every layer it can run with at most small_max output pixels (YOLOP_U2_SMALL_MAX; 0 = never). Read at create.
    long small_max = -1;
    int pB = 0, pH = 0, pW = 0;
    void* arena = nullptr;
    size_t arena_bytes = 0;
    float* d_fuse = nullptr;            // [6 weights | bias | min bits | max bits]
    // hipGraph replay (the crop is launch-bound: ~170 small kernels): the graph reads an engine-owned copy of the frame and writes
    // engine-owned results, so neither the caller's input nor its output pointers are baked in; both copies ride the same stream
    bool use_graph = false, warmed = false;      // replay measured slower than eager launches on this net (2.29 vs 2.08 ms at 380^2): off by default
    hipStream_t own_stream = nullptr;
    hipEvent_t ev_in = nullptr, ev_out = nullptr;
    hipGraphExec_t gexec = nullptr;
    int gB = 0, gH = 0, gW = 0;
    uint8_t* in_buf = nullptr; float* o_prob = nullptr; float* o_norm = nullptr; uint8_t* o_mask = nullptr; size_t io_cap = 0;
    int es() const { return dtype == DT_BF16 ? 2 : 4; }
};

namespace {

struct StageCfg { int n; bool flat; int cin, mid, co; };      // n = 7,6,5,4 ; flat = RSU-4F

struct U2Builder {
    yp_u2net& e;
    explicit U2Builder(yp_u2net& en) : e(en) {}
    int tensor(const std::string& name, int C, int lvl, bool f32 = false) {
        U2Tensor t;
        t.name = name; t.C = C; t.lvl = lvl; t.f32 = f32;
        e.tensors.push_back(t);
        return (int)e.tensors.size() - 1;
    }
    View full(int t) const { return View{t, 0, e.tensors[t].C}; }
    int weight(const std::string& name, int cout, int cin, int k) {
        U2Weight w;
        w.name = name; w.cout = cout; w.cin = cin; w.cin_pad = (cin + 7) / 8 * 8; w.k = k;
        w.Kpad = (k * k * w.cin_pad + 31) / 32 * 32;
        w.mat_bytes = (size_t)((cout + 127) / 128 * 128) * w.Kpad * e.es();
        e.weights.push_back(w);
        e.wmap[name] = (int)e.weights.size() - 1;
        return (int)e.weights.size() - 1;
    }
    void conv(const std::string& name, View in, View out, int cin_logical, int dil, int act = ACT_RELU, View res = View{}) {
        U2Op o;
        o.kind = U2_CONV; o.name = name; o.in = in; o.out = out; o.res = res; o.dil = dil; o.act = act;
        o.widx = weight(name, out.C, cin_logical, 3);
        e.ops.push_back(o);
    }
    void pool(const std::string& name, View in, View out) {
        U2Op o;
        o.kind = U2_POOL; o.name = name; o.in = in; o.out = out;
        e.ops.push_back(o);
    }
    void up(const std::string& name, View in, View out) {
        U2Op o;
        o.kind = U2_UP; o.name = name; o.in = in; o.out = out;
        e.ops.push_back(o);
    }
    int lvl_of(View v) const { return e.tensors[v.t].lvl; }

    // RSU-n / RSU-4F (U2Net.py:30-314): `in` at level L -> `out` (a view, usually half of the next concat buffer)
    void rsu(const std::string& p, const StageCfg& s, View in, View out, int cin_logical) {
        const int L = lvl_of(in);
        const int tin = tensor(p + ".hxin", s.co, L);
        conv(p + ".rebnconvin", in, full(tin), cin_logical, 1);
        const int n = s.n;
        // cat_i (i = 1..n-1): [ first half: hx_n (deepest) or the up-sampled decoder map | second half: hx_i ]
        std::vector<int> cat(n, -1);
        for (int i = 1; i <= n - 1; ++i) cat[i] = tensor(p + ".cat" + std::to_string(i), 2 * s.mid, s.flat ? L : L + i - 1);
        auto hx = [&](int i) { return View{cat[i], s.mid, s.mid}; };
        static const int fdil[5] = {0, 1, 2, 4, 8};
        conv(p + ".rebnconv1", full(tin), hx(1), s.co, 1);
        for (int i = 2; i <= n - 1; ++i) {
            if (s.flat) conv(p + ".rebnconv" + std::to_string(i), hx(i - 1), hx(i), s.mid, fdil[i]);
            else {
                const int pt = tensor(p + ".pool" + std::to_string(i - 1), s.mid, L + i - 1);
                pool(p + ".pool" + std::to_string(i - 1), hx(i - 1), full(pt));
                conv(p + ".rebnconv" + std::to_string(i), full(pt), hx(i), s.mid, 1);
            }
        }
        conv(p + ".rebnconv" + std::to_string(n), hx(n - 1), View{cat[n - 1], 0, s.mid}, s.mid, s.flat ? 8 : 2);
        for (int i = n - 1; i >= 1; --i) {
            const bool last = i == 1;
            const int dil = s.flat ? fdil[i] : 1;
            if (last) {
                conv(p + ".rebnconv1d", full(cat[1]), out, 2 * s.mid, dil, ACT_RELU, full(tin));
            } else {
                const View dst{cat[i - 1], 0, s.mid};
                if (s.flat) conv(p + ".rebnconv" + std::to_string(i) + "d", full(cat[i]), dst, 2 * s.mid, dil);
                else {
                    const int dt = tensor(p + ".hx" + std::to_string(i) + "d", s.mid, L + i - 1);
                    conv(p + ".rebnconv" + std::to_string(i) + "d", full(cat[i]), full(dt), 2 * s.mid, 1);
                    up(p + ".up" + std::to_string(i), full(dt), dst);
                }
            }
        }
    }
};

static int build_u2net(yp_u2net& e) {
    static const StageCfg enc_p[6] = {{7, false, 3, 16, 64}, {6, false, 64, 16, 64}, {5, false, 64, 16, 64}, {4, false, 64, 16, 64}, {4, true, 64, 16, 64}, {4, true, 64, 16, 64}};
    static const StageCfg dec_p[5] = {{4, true, 128, 16, 64}, {4, false, 128, 16, 64}, {5, false, 128, 16, 64}, {6, false, 128, 16, 64}, {7, false, 128, 16, 64}};
    static const StageCfg enc_f[6] = {{7, false, 3, 32, 64}, {6, false, 64, 32, 128}, {5, false, 128, 64, 256}, {4, false, 256, 128, 512}, {4, true, 512, 256, 512}, {4, true, 512, 256, 512}};
    static const StageCfg dec_f[5] = {{4, true, 1024, 256, 512}, {4, false, 1024, 128, 256}, {5, false, 512, 64, 128}, {6, false, 256, 32, 64}, {7, false, 128, 16, 64}};
    if (e.variant != 'p' && e.variant != 'f') return u2fail(YP_ERR_ARG, "unknown U^2-Net variant '%c' ('p' = U2NETP, 'f' = U2NET)", e.variant);
    const StageCfg* enc = e.variant == 'p' ? enc_p : enc_f;
    const StageCfg* dec = e.variant == 'p' ? dec_p : dec_f;
    U2Builder B(e);
    const int x8 = B.tensor("input", 8, 0);
    {
        U2Op o;
        o.kind = U2_INPUT; o.name = "input"; o.out = B.full(x8);
        e.ops.push_back(o);
    }
    // decoder concat buffers: dcat[i] = [ up(deeper decoder map) | stage(i+1) output ], i = 0..4 (levels 0..4)
    int dcat[5];
    for (int i = 0; i < 5; ++i) {
        const int cup = (i == 4) ? enc[5].co : dec[3 - i].co;            // what is up-sampled into the first half
        dcat[i] = B.tensor("dec" + std::to_string(i + 1) + ".cat", cup + enc[i].co, i);
        if (cup + enc[i].co != dec[4 - i].cin) return u2fail(YP_ERR_STATE, "internal: decoder %d channel table", i + 1);
    }
    auto skip = [&](int i) { const int cup = e.tensors[dcat[i]].C - enc[i].co; return View{dcat[i], cup, enc[i].co}; };
    View cur = B.full(x8);
    int cin_logical = 3;
    const int h6 = B.tensor("stage6", enc[5].co, 5);
    for (int i = 0; i < 6; ++i) {
        const View out = i < 5 ? skip(i) : B.full(h6);
        B.rsu("stage" + std::to_string(i + 1), enc[i], cur, out, cin_logical);
        if (i < 5) {
            const int pt = B.tensor("pool" + std::to_string(i + 1) + std::to_string(i + 2), enc[i].co, i + 1);
            B.pool("pool" + std::to_string(i + 1) + std::to_string(i + 2), out, B.full(pt));
            cur = B.full(pt);
            cin_logical = enc[i].co;
        }
    }
    // decoder: stage5d .. stage1d
    int feat[6];                                  // hx1d .. hx5d, hx6 (side inputs)
    feat[5] = h6;
    View d = B.full(h6);
    for (int j = 0; j < 5; ++j) {
        const int i = 4 - j;                      // decoder stage i+1 works at level i
        B.up("up" + std::to_string(i + 2) + "to" + std::to_string(i + 1), d, View{dcat[i], 0, d.C});
        const int dt = B.tensor("stage" + std::to_string(i + 1) + "d", dec[j].co, i);
        B.rsu("stage" + std::to_string(i + 1) + "d", dec[j], B.full(dcat[i]), B.full(dt), dec[j].cin);
        feat[i] = dt;
        d = B.full(dt);
    }
    for (int k = 0; k < 6; ++k) {                 // side1..6: conv3x3 -> 1 channel, fp32, no activation
        const int st = B.tensor("side" + std::to_string(k + 1), 1, k, true);
        B.conv("side" + std::to_string(k + 1), B.full(feat[k]), B.full(st), e.tensors[feat[k]].C, 1, ACT_NONE);
        e.side_t[k] = st;
    }
    e.outconv_w = B.weight("outconv", 1, 6, 1);
    // graph pass: a max pool whose output is read by exactly one op, a convolution over the whole pooled tensor, may be taken by that
    // convolution while loading (conv_small's POOL form); whether it is, is decided per plan (u2_tune_op)
    for (size_t i = 0; i < e.ops.size(); ++i) {
        if (e.ops[i].kind != U2_POOL) continue;
        const int t = e.ops[i].out.t;
        int readers = 0, conv = -1;
        for (size_t j = 0; j < e.ops.size(); ++j) {
            const U2Op& o = e.ops[j];
            if (j != i && (o.in.t == t || o.res.t == t)) { ++readers; conv = (int)j; }
        }
        bool side = false;
        for (int k = 0; k < 6; ++k) side = side || e.side_t[k] == t;
        if (readers != 1 || side || conv < (int)i) continue;
        U2Op& c = e.ops[conv];
        if (c.kind != U2_CONV || c.in.t != t || c.in.coff != 0 || c.in.C != e.tensors[t].C || c.res.t == t) continue;
        if (e.ops[i].out.coff != 0 || e.ops[i].out.C != e.tensors[t].C) continue;
        c.pool_op = (int)i;
        e.ops[i].consumer = conv;
    }
    // likewise a bilinear up-sample that fills the FIRST channels of a concat buffer whose only reader of those channels is a convolution
    // over the whole buffer (the decoder convolutions of every RSU and the first convolution of every decoder stage)
    for (size_t i = 0; i < e.ops.size(); ++i) {
        if (e.ops[i].kind != U2_UP) continue;
        const View uo = e.ops[i].out;
        if (uo.coff != 0 || (uo.C & 15)) continue;
        int readers = 0, conv = -1;
        for (size_t j = 0; j < e.ops.size(); ++j) {
            const U2Op& o = e.ops[j];
            if (j == i) continue;
            if (o.in.t == uo.t && o.in.coff < uo.C) { ++readers; conv = (int)j; }
            if (o.res.t == uo.t && o.res.coff < uo.C) { ++readers; conv = -2; }
        }
        bool side = false;
        for (int k = 0; k < 6; ++k) side = side || e.side_t[k] == uo.t;
        if (readers != 1 || conv < (int)i || side) continue;
        U2Op& c = e.ops[conv];
        if (c.kind != U2_CONV || c.in.coff != 0 || c.in.C != e.tensors[uo.t].C || c.pool_op >= 0) continue;
        c.up_op = (int)i;
        e.ops[i].consumer = conv;
    }
    return YP_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void u2_input_kernel(const uint8_t* __restrict__ img, T* __restrict__ out, size_t npix) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const uint8_t* s = img + i * 3;
    // numpy2tensor (yolo_seg/utils/transform.py:15-20): BGR -> RGB, ToTensor = float(u8) / 255
    const float r = (float)s[2] / 255.0f, g = (float)s[1] / 255.0f, b = (float)s[0] / 255.0f;
    T* o = out + i * 8;
    o[0] = (T)r; o[1] = (T)g; o[2] = (T)b;
#pragma unroll
    for (int c = 3; c < 8; ++c) o[c] = (T)0.f;
}

// nn.MaxPool2d(2, stride=2, ceil_mode=True): windows clipped at the border
template <typename T>
__global__ __launch_bounds__(256) void u2_pool_kernel(const T* __restrict__ x, int xs, int xc, T* __restrict__ y, int ys, int yc, int B, int H, int W,
                                                     int Ho, int Wo, int C) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int cg = C >> 2;
    if (i >= (size_t)B * Ho * Wo * cg) return;
    const int c = (int)(i % cg) * 4;
    size_t r = i / cg;
    const int wo = (int)(r % Wo); r /= Wo;
    const int ho = (int)(r % Ho);
    const int b = (int)(r / Ho);
    float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            const int hi = ho * 2 + dy, wi = wo * 2 + dx;
            if (hi < H && wi < W) {
                const T* p = x + ((size_t)(b * H + hi) * W + wi) * xs + xc + c;
#pragma unroll
                for (int q = 0; q < 4; ++q) m[q] = fmaxf(m[q], (float)p[q]);
            }
        }
    T* o = y + ((size_t)(b * Ho + ho) * Wo + wo) * ys + yc + c;
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = (T)m[q];
}

// F.upsample(size=..., mode='bilinear') = upsample_bilinear2d, align_corners=False: src = scale*(dst+0.5)-0.5 clamped at 0
__device__ __forceinline__ void u2_bil(int dst, int n_in, int n_out, int& i0, int& i1, float& l0, float& l1) {
    const float scale = (float)n_in / (float)n_out;
    float f = scale * ((float)dst + 0.5f) - 0.5f;
    f = fmaxf(f, 0.f);
    i0 = (int)f;
    i1 = i0 + ((i0 < n_in - 1) ? 1 : 0);
    l1 = f - (float)i0;
    l0 = 1.f - l1;
}
template <typename T>
__global__ __launch_bounds__(256) void u2_up_kernel(const T* __restrict__ x, int xs, int xc, T* __restrict__ y, int ys, int yc, int B, int H, int W,
                                                   int Ho, int Wo, int C) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int cg = C >> 2;
    if (i >= (size_t)B * Ho * Wo * cg) return;
    const int c = (int)(i % cg) * 4;
    size_t r = i / cg;
    const int wo = (int)(r % Wo); r /= Wo;
    const int ho = (int)(r % Ho);
    const int b = (int)(r / Ho);
    T* o = y + ((size_t)(b * Ho + ho) * Wo + wo) * ys + yc + c;
    if (H == Ho && W == Wo) {                       // same size: PyTorch returns the values unchanged (weights are exactly 1 and 0)
        const T* p = x + ((size_t)(b * H + ho) * W + wo) * xs + xc + c;
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = p[q];
        return;
    }
    int y0, y1, x0, x1;
    float ly0, ly1, lx0, lx1;
    u2_bil(ho, H, Ho, y0, y1, ly0, ly1);
    u2_bil(wo, W, Wo, x0, x1, lx0, lx1);
    const T* p00 = x + ((size_t)(b * H + y0) * W + x0) * xs + xc + c;
    const T* p01 = x + ((size_t)(b * H + y0) * W + x1) * xs + xc + c;
    const T* p10 = x + ((size_t)(b * H + y1) * W + x0) * xs + xc + c;
    const T* p11 = x + ((size_t)(b * H + y1) * W + x1) * xs + xc + c;
#pragma unroll
    for (int q = 0; q < 4; ++q)
        o[q] = (T)(ly0 * (lx0 * (float)p00[q] + lx1 * (float)p01[q]) + ly1 * (lx0 * (float)p10[q] + lx1 * (float)p11[q]));
}

struct U2Tail {
    const float* side[6];
    int h[6], w[6];
    int B, H, W;
    const float* fuse;         // [6 weights | bias]
    unsigned* minmax;          // [min bits | max bits] of the positive floats
    float* prob;
};
// d0 = outconv(cat(d1, up(d2), ..., up(d6))) ; sigmoid (U2Net.py:498-520) ; running min / max for normPRED
__global__ __launch_bounds__(256) void u2_tail_kernel(const U2Tail t) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n = (size_t)t.B * t.H * t.W;
    float pr = 0.5f;
    const bool live = i < n;
    if (live) {
        size_t r = i;
        const int wo = (int)(r % t.W); r /= t.W;
        const int ho = (int)(r % t.H);
        const int b = (int)(r / t.H);
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const float* s = t.side[k] + (size_t)b * t.h[k] * t.w[k];
            float v;
            if (t.h[k] == t.H && t.w[k] == t.W) v = s[(size_t)ho * t.W + wo];
            else {
                int y0, y1, x0, x1;
                float ly0, ly1, lx0, lx1;
                u2_bil(ho, t.h[k], t.H, y0, y1, ly0, ly1);
                u2_bil(wo, t.w[k], t.W, x0, x1, lx0, lx1);
                v = ly0 * (lx0 * s[y0 * t.w[k] + x0] + lx1 * s[y0 * t.w[k] + x1]) + ly1 * (lx0 * s[y1 * t.w[k] + x0] + lx1 * s[y1 * t.w[k] + x1]);
            }
            acc = fmaf(t.fuse[k], v, acc);
        }
        acc += t.fuse[6];
        pr = 1.f / (1.f + expf(-acc));
        t.prob[i] = pr;
    }
    // sigmoid outputs are >= 0: their bit patterns order like unsigned integers
    unsigned lo = live ? __float_as_uint(pr) : 0x7f800000u, hi = live ? __float_as_uint(pr) : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = min(lo, (unsigned)__shfl_xor((int)lo, o));
        hi = max(hi, (unsigned)__shfl_xor((int)hi, o));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&t.minmax[0], lo);
        atomicMax(&t.minmax[1], hi);
    }
}
// normPRED (unet_segment.py:24-30) + `> 0.5 -> 255` (:66-71)
__global__ __launch_bounds__(256) void u2_norm_kernel(const float* __restrict__ prob, const unsigned* __restrict__ minmax, float* __restrict__ norm,
                                                     uint8_t* __restrict__ mask, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float mi = __uint_as_float(minmax[0]), ma = __uint_as_float(minmax[1]);
    const float dn = (prob[i] - mi) / (ma - mi);
    if (norm) norm[i] = dn;
    if (mask) mask[i] = dn > 0.5f ? 255 : 0;
}

static int plan_u2(yp_u2net& e, int B, int H, int W) {
    if (B <= 0 || H < 32 || W < 32) return u2fail(YP_ERR_ARG, "input must be [B,H,W,3] with H,W >= 32 (got %d,%d,%d)", B, H, W);
    if (e.pB == B && e.pH == H && e.pW == W && e.arena) return YP_OK;
    e.warmed = false;
    int lh[6], lw[6];
    lh[0] = H; lw[0] = W;
    for (int l = 1; l < 6; ++l) { lh[l] = (lh[l - 1] + 1) / 2; lw[l] = (lw[l - 1] + 1) / 2; }
    size_t total = 0;
    for (auto& t : e.tensors) {
        t.H = lh[t.lvl]; t.W = lw[t.lvl];
        t.bytes = (size_t)B * t.H * t.W * t.C * (t.f32 ? 4 : e.es());
        total += (t.bytes + 255) & ~(size_t)255;
    }
    if ((size_t)B * H * W * 128 * 4 >= (1ull << 31)) return u2fail(YP_ERR_ARG, "input too large for 32-bit tensor offsets");
    U2HIP(hipSetDevice(e.device));
    if (total > e.arena_bytes) {
        U2HIP(hipDeviceSynchronize());
        if (e.arena) U2HIP(hipFree(e.arena));
        e.arena = nullptr; e.arena_bytes = 0;
        U2HIP(hipMalloc(&e.arena, total));
        e.arena_bytes = total;
    }
    size_t off = 0;
    for (auto& t : e.tensors) { t.ptr = (char*)e.arena + off; off += (t.bytes + 255) & ~(size_t)255; }
    e.pB = B; e.pH = H; e.pW = W;
    for (auto& o : e.ops) o.impl = -1;
    if (e.gexec) { (void)hipDeviceSynchronize(); (void)hipGraphExecDestroy(e.gexec); e.gexec = nullptr; }
    return YP_OK;
}

template <typename T>
static hipError_t run_small(const yp_u2net& e, const U2Op& o, const uint8_t* img, hipStream_t st) {
    const int B = e.pB;
    if (o.kind == U2_INPUT) {
        const U2Tensor& to = e.tensors[o.out.t];
        const size_t npix = (size_t)B * to.H * to.W;
        hipLaunchKernelGGL(u2_input_kernel<T>, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, st, img, (T*)to.ptr, npix);
        return hipGetLastError();
    }
    const U2Tensor &ti = e.tensors[o.in.t], &to = e.tensors[o.out.t];
    const size_t n = (size_t)B * to.H * to.W * (o.out.C / 4);
    if (o.kind == U2_POOL)
        hipLaunchKernelGGL(u2_pool_kernel<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const T*)ti.ptr, ti.C, o.in.coff, (T*)to.ptr, to.C, o.out.coff,
                           B, ti.H, ti.W, to.H, to.W, o.out.C);
    else
        hipLaunchKernelGGL(u2_up_kernel<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const T*)ti.ptr, ti.C, o.in.coff, (T*)to.ptr, to.C, o.out.coff,
                           B, ti.H, ti.W, to.H, to.W, o.out.C);
    return hipGetLastError();
}

static ConvParams u2_conv_params(const yp_u2net& e, const U2Op& o, bool fuse_pool = false) {
    const U2Weight& w = e.weights[o.widx];
    const U2Tensor &ti = e.tensors[o.in.t], &to = e.tensors[o.out.t];
    ConvParams p{};
    p.x = ti.ptr; p.x_stride = ti.C; p.x_coff = o.in.coff; p.H = ti.H; p.W = ti.W; p.Cin = w.cin_pad;
    p.w = w.d_w; p.Kpad = w.Kpad; p.bias = w.d_b;
    p.y = to.ptr; p.y_stride = to.C; p.y_coff = o.out.coff; p.Ho = to.H; p.Wo = to.W; p.Cout = o.out.C;
    if (o.res.t >= 0) { p.res = e.tensors[o.res.t].ptr; p.res_stride = e.tensors[o.res.t].C; p.res_coff = o.res.coff; }
    p.M = e.pB * to.H * to.W; p.ks = 3; p.stride = 1; p.pad = o.dil; p.dil = o.dil; p.act = o.act;
    p.out_f32 = (to.f32 && e.dtype == DT_BF16) ? 1 : 0;
    p.up = 1; p.cfg = -1;
    p.x_bytes = ti.bytes; p.w_bytes = w.mat_bytes; p.y_bytes = to.bytes;
    if (fuse_pool && o.pool_op >= 0) {                 // read the pool's input instead; H, W stay the pooled size
        const U2Op& po = e.ops[o.pool_op];
        const U2Tensor& ts = e.tensors[po.in.t];
        p.x = ts.ptr; p.x_stride = ts.C; p.x_coff = po.in.coff; p.x_bytes = ts.bytes;
        p.pool_in = 1; p.src_H = ts.H; p.src_W = ts.W;
    } else if (fuse_pool && o.up_op >= 0) {            // the first channels come from the up-sample's low-resolution input
        const U2Op& uo = e.ops[o.up_op];
        const U2Tensor& ts = e.tensors[uo.in.t];
        p.x2 = ts.ptr; p.x2_bytes = ts.bytes; p.x2_stride = ts.C; p.x2_coff = uo.in.coff; p.x2_C = uo.out.C; p.x2_H = ts.H; p.x2_W = ts.W;
        p.up_bilinear = 1;
    }
    return p;
}

// untuned choice: the K-split kernel (conv_small.hip) up to `small_max` output pixels
static int u2_default_impl(const yp_u2net& e, const ConvParams& p) {
    const long cap = e.small_max >= 0 ? e.small_max : 40000L;
    return ((long)p.M <= cap && conv_small_valid(p, e.dtype)) ? 1 : 0;
}

static hipError_t run_u2_op(const yp_u2net& e, const U2Op& o, const uint8_t* img, hipStream_t st) {
    if ((o.kind == U2_POOL || o.kind == U2_UP) && o.consumer >= 0 && e.ops[o.consumer].impl == 2) return hipSuccess;   // taken by its consumer while loading
    if (o.kind != U2_CONV) return e.dtype == DT_BF16 ? run_small<__bf16>(e, o, img, st) : run_small<float>(e, o, img, st);
    const int impl = o.impl >= 0 ? o.impl : 0;
    const ConvParams p = u2_conv_params(e, o, impl == 2);
    if (impl == 3) return launch_conv_halo_f32(p, e.dtype, st);
    return impl >= 1 ? launch_conv_small(p, e.dtype, st) : launch_conv_igemm(p, e.dtype, st);
}

// First pass of a plan: every convolution that both kernels can run is timed with both on its real input (the ops before it have
// run), the faster one is kept for this plan. An op rewrites the same output from the same inputs, so repeating it is harmless.
static int u2_tune_op(yp_u2net& e, U2Op& o, hipStream_t st) {
    const ConvParams p = u2_conv_params(e, o);
    const bool can_small = conv_small_valid(p, e.dtype), can_halo = conv_halo_f32_valid(p, e.dtype);
    if (!can_small && !can_halo) { o.impl = 0; return YP_OK; }
    static const bool fuse_pool = [] { const char* s = getenv("YOLOP_U2_FUSE_POOL"); return !(s && s[0] == '0'); }();
    const int pre_op = o.pool_op >= 0 ? o.pool_op : o.up_op;      // the pool / up-sample this conv could take while loading (at most one)
    const bool can_fuse = can_small && fuse_pool && pre_op >= 0 && conv_small_valid(u2_conv_params(e, o, true), e.dtype);
    if (e.small_max >= 0) {                              // forced (tests): conv_small up to small_max pixels, the halo kernel above it
        o.impl = (can_small && u2_default_impl(e, p)) ? (can_fuse ? 2 : 1) : ((can_halo && e.small_max > 0) ? 3 : 0);
        return YP_OK;
    }
    hipEvent_t e0, e1;
    U2HIP(hipEventCreate(&e0));
    U2HIP(hipEventCreate(&e1));
    // variants: 0 igemm, 1 small, 2 small + pool, 3 halo_f32, 4 = the pool launch alone (added to the unfused variants when comparing)
    float best[5] = {1e30f, 1e30f, 1e30f, 1e30f, 1e30f};
    for (int v = 0; v < 5; ++v) {
        if ((v == 1 && !can_small) || ((v == 2 || v == 4) && !can_fuse) || (v == 3 && !can_halo)) continue;
        const ConvParams pv = u2_conv_params(e, o, v == 2);
        for (int rep = 0; rep < 4; ++rep) {              // rep 0 warms (code object, caches)
            U2HIP(hipEventRecord(e0, st));
            hipError_t err;
            if (v == 4) err = e.dtype == DT_BF16 ? run_small<__bf16>(e, e.ops[pre_op], nullptr, st) : run_small<float>(e, e.ops[pre_op], nullptr, st);
            else if (v == 3) err = launch_conv_halo_f32(pv, e.dtype, st);
            else err = v >= 1 ? launch_conv_small(pv, e.dtype, st) : launch_conv_igemm(pv, e.dtype, st);
            if (err != hipSuccess) return u2fail(YP_ERR_HIP, "tuning launch of op '%s' failed: %s", o.name.c_str(), hipGetErrorString(err));
            U2HIP(hipEventRecord(e1, st));
            U2HIP(hipEventSynchronize(e1));
            float ms = 0.f;
            U2HIP(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best[v]) best[v] = ms;
        }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    o.impl = 0;
    if (best[1] < best[o.impl]) o.impl = 1;
    if (best[3] < best[o.impl]) o.impl = 3;
    if (can_fuse && best[2] < best[o.impl] + best[4]) o.impl = 2;
    static const char* names[4] = {"igemm", "small", "small+pre", "halo_f32"};
    if (getenv("YOLOP_U2_TUNE_LOG")) fprintf(stderr, "[u2 tune] %-28s M %7d Cin %3d Cout %2d dil %d: igemm %.1f us, small %.1f us, small+pool/up %.1f us (pool/up alone %.1f us), halo_f32 %.1f us -> %s\n",
                                             o.name.c_str(), p.M, p.Cin, p.Cout, p.dil, best[0] * 1e3f, can_small ? best[1] * 1e3f : 0.f, can_fuse ? best[2] * 1e3f : 0.f,
                                             can_fuse ? best[4] * 1e3f : 0.f, can_halo ? best[3] * 1e3f : 0.f, names[o.impl]);
    return YP_OK;
}

static void u2_put(std::vector<unsigned char>& buf, size_t idx, float v, int dtype) {
    if (dtype == DT_BF16) { const uint16_t h = f2bf(v); memcpy(&buf[idx * 2], &h, 2); }
    else memcpy(&buf[idx * 4], &v, 4);
}
// dense [Cout][Cin][k][k] -> [Cout^128][Kpad], k order (ky,kx,ci) over the channel-padded input (pure host arithmetic)
static void u2_pack(const yp_u2net& e, const U2Weight& w, std::vector<unsigned char>& buf) {
    buf.assign(w.mat_bytes, 0);
    for (int co = 0; co < w.cout; ++co)
        for (int ci = 0; ci < w.cin; ++ci)
            for (int ky = 0; ky < w.k; ++ky)
                for (int kx = 0; kx < w.k; ++kx)
                    u2_put(buf, (size_t)co * w.Kpad + (size_t)(ky * w.k + kx) * w.cin_pad + ci, w.w[(((size_t)co * w.cin + ci) * w.k + ky) * w.k + kx], e.dtype);
}

}  // namespace

extern "C" {

int yp_u2net_create(int variant, int dtype, int device, yp_u2net** out) {
    if (!out) return u2fail(YP_ERR_ARG, "null argument");
    if (dtype != YP_BF16 && dtype != YP_F32) return u2fail(YP_ERR_ARG, "bad dtype");
    std::unique_ptr<yp_u2net> e(new yp_u2net());
    e->variant = variant; e->dtype = dtype; e->device = device;
    if (const char* sm = getenv("YOLOP_U2_SMALL_MAX")) e->small_max = atol(sm);
    const int rc = build_u2net(*e);
    if (rc != YP_OK) return rc;
    *out = e.release();
    return YP_OK;
}

int yp_u2net_destroy(yp_u2net* e) {
    if (!e) return YP_OK;
    if (e->finalized || e->arena) { (void)hipSetDevice(e->device); (void)hipDeviceSynchronize(); }
    for (auto& w : e->weights) { if (w.d_w) (void)hipFree(w.d_w); if (w.d_b) (void)hipFree(w.d_b); }
    if (e->arena) (void)hipFree(e->arena);
    if (e->d_fuse) (void)hipFree(e->d_fuse);
    if (e->gexec) (void)hipGraphExecDestroy(e->gexec);
    if (e->in_buf) { (void)hipFree(e->in_buf); (void)hipFree(e->o_prob); (void)hipFree(e->o_norm); (void)hipFree(e->o_mask); }
    if (e->ev_in) (void)hipEventDestroy(e->ev_in);
    if (e->ev_out) (void)hipEventDestroy(e->ev_out);
    if (e->own_stream) (void)hipStreamDestroy(e->own_stream);
    delete e;
    return YP_OK;
}

int yp_u2net_weight_count(const yp_u2net* e) { return e ? (int)e->weights.size() * 2 : u2fail(YP_ERR_ARG, "null engine"); }

int yp_u2net_weight_info(const yp_u2net* e, int i, char* name, int cap, int64_t shape[4], int* ndim) {
    if (!e || i < 0 || i >= (int)e->weights.size() * 2) return u2fail(YP_ERR_ARG, "bad weight index");
    const U2Weight& w = e->weights[i / 2];
    const bool is_bias = i & 1;
    if (name && cap > 0) snprintf(name, cap, "%s.%s", w.name.c_str(), is_bias ? "bias" : "weight");
    if (is_bias) { if (shape) { shape[0] = w.cout; shape[1] = shape[2] = shape[3] = 1; } if (ndim) *ndim = 1; }
    else { if (shape) { shape[0] = w.cout; shape[1] = w.cin; shape[2] = shape[3] = w.k; } if (ndim) *ndim = 4; }
    return YP_OK;
}

int yp_u2net_set_weight(yp_u2net* e, const char* name, const float* host, const int64_t* shape, int ndim) {
    if (!e || !name || !host || !shape) return u2fail(YP_ERR_ARG, "null argument");
    if (e->finalized) return u2fail(YP_ERR_STATE, "engine already finalized");
    std::string n(name);
    const bool is_bias = n.size() > 5 && n.compare(n.size() - 5, 5, ".bias") == 0;
    const bool is_w = n.size() > 7 && n.compare(n.size() - 7, 7, ".weight") == 0;
    if (!is_bias && !is_w) return u2fail(YP_ERR_WEIGHT, "parameter name '%s' must end in .weight or .bias", name);
    auto it = e->wmap.find(n.substr(0, n.size() - (is_bias ? 5 : 7)));
    if (it == e->wmap.end()) return u2fail(YP_ERR_WEIGHT, "unknown parameter '%s'", name);
    U2Weight& w = e->weights[it->second];
    if (is_bias) {
        if (ndim != 1 || shape[0] != w.cout) return u2fail(YP_ERR_WEIGHT, "'%s': expected shape [%d]", name, w.cout);
        w.b.assign(host, host + w.cout);
        w.have_b = true;
    } else {
        if (ndim != 4 || shape[0] != w.cout || shape[1] != w.cin || shape[2] != w.k || shape[3] != w.k)
            return u2fail(YP_ERR_WEIGHT, "'%s': expected shape [%d,%d,%d,%d]", name, w.cout, w.cin, w.k, w.k);
        w.w.assign(host, host + (size_t)w.cout * w.cin * w.k * w.k);
        w.have_w = true;
    }
    return YP_OK;
}

int yp_u2net_finalize(yp_u2net* e) {
    if (!e) return u2fail(YP_ERR_ARG, "null engine");
    if (e->finalized) return YP_OK;
    for (const auto& w : e->weights)
        if (!w.have_w || !w.have_b) return u2fail(YP_ERR_WEIGHT, "parameter '%s.%s' was never set", w.name.c_str(), w.have_w ? "bias" : "weight");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return u2fail(YP_ERR_HIP, "no HIP device: the MI355X kernels cannot run here (no CPU fallback exists)");
    if (e->device < 0 || e->device >= ndev) return u2fail(YP_ERR_ARG, "device %d out of range (%d devices)", e->device, ndev);
    U2HIP(hipSetDevice(e->device));
    std::vector<unsigned char> buf;
    for (size_t i = 0; i < e->weights.size(); ++i) {
        U2Weight& w = e->weights[i];
        if ((int)i == e->outconv_w) continue;
        u2_pack(*e, w, buf);
        U2HIP(hipMalloc(&w.d_w, buf.size()));
        U2HIP(hipMemcpy(w.d_w, buf.data(), buf.size(), hipMemcpyHostToDevice));
        U2HIP(hipMalloc((void**)&w.d_b, (size_t)w.cout * 4));
        U2HIP(hipMemcpy(w.d_b, w.b.data(), (size_t)w.cout * 4, hipMemcpyHostToDevice));
    }
    {
        const U2Weight& w = e->weights[e->outconv_w];
        float f[9] = {w.w[0], w.w[1], w.w[2], w.w[3], w.w[4], w.w[5], w.b[0], 0.f, 0.f};
        U2HIP(hipMalloc((void**)&e->d_fuse, sizeof(f)));
        U2HIP(hipMemcpy(e->d_fuse, f, sizeof(f), hipMemcpyHostToDevice));
    }
    e->finalized = true;
    return YP_OK;
}

static int u2_run(yp_u2net* e, const uint8_t* bgr, float* prob, float* norm, uint8_t* mask, hipStream_t st) {
    for (U2Op& o : e->ops) {
        if (o.kind == U2_CONV && o.impl < 0) {             // (never under a capture: a plan's first pass is eager)
            const int rc = u2_tune_op(*e, o, st);
            if (rc != YP_OK) return rc;
        }
        hipError_t err = run_u2_op(*e, o, bgr, st);
        if (err != hipSuccess) return u2fail(YP_ERR_HIP, "launch of op '%s' failed: %s", o.name.c_str(), hipGetErrorString(err));
    }
    U2Tail t{};
    for (int k = 0; k < 6; ++k) { const U2Tensor& s = e->tensors[e->side_t[k]]; t.side[k] = (const float*)s.ptr; t.h[k] = s.H; t.w[k] = s.W; }
    t.B = e->pB; t.H = e->pH; t.W = e->pW; t.fuse = e->d_fuse; t.minmax = (unsigned*)(e->d_fuse + 7); t.prob = prob;
    U2HIP(hipMemsetD32Async((hipDeviceptr_t)t.minmax, (int)0x7f800000u, 1, st));
    U2HIP(hipMemsetD32Async((hipDeviceptr_t)(t.minmax + 1), 0, 1, st));
    const size_t n = (size_t)e->pB * e->pH * e->pW;
    hipLaunchKernelGGL(u2_tail_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, t);
    if (norm || mask) hipLaunchKernelGGL(u2_norm_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, prob, t.minmax, norm, mask, n);
    U2HIP(hipGetLastError());
    return YP_OK;
}

int yp_u2net_forward(yp_u2net* e, const uint8_t* bgr_dev, int B, int H, int W, float* prob_out, float* norm_out, uint8_t* mask_out, void* stream) {
    if (!e || !bgr_dev || !prob_out) return u2fail(YP_ERR_ARG, "null argument");
    if (!e->finalized) return u2fail(YP_ERR_STATE, "yp_u2net_finalize has not been called");
    int rc = plan_u2(*e, B, H, W);
    if (rc != YP_OK) return rc;
    U2HIP(hipSetDevice(e->device));
    hipStream_t st = (hipStream_t)stream;
    if (!e->use_graph || !e->warmed) {            // eager; the first pass of every shape always is (nothing is loaded or configured under a capture)
        rc = u2_run(e, bgr_dev, prob_out, norm_out, mask_out, st);
        if (rc == YP_OK) e->warmed = true;
        return rc;
    }
    const size_t n = (size_t)B * H * W;
    if (n > e->io_cap) {
        U2HIP(hipDeviceSynchronize());
        if (e->gexec) { (void)hipGraphExecDestroy(e->gexec); e->gexec = nullptr; }
        if (e->in_buf) { (void)hipFree(e->in_buf); (void)hipFree(e->o_prob); (void)hipFree(e->o_norm); (void)hipFree(e->o_mask); }
        e->in_buf = nullptr; e->io_cap = 0;
        U2HIP(hipMalloc((void**)&e->in_buf, n * 3));
        U2HIP(hipMalloc((void**)&e->o_prob, n * 4));
        U2HIP(hipMalloc((void**)&e->o_norm, n * 4));
        U2HIP(hipMalloc((void**)&e->o_mask, n));
        e->io_cap = n;
    }
    if (!e->own_stream) {
        U2HIP(hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking));
        U2HIP(hipEventCreateWithFlags(&e->ev_in, hipEventDisableTiming));
        U2HIP(hipEventCreateWithFlags(&e->ev_out, hipEventDisableTiming));
    }
    if (!e->gexec || e->gB != B || e->gH != H || e->gW != W) {
        if (e->gexec) { U2HIP(hipStreamSynchronize(e->own_stream)); (void)hipGraphExecDestroy(e->gexec); e->gexec = nullptr; }
        hipGraph_t g = nullptr;
        U2HIP(hipStreamBeginCapture(e->own_stream, hipStreamCaptureModeThreadLocal));
        rc = u2_run(e, e->in_buf, e->o_prob, e->o_norm, e->o_mask, e->own_stream);
        const hipError_t ce = hipStreamEndCapture(e->own_stream, &g);
        if (rc != YP_OK) { if (g) (void)hipGraphDestroy(g); return rc; }
        if (ce != hipSuccess) return u2fail(YP_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(ce));
        U2HIP(hipGraphInstantiate(&e->gexec, g, nullptr, nullptr, 0));
        (void)hipGraphDestroy(g);
        e->gB = B; e->gH = H; e->gW = W;
    }
    U2HIP(hipEventRecord(e->ev_in, st));
    U2HIP(hipStreamWaitEvent(e->own_stream, e->ev_in, 0));
    U2HIP(hipMemcpyAsync(e->in_buf, bgr_dev, n * 3, hipMemcpyDeviceToDevice, e->own_stream));
    U2HIP(hipGraphLaunch(e->gexec, e->own_stream));
    U2HIP(hipMemcpyAsync(prob_out, e->o_prob, n * 4, hipMemcpyDeviceToDevice, e->own_stream));
    if (norm_out) U2HIP(hipMemcpyAsync(norm_out, e->o_norm, n * 4, hipMemcpyDeviceToDevice, e->own_stream));
    if (mask_out) U2HIP(hipMemcpyAsync(mask_out, e->o_mask, n, hipMemcpyDeviceToDevice, e->own_stream));
    U2HIP(hipEventRecord(e->ev_out, e->own_stream));
    U2HIP(hipStreamWaitEvent(st, e->ev_out, 0));
    return YP_OK;
}

int yp_u2net_set_graph(yp_u2net* e, int enable) {
    if (!e) return u2fail(YP_ERR_ARG, "null engine");
    e->use_graph = enable != 0;
    return YP_OK;
}

int yp_u2net_tensor_count(const yp_u2net* e) { return e ? (int)e->tensors.size() : u2fail(YP_ERR_ARG, "null engine"); }

int yp_u2net_tensor_info(const yp_u2net* e, int i, char* name, int cap, int dims[4]) {
    if (!e || i < 0 || i >= (int)e->tensors.size()) return u2fail(YP_ERR_ARG, "bad tensor index");
    const U2Tensor& t = e->tensors[i];
    if (name && cap > 0) snprintf(name, cap, "%s", t.name.c_str());
    if (dims) { dims[0] = e->pB; dims[1] = t.H; dims[2] = t.W; dims[3] = t.C; }
    return YP_OK;
}

int yp_u2net_tensor_read(yp_u2net* e, int i, float* host_out) {
    if (!e || i < 0 || i >= (int)e->tensors.size() || !host_out) return u2fail(YP_ERR_ARG, "bad argument");
    if (!e->arena) return u2fail(YP_ERR_STATE, "no forward has run yet");
    const U2Tensor& t = e->tensors[i];
    U2HIP(hipSetDevice(e->device));
    U2HIP(hipDeviceSynchronize());
    const size_t n = (size_t)e->pB * t.H * t.W * t.C;
    if (t.f32 || e->dtype == DT_F32) {
        U2HIP(hipMemcpy(host_out, t.ptr, n * 4, hipMemcpyDeviceToHost));
    } else {
        std::vector<uint16_t> tmp(n);
        U2HIP(hipMemcpy(tmp.data(), t.ptr, n * 2, hipMemcpyDeviceToHost));
        for (size_t j = 0; j < n; ++j) host_out[j] = bf2f(tmp[j]);
    }
    return YP_OK;
}

}  // extern "C"
