// Internal types shared by the graph builder, the executor and the HIP kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

namespace yp {

enum DType { DT_BF16 = 0, DT_F32 = 1 };
enum Act { ACT_NONE = 0, ACT_SILU = 1, ACT_RELU = 2 };   // (ReLU: conv_igemm only - the U^2-Net path)

enum OpKind {
    OP_STEM = 0,      // u8 BGR NHWC -> first 3x3 s2 conv (+/255, BGR->RGB fused)
    OP_CONV = 1,      // dense conv k in {1,3}, s in {1,2}: implicit GEMM on MFMA, fused bias/SiLU/residual
    OP_DWCONV = 2,    // depthwise k in {3,7}
    OP_POOL5 = 3,     // max-pool 5x5 s1 p2 (SPPF)
    OP_UPSAMPLE = 4,  // nearest x2
    OP_ATTN = 5,      // PSA attention core: softmax(q^T k * scale) applied to v
    OP_HEAD = 6,      // DFL decode + sigmoid + two-stage top-k
    OP_CONVT = 7,     // ConvTranspose2d k2 s2 (Proto) = 4 strided 1x1 GEMMs
    OP_AMAX = 9,      // per-level class-max of the head (sigmoid(max_c logit)) - runs on the level's class lane
    OP_POOL3 = 8,     // SPPF: three chained 5x5 max-pools in one launch (out = first pooled slice, 3*C channels written)
};

// Channel slice of an NHWC tensor: element (b,y,x,c) at ((b*H+y)*W+x)*pix_stride + coff + c
struct View {
    int t = -1;    // tensor id
    int coff = 0;  // first channel
    int C = 0;     // channels in the slice
};

struct TensorDesc {
    std::string name;
    int C = 0;
    int sdiv = 1;        // spatial size = input size / sdiv
    bool f32 = false;    // fp32 regardless of engine dtype (head logits)
    // resolved by the plan:
    int H = 0, W = 0;
    size_t bytes = 0;
    void* ptr = nullptr;
};

struct WeightDesc {
    std::string name;    // folded name, e.g. "model.2.cv1" ; parameters "<name>.weight" / "<name>.bias"
    int cout = 0, cin_g = 0, k = 1, groups = 1;
    int cin_pad = 0;           // dense convs: channels per tap in the PACKED matrix (= cin_g rounded up to 32 when that is not a multiple of 32
                               // and > 32; zero weights in the gap) - the conv kernels then run with Cin = cin_pad, see engine.hip conv_params
    bool transposed = false;   // ConvTranspose2d layout [Cin][Cout][k][k]
    bool is_stem = false;
    bool have_w = false, have_b = false;
    std::vector<float> w, b;   // host fp32 copies until finalize
    // device, packed
    void* d_w = nullptr;       // layout depends on the consumer kernel
    void* d_w2 = nullptr;      // stem only: bf16 [C0][32] GEMM layout for the MFMA stem
    float* d_b = nullptr;
    int Kpad = 0;
    size_t mat_bytes = 0;      // bytes of one packed GEMM matrix
};

struct Op {
    int kind = OP_CONV;
    std::string name;
    View in, out, res;
    int widx = -1;
    int k = 1, s = 1, act = ACT_NONE;
    int gs = 0, gstride = 0;      // depthwise input channel gather: in_ch = coff + (c/gs)*gstride + c%gs (gs=0: identity)
    int nh = 0, kd = 0, hd = 0;   // attention
    View box[3], cls[3], cf[3];   // head inputs per level
    View amax[3];                 // head: per-level anchor-max keys (written by the OP_AMAX ops)
    int nlev = 0;
    double flops = 0, bytes = 0;  // algorithmic (filled by the plan)
    std::string kernel;           // device kernel symbol this op launches (filled by the plan)
    int cfg = -1;                 // autotuned conv_dma configuration (-1: heuristic)
    int fuse_dw = -1;             // OP_CONV 1x1: index of the depthwise 3x3 op feeding it that can be fused in (graph pass)
    bool fused = false;           // plan decision: this conv runs as the fused dw->pw kernel
    bool skip = false;            // plan decision: this op's work is done by a fused consumer
    int fuse_pre = -1;            // OP_CONV 1x1: index of the 3x3 stride-2 conv feeding it that can run as the first stage of one kernel
    bool fused2 = false;          // plan decision: this 1x1 runs as the second stage of conv_halo_s2's PW2 form
    int stem_op = -1;             // fused2 whose first stage reads the stem's output: index of the stem op
    bool fused3 = false;          // plan decision: stem -> 3x3 s2 -> this 1x1 run as frontend_kernel
    int fold_up = -1;             // OP_CONV 1x1 on a [upsampled | skip] concat: index of the nearest-x2 upsample op it can absorb
    bool folded = false;          // plan decision: the upsample is folded into this conv's input gather
    int c2f_m1 = -1, c2f_m2 = -1; // OP_CONV 1x1 closing a C2f with one plain bottleneck: indices of the bottleneck's two 3x3 convs
    bool fused4 = false;          // plan decision: both 3x3 convs and this 1x1 run as c2f_fused_kernel
    int scd_pre = -1;             // OP_DWCONV 3x3 s2 closing an SCDown: index of the 1x1 conv in front of it
    bool fused5 = false;          // plan decision: that 1x1 and this depthwise conv run as scdown_fused_kernel
    int fuse_tail = -1, tail_amax = -1;   // OP_CONV 1x1 without activation (fp32 logits): index of the dw->pw pointwise conv feeding it that can take it
                                  // on as a third stage, and of the OP_AMAX op behind it (or -1)
    bool fused6 = false;          // plan decision: dw -> pw -> this 1x1 (+ the class-max keys) run as conv_dwpw_kernel's TAIL form
    int amax_post = -1;           // OP_CONV 1x1 that writes a level's fp32 class logits: index of the OP_AMAX op that reads them (graph pass)
    bool fused8 = false;          // plan decision: logits and class-max keys come out of one cls_out_kernel launch; the OP_AMAX op is skipped
    int pw_pre = -1;              // OP_DWCONV (3x3 / 7x7, stride 1) / OP_POOL3: index of the 1x1 conv that produces its input and can run as the first stage of pwsp_kernel (graph pass)
    bool fused7 = false;          // plan decision: that 1x1 and this spatial op run as pwsp_kernel (one workgroup per image and channel slice); the 1x1 is skipped
    bool pw_store = false;        // fused7: the 1x1's own output has other readers and is written as well
    int lane = 0;                 // capture lane: independent head branches run on their own streams inside the hipGraph
    bool nms = false;             // OP_HEAD: conf filter + class-aware NMS (YOLOv8 / YOLO11) instead of the two-stage top-k (v10)
    int hb_box[3][3] = {{-1, -1, -1}, {-1, -1, -1}, {-1, -1, -1}}, hb_cf[3][3] = {{-1, -1, -1}, {-1, -1, -1}, {-1, -1, -1}};   // OP_HEAD (v10): op indices of the box / coefficient branch convs per level ({0,1,2} = 3x3, 3x3, 1x1), -1 = none
    bool sparse_box = false, sparse_cf = false;   // plan decision: that branch runs on the stage-1 winners only (head_branch.hip); its dense ops are skipped
};

// ---------------------------------------------------------------------------------------------------------
// Kernel parameter blocks (passed by value)
// ---------------------------------------------------------------------------------------------------------
struct ConvParams {
    const void* x; int x_stride, x_coff;        // input view (elements)
    int H, W, Cin;
    const void* w; int Kpad;                    // packed [CoutPad][Kpad], k order (ky,kx,ci)
    const float* bias;
    void* y; int y_stride, y_coff; int Ho, Wo, Cout;
    const void* res; int res_stride, res_coff;  // nullable; added after the activation
    int M;                                      // B*Ho*Wo
    int ks, stride, pad;
    int act, out_f32;
    int up, oy, ox;                             // output pixel (ho,wo) -> (ho*up+oy, wo*up+ox) in an (Ho*up,Wo*up) image
    size_t x_bytes, w_bytes, y_bytes;           // extents of the x / packed-weight / y buffers (buffer descriptors)
    int cfg;                                    // conv_dma tile configuration (-1: heuristic)
    int dbg;                                    // timing ablations (tests/tools only): 1 = drop stores, 2 = drop pixel loads
    // folded nearest-x2 upsample (1x1 convs on a [upsampled | skip] concat): input channels [0, x2_C) are read from the
    // low-resolution tensor x2 at pixel (ho >> 1, wo >> 1) instead of from x; channels >= x2_C come from x as usual
    const void* x2; size_t x2_bytes; int x2_stride, x2_coff, x2_C, x2_H, x2_W;
    // fused trailing 1x1 (conv_halo_s2 PW2 form): y = act2(W2 . act(conv(x)) + bias2); Cout is then the intermediate width,
    // y / y_stride / y_coff / y_bytes describe the FINAL output of C2 channels
    const void* w2; const float* bias2; int C2, act2, Kpad2; size_t w2_bytes;
    unsigned long long* clk;                    // debug (YOLOP_LC_CLOCKS=1, conv_dma_lc only): per-wave phase clocks, else null
    int dil;                                    // dilation of a 3x3 (0 = 1); conv_igemm only - the U^2-Net path (pad = dil there)
    // conv_small only: x is the tensor IN FRONT OF a 2x2 / stride-2 / ceil-mode max pool (src_H x src_W pixels) and the pool is taken
    // while loading; H, W stay the pooled size the convolution sees
    int pool_in, src_H, src_W;
    // conv_small only: the x2 channels are a BILINEAR resize (align_corners = False) of x2 to H x W instead of the nearest x2 above
    int up_bilinear;
};

struct DwParams {
    const void* x; int x_stride, x_coff; int H, W, C;
    const void* w;             // packed [k*k][C]
    const float* bias;
    void* y; int y_stride, y_coff; int Ho, Wo;
    const void* res; int res_stride, res_coff;
    int B, ks, stride, pad, act;
    int gs, gstride;
    size_t x_bytes;            // extent of the x tensor (buffer descriptor of the row kernels)
};

struct StemParams {
    const uint8_t* x; int H, W;       // [B,H,W,3] BGR
    const float* w;                   // [3][3][3(bgr)][C0]
    const void* wpk;                  // bf16 [C0][32], k = (ky,kx,c_bgr), zero padded (MFMA stem); may be null
    const float* bias;
    void* y; int y_stride, y_coff; int Ho, Wo, C0; int B;
    int act;
};

// fused front end (frontend.hip): stem -> 3x3 s2 conv -> 1x1 conv
struct FrontParams {
    const uint8_t* img; int imgH, imgW, B;                               // [B,imgH,imgW,3] BGR u8
    const void* w0; const float* bias0; int act0, C0; int H1, W1;         // stem: bf16 [C0][32], output grid H1 x W1
    const void* w1; const float* bias1; int act1, C1, Kpad1; size_t w1_bytes;   // 3x3 s2: packed [C1^][9*C0]
    const void* w2; const float* bias2; int act2, C2, Kpad2; size_t w2_bytes;   // 1x1:    packed [C2^][C1]
    void* y; int y_stride, y_coff; size_t y_bytes; int Ho, Wo;           // final output view (Ho x Wo = H1/2 x W1/2)
    unsigned long long* clk;                                              // debug (YOLOP_FRONT_CLOCKS=1): per-wave stage clocks, else null
};
bool frontend_valid(const FrontParams& p);
hipError_t launch_frontend(const FrontParams& p, hipStream_t st);

// fused C2f tail (c2f_fused.hip): [a | b] -> 3x3 -> 3x3 (+ b) -> 1x1 over [a | b | c]
struct C2fParams {
    const void* x; int x_stride, x_coff; size_t x_bytes;                 // the concat buffer: a = channels [coff, coff+C), b = the next C
    int B, H, W, C;
    const void* w1; const float* bias1; int act1, Kpad1; size_t w1_bytes;   // m.0.cv1: packed [C^][9*C]
    const void* w2; const float* bias2; int act2, Kpad2; size_t w2_bytes;   // m.0.cv2: packed [C^][9*C]
    int shortcut;                                                         // c += b
    const void* w3; const float* bias3; int act3, Kpad3, Cout; size_t w3_bytes;   // cv2: packed [Cout^][3*C]
    void* y; int y_stride, y_coff; size_t y_bytes;
    unsigned long long* clk;                                              // debug (YOLOP_C2F_CLOCKS=1): per-wave stage clocks, else null
};
bool c2f_fused_valid(const C2fParams& p);

// fused SCDown (scdown_fused.hip): 1x1 conv + act -> depthwise 3x3 stride 2
struct ScdParams {
    const void* x; int x_stride, x_coff; size_t x_bytes; int B, H, W, K;          // input view [B,H,W,K]
    const void* w1; const float* bias1; int act1, Kpad1, C; size_t w1_bytes;        // 1x1: packed [C^][K]
    const void* wd; const float* biasd; int actd;                                    // depthwise 3x3 s2: packed [9][C] bf16, bias fp32
    void* y; int y_stride, y_coff; size_t y_bytes; int Ho, Wo;
    unsigned long long* clk;                                                          // debug (YOLOP_SCD_CLOCKS=1), else null
};
bool scdown_fused_valid(const ScdParams& p);
const char* scdown_fused_kernel_name(const ScdParams& p);
bool scdown_stream_valid(const ScdParams& p);                     // the streaming form for 256 output channels (scdown_stream.hip)
const char* scdown_stream_kernel_name(const ScdParams& p);
hipError_t launch_scdown_stream(const ScdParams& p, hipStream_t st);
hipError_t launch_scdown_fused(const ScdParams& p, hipStream_t st);
hipError_t launch_c2f_fused(const C2fParams& p, hipStream_t st);

struct PoolParams {
    const void* x; int x_stride, x_coff;
    void* y; int y_stride, y_coff;
    int B, H, W, C;
};

struct UpParams {
    const void* x; int x_stride, x_coff;
    void* y; int y_stride, y_coff;
    int B, H, W, C;   // input dims; output 2H x 2W
};

struct AttnParams {
    const void* qkv; int q_stride, q_coff;   // [B,N,nh*(2kd+hd)]
    void* o; int o_stride, o_coff;           // [B,N,nh*hd]
    int B, N, nh, kd, hd;
    float scale;
};

struct HeadParams {
    const float* box[3]; const float* cls[3]; const float* cf[3];   // fp32 NHWC logits per level
    int hw[3][2]; int nlev;
    int B, nc, max_det, A;
    float* det; int32_t* idx; float* coeff;   // user outputs
    void* scratch;                            // device scratch, head_scratch_bytes(B, A)
    const unsigned* mk[3];                    // per-level anchor-max keys [B][HW_l] (bits of sigmoid(max_c logit)); when set, the
                                              // class-max pass already ran (OP_AMAX) and `scratch` is not used
    // NMS heads (YOLOv8 / YOLO11, head_nms.hip): device parameters [conf, iou] and per-image candidate scratch [B][A][8] floats
    const float* nms_params;
    float* nms_ws;
    // winners-only head (round 3): stage 1 leaves its winners in sp_sel / sp_wlist / sp_wcount / sp_thr, the branch kernel(s) fill
    // sp_box [B][max_det][64] (and sp_cf [B][max_det][32]), stage 2 + decode read those rows by rank instead of box[] / cf[]
    int* sp_sel; int* sp_wlist; int* sp_wcount; unsigned* sp_thr; float* sp_box; float* sp_cf;
    int* sp_plist; int* sp_pcount; int sp_plist_off[3], sp_plist_cap[3];   // needed positions per level (see HeadBranchParams); null = not wanted
};
hipError_t launch_head_stage1(const HeadParams& p, hipStream_t st);
hipError_t launch_head_stage2(const HeadParams& p, hipStream_t st);
constexpr int HEAD_MAXK = 512;
// The box / mask-coefficient branch of the v10 one-to-one head evaluated at the stage-1 winners only (head_branch.hip)
struct HeadBranchParams {
    const void* x[3]; int x_stride[3], x_coff[3], H[3], W[3], Cin[3]; size_t x_bytes[3];     // the levels' feature maps (bf16 NHWC views)
    const void* w0[3]; int Kpad0[3]; const float* b0[3];                                       // 3x3 Cin -> cmid, packed [cmid^][9*Cin]
    const void* w1[3]; int Kpad1[3]; const float* b1[3];                                       // 3x3 cmid -> cmid
    const void* w2[3]; int Kpad2[3]; const float* b2[3];                                       // 1x1 cmid -> cout, no activation
    int act0, act1;
    int B, max_det, maxk, cmid, cout;
    int A0, A1;                                   // anchors of level 0 / 1 (anchor id -> level-local pixel)
    const int* sel;                               // [B][maxk] stage-1 winners (anchor ids, rank order)
    const int* wlist;                             // [B][3][maxk] ranks of the winners that lie on a level
    const int* wcount;                            // [B][3]
    float* out;                                   // [B][max_det][cout] fp32 rows by rank
    // positions form (head_pos_kernel + head_win_kernel): the first 3x3 once per NEEDED position (the union of the winners' in-frame 3x3
    // neighbourhoods, listed by the stage-1 kernel), its output in a position-addressed map that the winners' second 3x3 gathers from
    const int* plist; const int* pcount;          // per level: entries image << 20 | level-local pixel; pcount[3]
    int plist_off[3], plist_cap[3];               // level l's list = plist + plist_off[l], at most plist_cap[l] entries
    void* t0; size_t t0_off[3], t0_bytes;         // bf16 [level][B][H_l*W_l][cmid]: element offset of level l
    int pos_grid;                                 // workgroups of the (persistent) position kernel
};
bool head_branch_valid(const HeadBranchParams& p);
hipError_t launch_head_branch(const HeadBranchParams& p, hipStream_t st);
hipError_t launch_head_nms(const HeadParams& p, hipStream_t st);
size_t head_nms_scratch_bytes(int B, int A);

#if defined(__HIPCC__)
// LDS byte offset of the 16-byte piece `c` of the 64-byte row `row` under the chunk swizzle c ^ (((row >> 2) & 1) << 1), computed from the
// UNSWIZZLED offset L = row * 64 + c * 16: the swizzle flips bit 5 of L where bit 2 of row = bit 8 of L is set. Written on L, a fragment read
// costs one add (row offset of the tap, usually a constant) + two bit operations; written on `row`, the compiler spent ~9 VALU instructions per
// read (PMC on conv_tile1: VALU issue 48 % of the kernel's cycles, matrix pipe busy 30 %).
__device__ __forceinline__ unsigned swz64(unsigned L) { return L ^ ((L >> 3) & 32u); }
#endif

// Four SiLUs with the two multiplies and the add as packed fp32 operations (v_pk_mul_f32 / v_pk_add_f32: two values per
// instruction). Same operations and roundings as  x * rcp(1 + exp2(-x * log2e))  element by element, so the same bits; the
// conv epilogues are VALU-bound on exactly this sequence (28 -> 22 cycles per element).
#if defined(__HIPCC__)
typedef float yp_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void silu4_packed(float* v) {
#pragma unroll
    for (int i = 0; i < 4; i += 2) {
        yp_f32x2 x = {v[i], v[i + 1]};
        yp_f32x2 t = x * -1.4426950408889634f;
        t[0] = __builtin_amdgcn_exp2f(t[0]); t[1] = __builtin_amdgcn_exp2f(t[1]);
        t = t + 1.0f;
        t[0] = __builtin_amdgcn_rcpf(t[0]); t[1] = __builtin_amdgcn_rcpf(t[1]);
        x = x * t;
        v[i] = x[0]; v[i + 1] = x[1];
    }
}
#endif

// launches (implemented in the .hip files); dtype selects the template instance
hipError_t launch_conv(const ConvParams& p, int dtype, hipStream_t st);
hipError_t launch_conv_igemm(const ConvParams& p, int dtype, hipStream_t st);   // always the register-staged kernel (dilation, ReLU, any Cin % 8 == 0)
const char* conv_kernel_name(const ConvParams& p, int dtype);
bool conv_cfg_usable(const ConvParams& p, int dtype, int cfg);   // a configuration id from a cache file / yp_tuning_import is launchable for p
hipError_t launch_conv_dma(const ConvParams& p, hipStream_t st);
bool conv_dma_supported(const ConvParams& p);
const char* conv_dma_kernel_name(const ConvParams& p);
int conv_dma_num_cfgs();
void conv_dma_force_cfg(int cfg);
int conv_dma_forced_cfg();
void conv_set_debug_ablation(int v);
int conv_debug_ablation();
bool tile_balance_enabled(int family);  // tiles sized to whole rounds of workgroups? family 1 = conv_pxd, 2 = conv_wres, 4 = cls_out; YOLOP_BALANCE=<mask> (A/B switch)
// halo-tiled 3x3 s1 kernel (conv_halo.hip); configuration ids are offset by 100 in ConvParams::cfg
int conv_halo_num_cfgs();
bool conv_halo_cfg_valid(const ConvParams& p, int c);
const char* conv_halo_kernel_name(int c);
hipError_t launch_conv_halo(const ConvParams& p, int c, hipStream_t st);
// persistent im2col kernel (conv_dma_p.hip); ids offset by 300
int conv_dma_p_num_cfgs();
bool conv_dma_p_cfg_valid(const ConvParams& p, int c);
const char* conv_dma_p_kernel_name(int c);
hipError_t launch_conv_dma_p(const ConvParams& p, int c, hipStream_t st);
int conv_tile1_num_cfgs();
bool conv_tile1_cfg_valid(const ConvParams& p, int c);
const char* conv_tile1_kernel_name(int c);
hipError_t launch_conv_tile1(const ConvParams& p, int c, hipStream_t st);
// K-split kernel for small pixel counts (conv_ks.hip); ids offset by 900
int conv_ks_num_cfgs();
bool conv_ks_cfg_valid(const ConvParams& p, int c);
const char* conv_ks_kernel_name(int c);
hipError_t launch_conv_ks(const ConvParams& p, int c, hipStream_t st);
// weights-resident streaming 1x1 kernel (conv_wres.hip); ids offset by 1100
int conv_wres_num_cfgs();
bool conv_wres_cfg_valid(const ConvParams& p, int c);
const char* conv_wres_kernel_name(int c);
hipError_t launch_conv_wres(const ConvParams& p, int c, hipStream_t st);

// weights-in-registers streaming 1x1 kernel (conv_wrs.hip); ids offset by 1200
int conv_wrs_num_cfgs();
bool conv_wrs_cfg_valid(const ConvParams& p, int c);
const char* conv_wrs_kernel_name(int c);
hipError_t launch_conv_wrs(const ConvParams& p, int c, hipStream_t st);

// pixels-direct 1x1 kernel (conv_pxd.hip); ids offset by 800
int conv_pxd_num_cfgs();
bool conv_pxd_cfg_valid(const ConvParams& p, int c);
const char* conv_pxd_kernel_name(int c);
hipError_t launch_conv_pxd(const ConvParams& p, int c, hipStream_t st);
// weights-in-registers 3x3 s1 kernel (conv_wreg.hip); ids offset by 700
int conv_wreg_num_cfgs();
bool conv_wreg_cfg_valid(const ConvParams& p, int c);
const char* conv_wreg_kernel_name(int c);
hipError_t launch_conv_wreg(const ConvParams& p, int c, hipStream_t st);
int conv_halo_s2_num_cfgs();
bool conv_halo_s2_cfg_valid(const ConvParams& p, int c);
const char* conv_halo_s2_kernel_name(int c);
hipError_t launch_conv_halo_s2(const ConvParams& p, int c, hipStream_t st);
int conv_halo_s2_pw_cfg(const ConvParams& p);          // configuration for the fused trailing-1x1 form, or -1
const char* conv_halo_s2_pw_kernel_name(int c);
int conv_dma_lc_num_cfgs();
bool conv_dma_lc_cfg_valid(const ConvParams& p, int c);
const char* conv_dma_lc_kernel_name(int c);
hipError_t launch_conv_dma_lc(const ConvParams& p, int c, hipStream_t st);
// persistent weight-resident halo kernel (conv_halo_p.hip); ids offset by 200
int conv_halo_p_num_cfgs();
bool conv_halo_p_cfg_valid(const ConvParams& p, int c);
const char* conv_halo_p_kernel_name(int c);
hipError_t launch_conv_halo_p(const ConvParams& p, int c, hipStream_t st);
bool conv_dma_cfg_valid(const ConvParams& p, int cfg);
hipError_t launch_dwconv(const DwParams& p, int dtype, hipStream_t st);
hipError_t launch_stem(const StemParams& p, int dtype, hipStream_t st);
hipError_t launch_pool5(const PoolParams& p, int dtype, hipStream_t st);
hipError_t launch_sppf_pool3(const PoolParams& p, int dtype, hipStream_t st);
bool sppf_pool3_fits(const PoolParams& p, int dtype);
bool dwconv_mfma_valid(const DwParams& p, int dtype);
hipError_t launch_dwconv_mfma(const DwParams& p, hipStream_t st);
hipError_t launch_upsample(const UpParams& p, int dtype, hipStream_t st);
hipError_t launch_attention(const AttnParams& p, int dtype, hipStream_t st);
// conv_small.hip: fp32 3x3 for small maps (four waves split K, operands straight from L2)
bool conv_small_valid(const ConvParams& p, int dtype);
hipError_t launch_conv_small(const ConvParams& p, int dtype, hipStream_t st);
// conv_halo_f32.hip: fp32 3x3 for large maps (8 x 32 pixel tiles, halo patch + weights by LDS-DMA, 16-channel chunks)
bool conv_halo_f32_valid(const ConvParams& p, int dtype);
hipError_t launch_conv_halo_f32(const ConvParams& p, int dtype, hipStream_t st);
hipError_t launch_head(const HeadParams& p, hipStream_t st);
hipError_t launch_letterbox(const uint8_t* src, int h0, int w0, uint8_t* dst, int out_h, int out_w, int new_h, int new_w, int top,
                            int left, int pad, hipStream_t st);
size_t head_scratch_bytes(int B, int A);
hipError_t head_read_clocks(unsigned long long* out8);
hipError_t head_branch_read_clocks(unsigned long long* out8);
hipError_t launch_copy_out(const float* det, float* det_out, const int32_t* idx, int32_t* idx_out, const float* coeff, float* coeff_out,
                           size_t rows, hipStream_t st);
hipError_t launch_anchor_max_level(const float* cls, int B, int HW, int nc, unsigned* out, hipStream_t st);

struct DwPwParams {                              // fused depthwise 3x3 s1 -> pointwise 1x1 (conv_dwpw.hip)
    const void* x; int x_stride, x_coff; int B, H, W, C; size_t x_bytes;
    const void* w_dw; const float* b_dw; int act_dw;            // depthwise: packed [9][C] bf16, bias fp32
    const void* w_pw; int Kpad; size_t wpw_bytes; const float* b_pw; int act_pw;   // pointwise: packed [Cout^][Kpad]
    void* y; int y_stride, y_coff; size_t y_bytes; int Cout; int out_f32;
    unsigned long long* clk;                                    // debug (YOLOP_DWPW_CLOCKS=1): per-wave phase clocks, else null
    // TAIL form (round 3): a trailing 1x1 (the class branch's logit conv, no activation, fp32 output) runs as a third stage on the tile
    // while it is still in LDS: y3 = W3 . act_pw(pointwise(...)) + b3; the pointwise result itself is NOT written (y is unused). When
    // `keys` is set, the per-pixel class maximum goes out as well: keys[b*H*W + pixel] = bits of sigmoid(max_c y3) (what OP_AMAX makes)
    const void* w3; int Kpad3; size_t w3_bytes; const float* b3; int C3;
    float* y3; int y3_stride, y3_coff; size_t y3_bytes;
    unsigned* keys;
};
bool dwpw_stream_valid(const DwPwParams& p);                    // the streaming form for 128-channel inputs (conv_dwpw_stream.hip)
hipError_t launch_dwpw_stream(const DwPwParams& p, hipStream_t st);
// pointwise 1x1 -> per-channel spatial operator, one workgroup per (image, channel slice) (pwsp.hip): the small-map layers
struct PwSpParams {
    const void* x; int x_stride, x_coff; size_t x_bytes; int B, H, W, K;           // input view [B,H,W,K]
    const void* w1; const float* bias1; int act1, Kpad1, C1; size_t w1_bytes;       // 1x1: packed [C1^][K]
    void* y1; int y1_stride, y1_coff;                                              // the pointwise result's own tensor (null: not stored)
    const void* res1; int res1_stride, res1_coff;                                   // added to the pointwise result after act1 (sp = 0 only; nullable)
    int sp;                                                                         // 0 none, 1 depthwise 3x3, 2 depthwise 7x7, 3 SPPF's three chained 5x5 max-pools
    int sp_c0, Csp;                                                                 // the spatial operator reads channels [sp_c0, sp_c0 + Csp) of the pointwise result
    const void* wd; const float* biasd; int actd;                                   // depthwise: packed [k*k][Csp] bf16, bias fp32 (indexed from sp_c0)
    const void* res; int res_stride, res_coff;                                      // added after actd (nullable)
    void* y2; int y2_stride, y2_coff;                                              // spatial result (pool: stage s at channel offset s * Csp)
    int dbg;                                                                        // timing ablations (tools only, results wrong): 1 no MFMAs, 2 no pixel loads, 4 no spatial stage
};
bool pwsp_valid(const PwSpParams& p);
const char* pwsp_kernel_name(const PwSpParams& p);
constexpr int PWSP_CFG = 1000;                                                       // Op::cfg of a plain 1x1 conv that runs as pwsp_kernel<NS,0> (a tuner candidate)
hipError_t launch_pwsp(const PwSpParams& p, hipStream_t st);
hipError_t pwsp_read_clocks(unsigned long long* out32);

// class logits + class-max keys in one kernel (cls_out.hip)
struct ClsOutParams {
    const void* x; int x_stride, x_coff; size_t x_bytes;      // bf16 [M][x_stride] view of K channels
    int M, K, nc;
    const void* w; int Kpad; size_t w_bytes; const float* bias;   // packed [nc^][K]
    float* y; int y_stride, y_coff;                           // fp32 logits [M][y_stride]
    unsigned* keys;                                           // [M]: bits of sigmoid(max_c y)
};
bool cls_out_valid(const ClsOutParams& p);
const char* cls_out_kernel_name(const ClsOutParams& p);
hipError_t launch_cls_out(const ClsOutParams& p, hipStream_t st);

bool conv_dwpw_valid(const DwPwParams& p);
const char* conv_dwpw_kernel_name(const DwPwParams& p);
hipError_t launch_conv_dwpw(const DwPwParams& p, hipStream_t st);

struct MaskParams {
    const void* proto; int Hp, Wp;        // [Hp,Wp,32] engine dtype (one image)
    const float* coeff; const float* boxes; int n;
    int oh, ow;                           // output size
    int t, l, ch, cw;                     // crop rectangle of the proto image that maps onto (oh,ow) (retina) or full
    float bsx, bsy;                       // box scale applied before cropping (process_mask: mw/iw, mh/ih ; retina: 1)
    int crop_before;                      // 1: process_mask (crop at proto res then upsample), 0: native (upsample then crop)
    uint8_t* masks; int64_t* ids; int32_t* kept; int32_t* area;
    int suppress_small, min_area;
    int rh, rw;                           // > 0: second, antialiased bilinear resize of the {0,1} masks to (rh,rw) before the area test and
                                          // the id paint (auto_segment with min_side > 0); ids is then [rh,rw]
};
hipError_t launch_masks(const MaskParams& p, int dtype, hipStream_t st);
size_t masks_workspace_bytes(const MaskParams& p);
hipError_t contour_read_clocks(unsigned long long* out12);
hipError_t launch_contours(const uint8_t* masks, int n, int H, int W, int strategy, int max_pts, int32_t* pts, int32_t* count, int32_t* parts, int parts_cap,
                           double* rect, hipStream_t st);

// host-side float -> bf16 (round to nearest even), as the device's v_cvt_pk_bf16_f32
static inline uint16_t f2bf(float f) {
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline float bf2f(uint16_t h) {
    uint32_t u = ((uint32_t)h) << 16;
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}

}  // namespace yp
