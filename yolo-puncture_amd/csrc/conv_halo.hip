// bf16 3x3 stride-1 convolution with the input tile (+1-pixel halo) staged ONCE per 32-channel chunk in LDS and
// reused by all 9 taps - the im2col form in conv_dma.hip pulls every input pixel through the texture path 9 times
// (measured: 8-11 TB/s of L2->LDS gather traffic, the limiter of the dense 3x3 layers); here it is pulled once.
//
// Workgroup tile: TH x 16 output pixels x BN output channels; TH = WGM*FM rows (a fragment = one output row of 16
// pixels), each wave owns FM rows x FN*16 channels. Work is cut into stages (c, kx): 32-channel chunk c of the input,
// kernel column kx. A stage needs the chunk's halo tile [(TH+2)*18 pixels][32 ch] and the weight slab
// [3 ky][BN][32]; per stage a wave keeps the 3*FN weight fragments in registers and streams the FM+2 halo rows
// through them, so one LDS read of a halo-row fragment feeds up to 3*FN MFMAs (rows r = hy-ky).
// All operands arrive by LDS-DMA (`buffer_load ... lds`, out-of-range offsets = zero padding); loads run one whole
// chunk (3 stages) ahead: 6 weight slots + 2 halo slots, counted `s_waitcnt vmcnt` per kx block (compile-time
// constants derived from the fixed issue order below), one raw s_barrier per stage.
//
// Issue order per wave (H = halo chunk, W = weight slab):  H0 W00 W01 W02 | c=0: [W10 H1] [W11] [W12] | c=1: ...
//   wait before block kx=0 of chunk c : W(c,0),H(c) landed   -> younger ops W(c,1) W(c,2)            = 2*LW
//   wait before block kx=1            : W(c,1) landed          -> younger W(c,2) W(c+1,0) H(c+1)      = 2*LW+LH
//   wait before block kx=2            : W(c,2) landed          -> younger W(c+1,0) H(c+1) W(c+1,1)    = 2*LW+LH
// Loads for the chunk after the last one are issued out of range so the counts stay uniform.
//
// Replaces the 3x3 `Conv`s of Bottleneck / the box branch inside `.predict` (reference yolo_seg/app.py:91; blocks per
// SURVEY.md Appendix A.2 [U]). Same numerics contract as conv_dma.hip.
#include "common.h"

namespace yp {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ float silu3(float x) { return x / (1.0f + __expf(-x)); }
template <int N> __device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt range");
    __builtin_amdgcn_s_waitcnt((N & 0xF) | (7 << 4) | (0xF << 8) | (((N >> 4) & 3) << 14));
}
__device__ __forceinline__ int hswz(int row) { return ((row >> 2) & 1) << 1; }   // 64-B rows, any 16 consecutive rows conflict-free

template <int FM, int FN, int WGM, int WGN>
__global__ __launch_bounds__(WGM * WGN * 64) void conv_halo_kernel(const ConvParams p, const int tiles_h, const int tiles_w,
                                                                  const int ntiles) {
    constexpr int NW = WGM * WGN;
    constexpr int TH = WGM * FM, BN = WGN * FN * 16;
    constexpr int HP = (TH + 2) * 18;                 // halo pixels
    constexpr int H_INSTR = (HP * 4 + 63) / 64;       // 1-KiB pieces of one halo chunk
    constexpr int LH = (H_INSTR + NW - 1) / NW;       // halo loads per wave
    constexpr int HB = H_INSTR * 1024;                // halo slot bytes (rounded up to whole pieces)
    constexpr int W_ROWS = 3 * BN;
    constexpr int W_INSTR = W_ROWS * 4 / 64;
    constexpr int LW = (W_INSTR + NW - 1) / NW;
    constexpr int WB = W_INSTR * 1024;                // weight slot bytes
    constexpr unsigned OOB = 0x80000000u;
    static_assert((W_ROWS * 4) % 64 == 0, "weight slab is whole pieces");

    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char* const Hs = smem;                   // 2 halo slots
    unsigned char* const Ws = smem + 2 * HB;          // 6 weight slots: (chunk parity)*3 + kx
    unsigned char* const dump = Ws + 6 * WB;          // 1 KiB landing zone of the padding loads

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WGM, wn = wave / WGM;
    const int fr = lane & 15, fc = lane >> 4;

    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, j = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int nt = bid % ntiles;
    int t = bid / ntiles;
    const int tw = t % tiles_w; t /= tiles_w;
    const int th = t % tiles_h;
    const int b = t / tiles_h;
    const int h0 = th * TH, w0 = tw * 16, n0 = nt * BN;

    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)p.w_bytes, 0x00020000);

    // ---- per-lane constants: halo pieces -------------------------------------------------------------------
    unsigned hconst[LH];
#pragma unroll
    for (int j = 0; j < LH; ++j) {
        const int ii = wave * LH + j;
        const int s = ii * 64 + lane;
        const int hp = s >> 2, pc = s & 3;
        const int c = pc ^ hswz(hp);
        const int hy = hp / 18, hx = hp - hy * 18;
        const int hi = h0 - 1 + hy, wi = w0 - 1 + hx;
        const bool ok = (ii < H_INSTR) && (hp < HP) && ((unsigned)hi < (unsigned)p.H) && ((unsigned)wi < (unsigned)p.W);
        hconst[j] = ok ? (unsigned)((((b * p.H + hi) * p.W + wi) * p.x_stride + p.x_coff) * 2 + c * 16) : OOB;
    }
    // weight slab pieces: LDS row rw = ky*BN + n  <-  packed W[n0+n][(ky*3+kx)*Cin + c*32 + cc*8]
    unsigned wconst[LW];
#pragma unroll
    for (int j = 0; j < LW; ++j) {
        const int ii = wave * LW + j;
        const int s = ii * 64 + lane;
        const int rw = s >> 2, pc = s & 3;
        const int c = pc ^ hswz(rw);
        const int ky = rw / BN, n = rw - ky * BN;
        wconst[j] = (ii < W_INSTR) ? (unsigned)(((n0 + n) * p.Kpad + ky * 3 * p.Cin + c * 8) * 2) : OOB;
    }
    const int nchunk = p.Cin >> 5;

    auto issue_halo = [&](int chunk) {   // chunk may be == nchunk (padding loads)
        unsigned char* dst = Hs + (chunk & 1) * HB;
        const unsigned coff = (unsigned)chunk * 64u;
#pragma unroll
        for (int j = 0; j < LH; ++j) {
            const int ii = wave * LH + j;
            const unsigned voff = (hconst[j] == OOB || chunk >= nchunk) ? OOB : hconst[j] + coff;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_void*)((ii < H_INSTR) ? dst + ii * 1024 : dump), 16, voff, 0, 0, 0);
        }
    };
    auto issue_w = [&](int chunk, int kx) {
        unsigned char* dst = Ws + ((chunk & 1) * 3 + kx) * WB;
        const unsigned koff = (unsigned)((kx * p.Cin + chunk * 32) * 2);
#pragma unroll
        for (int j = 0; j < LW; ++j) {
            const int ii = wave * LW + j;
            const unsigned voff = (wconst[j] == OOB || chunk >= nchunk) ? OOB : wconst[j] + koff;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_void*)((ii < W_INSTR) ? dst + ii * 1024 : dump), 16, voff, 0, 0, 0);
        }
    };

    f32x4 acc[FN][FM];
#pragma unroll
    for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int r = 0; r < FM; ++r) acc[a][r] = f32x4{0.f, 0.f, 0.f, 0.f};

    // one (chunk, kx) stage of MFMAs for this wave
    auto compute = [&](int chunk, int kx) {
        const unsigned char* wsl = Ws + ((chunk & 1) * 3 + kx) * WB;
        const unsigned char* hsl = Hs + (chunk & 1) * HB;
        bf16x8 wf[3][FN];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int a = 0; a < FN; ++a) {
                const int rw = ky * BN + wn * (FN * 16) + a * 16 + fr;
                wf[ky][a] = *(const bf16x8*)(wsl + swz64((unsigned)(rw * 64 + fc * 16)));
            }
#pragma unroll
        for (int hy = 0; hy < FM + 2; ++hy) {
            const int hp = (wm * FM + hy) * 18 + kx + fr;
            const bf16x8 xf = *(const bf16x8*)(hsl + swz64((unsigned)(hp * 64 + fc * 16)));
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int r = hy - ky;
                if (r >= 0 && r < FM) {
#pragma unroll
                    for (int a = 0; a < FN; ++a)
                        acc[a][r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ky][a], xf, acc[a][r], 0, 0, 0);
                }
            }
        }
    };

    // ---- pipeline ------------------------------------------------------------------------------------------------
    issue_halo(0);
    issue_w(0, 0);
    issue_w(0, 1);
    issue_w(0, 2);
    for (int c = 0; c < nchunk; ++c) {
        wait_vm<2 * LW>();
        __builtin_amdgcn_s_barrier();
        issue_w(c + 1, 0);
        issue_halo(c + 1);
        compute(c, 0);

        wait_vm<2 * LW + LH>();
        __builtin_amdgcn_s_barrier();
        issue_w(c + 1, 1);
        compute(c, 1);

        wait_vm<2 * LW + LH>();
        __builtin_amdgcn_s_barrier();
        issue_w(c + 1, 2);
        compute(c, 2);
    }
    wait_vm<0>();

    // ---- epilogue ------------------------------------------------------------------------------------------------
    const bool vec_ok = ((p.Cout & 3) == 0) && ((p.y_stride & 3) == 0) && ((p.y_coff & 3) == 0) &&
                        (p.res == nullptr || (((p.res_stride & 3) == 0) && ((p.res_coff & 3) == 0)));
    float bias[FN][4];
#pragma unroll
    for (int a = 0; a < FN; ++a) {
        const int co = n0 + wn * (FN * 16) + a * 16 + fc * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) bias[a][r] = (co + r < p.Cout) ? p.bias[co + r] : 0.f;
    }
    const int wo = w0 + fr;
#pragma unroll
    for (int r = 0; r < FM; ++r) {
        const int ho = h0 + wm * FM + r;
        if (ho >= p.Ho || wo >= p.Wo) continue;
        const size_t m = ((size_t)b * p.Ho + ho) * p.Wo + wo;
#pragma unroll
        for (int a = 0; a < FN; ++a) {
            const int co = n0 + wn * (FN * 16) + a * 16 + fc * 4;
            if (co >= p.Cout) continue;
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float tt = acc[a][r][i] + bias[a][i];
                if (p.act == ACT_SILU) tt = silu3(tt);
                v[i] = tt;
            }
            if (p.res) {
                const __bf16* rp = (const __bf16*)p.res + m * p.res_stride + p.res_coff + co;
                if (vec_ok) {
                    const uint2 rr = *(const uint2*)rp;
                    v[0] += __uint_as_float(rr.x << 16); v[1] += __uint_as_float(rr.x & 0xffff0000u);
                    v[2] += __uint_as_float(rr.y << 16); v[3] += __uint_as_float(rr.y & 0xffff0000u);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (co + i < p.Cout) v[i] += (float)rp[i];
                }
            }
            if (p.out_f32) {
                float* yp = (float*)p.y + m * p.y_stride + p.y_coff + co;
                if (vec_ok) *(float4*)yp = make_float4(v[0], v[1], v[2], v[3]);
                else
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (co + i < p.Cout) yp[i] = v[i];
            } else {
                __bf16* yp = (__bf16*)p.y + m * p.y_stride + p.y_coff + co;
                if (vec_ok) {
                    __attribute__((aligned(8))) __bf16 o[4] = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                    *(uint2*)yp = *(const uint2*)o;
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (co + i < p.Cout) yp[i] = (__bf16)v[i];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
struct HaloCfg { int FM, FN, WGM, WGN; const char* name; };
static const HaloCfg kHalo[] = {
    {8, 2, 2, 2, "conv_halo_kernel<8,2,2,2>"},   // 0: 16x16 px x 64 ch, 4 waves
    {4, 2, 4, 1, "conv_halo_kernel<4,2,4,1>"},   // 1: 16x16 px x 32 ch, 4 waves
    {4, 2, 2, 2, "conv_halo_kernel<4,2,2,2>"},   // 2:  8x16 px x 64 ch, 4 waves
    {4, 2, 4, 2, "conv_halo_kernel<4,2,4,2>"},   // 3: 16x16 px x 64 ch, 8 waves
    {4, 2, 2, 1, "conv_halo_kernel<4,2,2,1>"},   // 4:  8x16 px x 32 ch, 2 waves -> not instantiated (NW must be 4/8)
};
constexpr int kNumHalo = 4;

int conv_halo_num_cfgs() { return kNumHalo; }

bool conv_halo_cfg_valid(const ConvParams& p, int c) {
    if (c < 0 || c >= kNumHalo) return false;
    if (p.ks != 3 || p.stride != 1 || p.pad != 1 || p.up != 1 || (p.Cin % 32) != 0 || (p.Kpad != 9 * p.Cin)) return false;
    if (p.x_bytes >= (1ull << 31) || p.w_bytes >= (1ull << 31)) return false;
    const HaloCfg& k = kHalo[c];
    const int BN = k.WGN * k.FN * 16, TH = k.WGM * k.FM;
    const int cpad = (p.Cout + 31) / 32 * 32;
    if (BN > cpad) return false;
    // reject shapes where the fixed 16-wide tiles waste more than a third of the MFMAs
    const long covered = (long)((p.Ho + TH - 1) / TH * TH) * ((p.Wo + 15) / 16 * 16);
    if (covered * 2 > (long)p.Ho * p.Wo * 3) return false;
    return true;
}

const char* conv_halo_kernel_name(int c) { return kHalo[c].name; }

template <int FM, int FN, int WGM, int WGN>
static hipError_t launch_halo_one(const ConvParams& p, hipStream_t st) {
    constexpr int NW = WGM * WGN, TH = WGM * FM, BN = WGN * FN * 16;
    constexpr int HP = (TH + 2) * 18, H_INSTR = (HP * 4 + 63) / 64, W_INSTR = 3 * BN * 4 / 64;
    (void)NW;
    const size_t sh = (size_t)2 * H_INSTR * 1024 + (size_t)6 * W_INSTR * 1024 + 1024;
    const int B = p.M / (p.Ho * p.Wo);
    const int tiles_h = (p.Ho + TH - 1) / TH, tiles_w = (p.Wo + 15) / 16, ntiles = (p.Cout + BN - 1) / BN;
    auto kern = conv_halo_kernel<FM, FN, WGM, WGN>;
    static bool attr = false;
    if (!attr && sh > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        if (e != hipSuccess) return e;
        attr = true;
    }
    hipLaunchKernelGGL(kern, dim3(B * tiles_h * tiles_w * ntiles), dim3(WGM * WGN * 64), sh, st, p, tiles_h, tiles_w, ntiles);
    return hipGetLastError();
}

hipError_t launch_conv_halo(const ConvParams& p, int c, hipStream_t st) {
    switch (c) {
        case 0: return launch_halo_one<8, 2, 2, 2>(p, st);
        case 1: return launch_halo_one<4, 2, 4, 1>(p, st);
        case 2: return launch_halo_one<4, 2, 2, 2>(p, st);
        default: return launch_halo_one<4, 2, 4, 2>(p, st);
    }
}

}  // namespace yp
