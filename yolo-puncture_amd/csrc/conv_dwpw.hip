// Fused  depthwise 3x3 (s1, +bias, SiLU)  ->  pointwise 1x1 (+bias, SiLU)  in one persistent kernel (bf16).
//
// The depthwise result never goes to HBM: per 32-channel chunk the input tile (+1-pixel halo) arrives in LDS by LDS-DMA
// (same ring / exact vmcnt accounting as conv_halo_p.hip), the depthwise stage (MFMA with diagonal weight fragments,
// fp32 accumulate - the VALU form was measured VALU-bound at the speed of the two unfused kernels) writes its result
// - rounded to bf16 exactly like the materialised tensor of the unfused graph - straight into the LDS tile that the MFMA
// stage reads as its pixel operand, and the pointwise GEMM accumulates over the chunks. Both weight sets and the depthwise
// bias are LDS-resident for the lifetime of the workgroup. The loop is skewed: phase g runs depthwise(g) and MFMA(g-1)
// between the same pair of barriers, so the VALU-heavy stage of one wave overlaps the matrix stage of another.
//
// Used for the class branch of the v10Detect head (`one2one_cv3.{l}.{0,1}.{0,1}`) and the dw->pw pairs of CIB
// (SURVEY.md Appendix A.2/A.4 [U]; run inside `.predict`, reference yolo_seg/app.py:91). Saves one launch and one
// write + read of the intermediate tensor per pair.
#include "common.h"
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace yp {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ float silu_q(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
template <int N> __device__ __forceinline__ void wait_vmq() {
    static_assert(N >= 0 && N < 64, "vmcnt range");
    __builtin_amdgcn_s_waitcnt((N & 0xF) | (7 << 4) | (0xF << 8) | (((N >> 4) & 3) << 14));
}
// LDS accesses behind the compiler's back: it cannot tell them from the in-flight LDS-DMA of the next chunk apart and drains
// vmcnt to 0 in front of them (= no prefetch at all). The caller orders them with explicit lgkmcnt waits.
__device__ __forceinline__ f32x4 lds_read16_async(const unsigned char* src) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"((unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)src) : "memory");
    return v;
}
__device__ __forceinline__ void lds_write8(unsigned char* dst, uint2 v) {
    asm volatile("ds_write_b64 %0, %1" ::"v"((unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)dst), "v"(*(const unsigned long long*)&v) : "memory");
}
__device__ __forceinline__ int qswz(int row) { return ((row >> 2) & 1) << 1; }

// TH x 16 output pixels, BN output channels; NW waves as 2(m) x NW/2(n). NW = 8 puts two waves on every SIMD so that one
// wave's VALU work (fragment build, SiLU) overlaps the other's MFMAs with a single set of resident weights per CU.
template <int TH, int BN, bool OUT_F32, int NW, bool TAIL = false>
__global__ __launch_bounds__(NW * 64) void conv_dwpw_kernel(const DwPwParams p, const int tiles_h, const int tiles_w, const int ntiles,
                                                       const int G) {
    constexpr int WGM = 2, WGN = NW / 2;
    constexpr int RPW = TH / NW;                      // depthwise output rows owned by one wave
    static_assert(RPW * NW == TH && RPW >= 1, "rows per wave");
    constexpr int BM = TH * 16;
    constexpr int HP = (TH + 2) * 18;
    constexpr int H_INSTR = (HP * 4 + 63) / 64;
    constexpr int LH = (H_INSTR + NW - 1) / NW;
    constexpr int HB = H_INSTR * 1024;
    constexpr int NSH = 3;
    constexpr int WM = BM / WGM, WN = BN / WGN, FM = WM / 16, FN = WN / 16;
    constexpr int S = FM * FN;
    constexpr unsigned OOB = 0x80000000u;
    static_assert(LH + 2 * S < 64, "vmcnt immediate");

    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int nchunk = p.C >> 5;
    unsigned char* const Hs = smem;                                   // NSH halo slots
    unsigned char* const As = Hs + NSH * HB;                          // 2 pixel-operand tiles [BM][32] bf16
    unsigned char* const dump = As + 2 * BM * 64;                     // 1 KiB
    unsigned char* const Wpw = dump + 1024;                           // [nchunk][BN][32] bf16
    unsigned char* const Wdw = Wpw + (size_t)nchunk * BN * 64;        // [nchunk*9 rows][32] bf16, rounded up to whole KiB
    const int wdw_instr = (nchunk * 9 + 15) / 16;
    float* const Bdw = (float*)(Wdw + (size_t)wdw_instr * 1024);      // [C] fp32, rounded up to whole KiB
    // TAIL: the tile's activated pointwise result [BN/32 chunks][BM][32] bf16 (pixel operand of the third stage) and the trailing 1x1's
    // weights [BN/32 chunks][C3R][32] bf16, C3R = C3 rounded up to 16 (rows beyond C3 are the packed matrix's zero padding)
    unsigned char* const Ts = (unsigned char*)Bdw + (size_t)((p.C * 4 + 1023) / 1024) * 1024;
    unsigned char* const W3s = Ts + (size_t)(BN / 32) * BM * 64;
    const int C3R = TAIL ? ((p.C3 + 15) & ~15) : 0;
    float* const B3s = (float*)(W3s + (size_t)(BN / 32) * C3R * 64);      // [128] fp32 (1 KiB): the trailing 1x1's bias, zero behind C3

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WGM, wn = wave / WGM;
    const int fr = lane & 15, fc = lane >> 4;

    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, j = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int nt = bid % ntiles, j0 = bid / ntiles;
    const int n0 = nt * BN;
    const int num_tiles = p.B * tiles_h * tiles_w;

    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w_pw, 0, (int)p.wpw_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w_dw, 0, (int)(9 * p.C * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc((void*)p.b_dw, 0, (int)(p.C * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, (int)p.y_bytes, 0x00020000);

    float bias[FN][4];
#pragma unroll
    for (int a = 0; a < FN; ++a) {
        const int co = n0 + wn * WN + a * 16 + fc * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) bias[a][r] = (co + r < p.Cout) ? p.b_pw[co + r] : 0.f;
    }

    // ---- resident operands -----------------------------------------------------------------------------------------------
    {
        const int ninstr = nchunk * BN / 16;                           // pointwise weights: row rg = chunk*BN + n
        for (int ii = wave; ii < ninstr; ii += NW) {
            const int s = ii * 64 + lane;
            const int rg = s >> 2, pc = s & 3;
            const int ch = rg / BN, n = rg - ch * BN;
            const int c8 = pc ^ qswz(rg);
            const unsigned voff = (unsigned)(((n0 + n) * p.Kpad + ch * 32 + c8 * 8) * 2);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_void*)(Wpw + ii * 1024), 16, voff, 0, 0, 0);
        }
        for (int ii = wave; ii < wdw_instr; ii += NW) {                // depthwise weights: row = chunk*9 + tap <- [tap][C]
            const int s = ii * 64 + lane;
            const int row = s >> 2, c8 = s & 3;
            const int ch = row / 9, tap = row - ch * 9;
            const unsigned voff = (row < nchunk * 9) ? (unsigned)((tap * p.C + ch * 32 + c8 * 8) * 2) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(drs, (lds_void*)(Wdw + ii * 1024), 16, voff, 0, 0, 0);
        }
        const int b_instr = (p.C * 4 + 1023) / 1024;
        for (int ii = wave; ii < b_instr; ii += NW) {
            const unsigned voff = (unsigned)(ii * 1024 + lane * 16);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(brs, (lds_void*)((unsigned char*)Bdw + ii * 1024), 16, voff, 0, 0, 0);
        }
        if (TAIL) {                                                    // trailing 1x1: row rg = chunk*C3R + n, k = chunk*32 + 8*c8
            const __amdgpu_buffer_rsrc_t w3rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w3, 0, (int)p.w3_bytes, 0x00020000);
            const int n3 = (BN / 32) * C3R / 16;
            for (int ii = wave; ii < n3; ii += NW) {
                const int s = ii * 64 + lane;
                const int rg = s >> 2, pc = s & 3;
                const int ch = rg / C3R, n = rg - ch * C3R;
                const int c8 = pc ^ qswz(rg);
                const unsigned voff = (unsigned)((n * p.Kpad3 + ch * 32 + c8 * 8) * 2);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(w3rs, (lds_void*)(W3s + ii * 1024), 16, voff, 0, 0, 0);
            }
            if (wave == 0) {                                           // bias: 512 bytes, bytes past C3 * 4 come back as zeros (descriptor bound)
                const __amdgpu_buffer_rsrc_t b3rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.b3, 0, (int)(p.C3 * 4), 0x00020000);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(b3rs, (lds_void*)B3s, 16, (unsigned)(lane * 16) < 512u ? (unsigned)(lane * 16) : OOB, 0, 0, 0);
            }
        }
    }

    // ---- issue side: halo pieces ------------------------------------------------------------------------------------------
    unsigned hconst[LH];
    auto set_tile = [&](int tile) {
        int t = tile;
        const int tw = t % tiles_w; t /= tiles_w;
        const int th = t % tiles_h;
        const int b = t / tiles_h;
        const int h0 = th * TH, w0 = tw * 16;
#pragma unroll
        for (int j = 0; j < LH; ++j) {
            const int ii = wave * LH + j;
            const int s = ii * 64 + lane;
            const int hp = s >> 2, pc = s & 3;
            const int c8 = pc ^ qswz(hp);
            const int hy = hp / 18, hx = hp - hy * 18;
            const int hi = h0 - 1 + hy, wi = w0 - 1 + hx;
            const bool ok = (tile < num_tiles) && (ii < H_INSTR) && (hp < HP) && ((unsigned)hi < (unsigned)p.H) && ((unsigned)wi < (unsigned)p.W);
            hconst[j] = ok ? (unsigned)((((b * p.H + hi) * p.W + wi) * p.x_stride + p.x_coff) * 2 + c8 * 16) : OOB;
        }
    };
    int it_tile = j0, it_c = 0, it_slot = 0;
    set_tile(it_tile);
    auto issue_next = [&]() {
        unsigned char* dst = Hs + it_slot * HB;
        const unsigned coff = (unsigned)it_c * 64u;
#pragma unroll
        for (int j = 0; j < LH; ++j) {
            const int ii = wave * LH + j;
            const unsigned voff = (hconst[j] == OOB) ? OOB : hconst[j] + coff;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_void*)((ii < H_INSTR) ? dst + ii * 1024 : dump), 16, voff, 0, 0, 0);
        }
        it_slot = (it_slot + 1 == NSH) ? 0 : it_slot + 1;
        if (++it_c == nchunk) {
            it_c = 0;
            it_tile += G;
            set_tile(it_tile);
        }
    };
#pragma unroll
    for (int s = 0; s < NSH - 1; ++s) issue_next();
    wait_vmq<0>();
    __builtin_amdgcn_s_barrier();

    // Depthwise stage of one chunk ON THE MATRIX CORES: out[ch][px] = sum_taps w[tap][ch] * x[ch][px + tap] as MFMAs whose A operand is
    // a DIAGONAL weight fragment (exact zeros elsewhere, fp32 accumulate: same numbers as the VALU form, but no bf16 unpacking and a
    // fraction of its issue slots). Round 3: TWO taps per MFMA. The 32-deep k of v_mfma_f32_16x16x32_bf16 is split into two halves of 16
    // channels, k < 16 = channel k of tap A, k >= 16 = channel k - 16 of tap B: a lane of the pixel fragment (px fr, k group fc) reads its
    // 16 bytes from the halo pixel of tap (fc >> 1) at channel piece 2h + (fc & 1), the weight fragment carries w[tap A][c] at k = c and
    // w[tap B][c] at k = 16 + c. One MFMA then updates 16 channels x 16 pixels for two taps: 5 tap pairs x 2 channel halves = 10 MFMAs
    // and 10 fragment builds (one LDS dword + four ANDs each) per 32-channel chunk instead of 18 + 18.
    // A lane of the result holds 4 consecutive channels of one pixel, which go (bias + SiLU + bf16) with one ds_write_b64 into the
    // pixel-operand tile of the pointwise GEMM. Wave w owns output row w of the 8-row tile.
    static_assert(RPW == 1, "tap pairing is written for one depthwise row per wave");
    // per-lane constants of the diagonal fragments: row fr carries channel fr of the half; lane group fc holds k = 8fc..8fc+7, so the lane
    // has a non-zero entry only when (fc & 1) == fr >> 3, at dword (fr & 7) >> 1, halfword fr & 1 - the same halfword the channel occupies in
    // its LDS dword, so a fragment register is  (LDS dword of tap (fc >> 1)) & mask.
    unsigned dmask[4];
    {
        const bool valid = (fc & 1) == (fr >> 3);
        const unsigned hw = (fr & 1) ? 0xffff0000u : 0x0000ffffu;
#pragma unroll
        for (int q = 0; q < 4; ++q) dmask[q] = (valid && q == ((fr & 7) >> 1)) ? hw : 0u;
    }
    const int tapsel = fc >> 1;                        // which tap of a pair this lane's k range belongs to
    auto dw_stage = [&](int c, int hslot, int aslot) {
        const unsigned char* hsl = Hs + hslot * HB;
        unsigned char* asl = As + aslot * BM * 64;
        bf16x8 wd[5][2];
#pragma unroll
        for (int pr = 0; pr < 5; ++pr) {
            const int t = (pr == 4) ? 8 : 2 * pr + tapsel;            // (the ninth tap has no partner: its B half gets zero weights)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                unsigned wbits = *(const unsigned*)(Wdw + (c * 9 + t) * 64 + ((fr + 16 * h) & ~1) * 2);
                if (pr == 4 && tapsel) wbits = 0u;
                const uint4 v = make_uint4(wbits & dmask[0], wbits & dmask[1], wbits & dmask[2], wbits & dmask[3]);
                wd[pr][h] = *(const bf16x8*)&v;
            }
        }
        f32x4 dacc[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) dacc[h] = lds_read16_async((const unsigned char*)(Bdw + c * 32 + 16 * h + fc * 4));
#pragma unroll
        for (int pr = 0; pr < 5; ++pr) {
            const int tA = 2 * pr, tB = (pr == 4) ? 8 : 2 * pr + 1;
            const int hpA = (wave + tA / 3) * 18 + tA % 3, hpB = (wave + tB / 3) * 18 + tB % 3;
            const int hp = (tapsel ? hpB : hpA) + fr;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const bf16x8 xf = *(const bf16x8*)(hsl + swz64((unsigned)(hp * 64 + (2 * h + (fc & 1)) * 16)));
                dacc[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wd[pr][h], xf, dacc[h], 0, 0, 0);
            }
        }
        {
            const int px = wave * 16 + fr;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                __attribute__((aligned(8))) __bf16 o[4];
                float sv[4] = {dacc[h][0], dacc[h][1], dacc[h][2], dacc[h][3]};
                if (p.act_dw == ACT_SILU) silu4_packed(sv);
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = (__bf16)sv[i];
                const int c8 = 2 * h + (fc >> 1);
                lds_write8(asl + px * 64 + ((c8 ^ qswz(px)) * 16) + (fc & 1) * 8, *(const uint2*)o);
            }
        }
    };

    f32x4 acc[FN][FM];
    auto reset_acc = [&]() {
#pragma unroll
        for (int a = 0; a < FN; ++a)
#pragma unroll
            for (int b = 0; b < FM; ++b) acc[a][b] = f32x4{bias[a][0], bias[a][1], bias[a][2], bias[a][3]};
    };
    auto mfma_stage = [&](int c, int aslot) {
        const unsigned char* asl = As + aslot * BM * 64;
        bf16x8 wf[FN], xf[FM];
#pragma unroll
        for (int a = 0; a < FN; ++a) {
            const int rw = c * BN + wn * WN + a * 16 + fr;
            wf[a] = *(const bf16x8*)(Wpw + swz64((unsigned)(rw * 64 + fc * 16)));
        }
#pragma unroll
        for (int b = 0; b < FM; ++b) {
            const int row = wm * WM + b * 16 + fr;
            xf[b] = *(const bf16x8*)(asl + swz64((unsigned)(row * 64 + fc * 16)));
        }
#pragma unroll
        for (int a = 0; a < FN; ++a)
#pragma unroll
            for (int b = 0; b < FM; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[a], xf[b], acc[a][b], 0, 0, 0);
    };
    auto epilogue = [&](int tile) {
        int t = tile;
        const int tw = t % tiles_w; t /= tiles_w;
        const int th = t % tiles_h;
        const int b = t / tiles_h;
        const int wo = tw * 16 + fr;
#pragma unroll
        for (int bb = 0; bb < FM; ++bb) {
            const int ho = th * TH + wm * (WM / 16) + bb;
            const bool pix_ok = (ho < p.H) && (wo < p.W);
            const unsigned m = (unsigned)((b * p.H + ho) * p.W + wo);
#pragma unroll
            for (int a = 0; a < FN; ++a) {
                const int co = n0 + wn * WN + a * 16 + fc * 4;
                const bool ok = pix_ok && (co < p.Cout);
                float v[4] = {acc[a][bb][0], acc[a][bb][1], acc[a][bb][2], acc[a][bb][3]};
                if (p.act_pw == ACT_SILU) silu4_packed(v);
                if (TAIL) {
                    // the activated tile stays on chip: bf16 (the rounding the materialised tensor would have had) into the third stage's
                    // pixel-operand image, chunk = 32 output channels of this stage = 32 k of the next
                    __attribute__((aligned(8))) __bf16 o[4] = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                    const int px = (wm * (WM / 16) + bb) * 16 + fr;
                    const int cch = wn * WN + a * 16 + fc * 4;               // channel inside the BN block
                    const int c8 = (cch >> 3) & 3;
                    lds_write8(Ts + (size_t)(cch >> 5) * BM * 64 + px * 64 + ((c8 ^ qswz(px)) * 16) + (fc & 1) * 8, *(const uint2*)o);
                } else if (OUT_F32) {
                    const unsigned off = ok ? (m * (unsigned)p.y_stride + (unsigned)(p.y_coff + co)) * 4u : OOB;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, make_float4(v[0], v[1], v[2], v[3])), yrs, off, 0, 0);
                } else {
                    const unsigned off = ok ? (m * (unsigned)p.y_stride + (unsigned)(p.y_coff + co)) * 2u : OOB;
                    __attribute__((aligned(8))) __bf16 o[4] = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                    __builtin_amdgcn_raw_buffer_store_b64(*(const __attribute__((ext_vector_type(2))) unsigned*)o, yrs, off, 0, 0);
                }
            }
        }
    };

    // TAIL third stage, one phase (= one barrier) after the tile's epilogue put its activated result into Ts: wave w takes the tile's row w
    // (16 pixels) and ALL C3 output channels - C3R / 16 accumulator fragments - so a lane ends up with every channel of one pixel in
    // groups of four: bias, fp32 store, and the class maximum needs two cross-lane steps, no LDS.
    constexpr int F3MAX = 8;                                        // C3 <= 128
    const __amdgpu_buffer_rsrc_t y3rs = __builtin_amdgcn_make_buffer_rsrc((void*)(TAIL ? (void*)p.y3 : p.y), 0, (int)(TAIL ? p.y3_bytes : p.y_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc((void*)((TAIL && p.keys) ? (void*)p.keys : p.y), 0, (int)((TAIL && p.keys) ? (size_t)p.B * p.H * p.W * 4 : p.y_bytes), 0x00020000);
    constexpr int ST3 = F3MAX + 1;                                  // stores a wave issues in a third-stage phase (dummies included: fixed count)
    auto gemm3 = [&](int tile) {
        int t = tile;
        const int tw = t % tiles_w; t /= tiles_w;
        const int th = t % tiles_h;
        const int b = t / tiles_h;
        const int ho = th * TH + wave, wo = tw * 16 + fr;
        const bool pix_ok = (ho < p.H) && (wo < p.W);
        const unsigned m = (unsigned)((b * p.H + ho) * p.W + wo);
        const int nf3 = C3R >> 4, nk3 = p.Kpad3 >> 5;
        f32x4 a3[F3MAX];
#pragma unroll
        for (int f = 0; f < F3MAX; ++f) a3[f] = lds_read16_async((const unsigned char*)(B3s + f * 16 + fc * 4));   // bias rides in the accumulator
        for (int kc = 0; kc < nk3; ++kc) {
            const int row = wave * 16 + fr;
            const bf16x8 xf = *(const bf16x8*)(Ts + (size_t)kc * BM * 64 + swz64((unsigned)(row * 64 + fc * 16)));
#pragma unroll
            for (int f = 0; f < F3MAX; ++f)
                if (f < nf3) {
                    const int rw = kc * C3R + f * 16 + fr;
                    const bf16x8 wf = *(const bf16x8*)(W3s + swz64((unsigned)(rw * 64 + fc * 16)));
                    a3[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf, a3[f], 0, 0, 0);
                }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int f = 0; f < F3MAX; ++f) {
            const int co = f * 16 + fc * 4;
            const bool ok = pix_ok && f < nf3 && (co < p.C3);
            if (f < nf3 && co < p.C3) mx = fmaxf(mx, fmaxf(fmaxf(a3[f][0], a3[f][1]), fmaxf(a3[f][2], a3[f][3])));
            const unsigned off = ok ? (m * (unsigned)p.y3_stride + (unsigned)(p.y3_coff + co)) * 4u : OOB;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, make_float4(a3[f][0], a3[f][1], a3[f][2], a3[f][3])), y3rs, off, 0, 0);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        {   // sigmoid(max logit) exactly as anchor_max_level_kernel forms it (head.hip): the key the top-k kernel reads
            const float sg = 1.0f / (1.0f + expf(-mx));
            const unsigned off = (p.keys && pix_ok && fc == 0) ? m * 4u : OOB;
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(sg), krs, off, 0, 0);
        }
    };

    // ---- skewed pipeline over the flattened (tile, chunk) sequence --------------------------------------------------------
    unsigned long long clk[4] = {0, 0, 0, 0};
#define DP_STAMP(i) if (p.clk) { const unsigned long long now = __builtin_amdgcn_s_memtime(); clk[i] += now - last; last = now; }
    unsigned long long last = p.clk ? __builtin_amdgcn_s_memtime() : 0ull;
    int rd_slot = 0, g = 0;
    unsigned epmask = 0;                 // bit i: the wave issued its tile's stores i phases ago (S of them, TAIL: ST3)
    constexpr int SS = TAIL ? ST3 : S;
    static_assert(LH + 2 * SS < 64, "vmcnt immediate");
    bool first = true;
    int prev_tile = -1, prev_c = 0, tail_tile = -1;
    reset_acc();
    for (int tile = j0; tile < num_tiles; tile += G) {
        for (int c = 0; c < nchunk; ++c, ++g) {
            if (!first) {
                const int k = __builtin_popcount(epmask & 3u);
                if (k == 0) wait_vmq<LH>();
                else if (k == 1) wait_vmq<LH + SS>();
                else wait_vmq<LH + 2 * SS>();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // this wave's A-tile (and TAIL: T-tile) writes of the previous phase are done
                __builtin_amdgcn_s_barrier();
            }
            first = false;
            DP_STAMP(0)
            issue_next();
            epmask <<= 1;
            dw_stage(c, rd_slot, g & 1);
            DP_STAMP(1)
            if (TAIL && tail_tile >= 0) {                                  // third stage of the tile whose epilogue ran in the previous phase
                gemm3(tail_tile);
                tail_tile = -1;
                epmask |= 1u;
            }
            if (prev_tile >= 0) {
                mfma_stage(prev_c, (g - 1) & 1);
                DP_STAMP(2)
                if (prev_c == nchunk - 1) {
                    epilogue(prev_tile);
                    reset_acc();
                    if (TAIL) tail_tile = prev_tile;
                    else epmask |= 1u;
                    DP_STAMP(3)
                }
            }
            prev_tile = tile;
            prev_c = c;
            rd_slot = (rd_slot + 1 == NSH) ? 0 : rd_slot + 1;
        }
    }
    // drain: MFMA + epilogue of the last chunk (TAIL: + its third stage behind one more barrier; nchunk >= 2, so no third stage is pending here)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (prev_tile >= 0) {
        mfma_stage(prev_c, (g - 1) & 1);
        epilogue(prev_tile);
        if (TAIL) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            gemm3(prev_tile);
        }
    }
    wait_vmq<0>();
    if (p.clk && lane == 0)
        for (int i = 0; i < 4; ++i) p.clk[((size_t)blockIdx.x * NW + wave) * 4 + i] = clk[i];
}

// ---------------------------------------------------------------------------------------------------------------
static size_t dwpw_lds(int TH, int BN, int C, int C3 = 0) {
    const int HP = (TH + 2) * 18, H_INSTR = (HP * 4 + 63) / 64;
    const int nchunk = C / 32;
    size_t b = (size_t)3 * H_INSTR * 1024 + (size_t)2 * TH * 16 * 64 + 1024 + (size_t)nchunk * BN * 64 +
               (size_t)((nchunk * 9 + 15) / 16) * 1024 + (size_t)((C * 4 + 1023) / 1024) * 1024;
    if (C3 > 0) b += (size_t)(BN / 32) * TH * 16 * 64 + (size_t)(BN / 32) * ((C3 + 15) / 16 * 16) * 64 + 1024;     // Ts, W3s, B3s
    return b;
}

static bool dwpw_tail(const DwPwParams& p) { return p.w3 != nullptr; }

static int dwpw_bn(const DwPwParams& p) {          // widest output-channel block whose weights fit next to the rings
    for (int bn : {128, 64}) {
        if (bn > ((p.Cout + 63) / 64) * 64) continue;
        if (dwpw_lds(8, bn, p.C, dwpw_tail(p) ? p.C3 : 0) <= 150 * 1024) return bn;
    }
    return 0;
}

bool conv_dwpw_valid(const DwPwParams& p) {
    if (dwpw_stream_valid(p)) return true;
    if ((p.C % 32) != 0 || p.Kpad != p.C || (p.Cout & 3) || (p.y_stride & 3) || (p.y_coff & 3)) return false;
    if (p.x_bytes >= (1ull << 31) || p.y_bytes >= (1ull << 31) || p.wpw_bytes >= (1ull << 31)) return false;
    if ((long)((p.H + 7) / 8 * 8) * ((p.W + 15) / 16 * 16) * 2 > (long)p.H * p.W * 3) return false;
    const int bn = dwpw_bn(p);
    if (bn == 0) return false;
    // every output-channel block repeats the depthwise stage: with more than two blocks (v10-X: 320 -> 640 in 64-wide blocks) the
    // fused form costs more than the two kernels it replaces (measured 72 us vs ~35 us per pair at 40x40, bs 8)
    if ((p.Cout + bn - 1) / bn > 2) return false;
    if (dwpw_tail(p)) {
        // third stage: needs every channel of the pointwise result in one workgroup, >= 2 chunks (its tile image is single-buffered, read one
        // phase after it is written), at most 128 logit channels in groups of four, fp32 output
        if (p.Cout > bn || p.C < 64 || p.C3 <= 0 || p.C3 > 128 || (p.C3 & 3) || (p.y3_stride & 3) || (p.y3_coff & 3) || !p.y3 || !p.b3) return false;
        if (p.Kpad3 != (p.Cout + 31) / 32 * 32 || p.y3_bytes >= (1ull << 31) || p.w3_bytes >= (1ull << 31) || (size_t)p.B * p.H * p.W * 4 >= (1ull << 31)) return false;
    }
    return true;
}

const char* conv_dwpw_kernel_name(const DwPwParams& p) {
    if (dwpw_stream_valid(p)) return "conv_dwpw_stream_kernel";
    const int bn = dwpw_bn(p);
    if (dwpw_tail(p)) return bn == 128 ? "conv_dwpw_kernel<8,128,tail,8>" : "conv_dwpw_kernel<8,64,tail,8>";
    if (p.out_f32) return bn == 128 ? "conv_dwpw_kernel<8,128,true,8>" : "conv_dwpw_kernel<8,64,true,8>";
    return bn == 128 ? "conv_dwpw_kernel<8,128,false,8>" : "conv_dwpw_kernel<8,64,false,8>";
}

template <int BN, bool OUT_F32, bool TAIL = false>
static hipError_t launch_dwpw_t(const DwPwParams& p, hipStream_t st) {
    constexpr int TH = 8, NW = 8;       // (16-row tiles - two depthwise rows per wave - measured no faster: 20 % less work per row, lost to tile imbalance)
    const size_t sh = dwpw_lds(TH, BN, p.C, TAIL ? p.C3 : 0);
    const int tiles_h = (p.H + TH - 1) / TH, tiles_w = (p.W + 15) / 16, ntiles = (p.Cout + BN - 1) / BN;
    const int num_tiles = p.B * tiles_h * tiles_w;
    int G = 256 / ntiles;
    if (sh <= 80 * 1024 && NW == 4) G *= 2;
    if (G < 1) G = 1;
    if (G > num_tiles) G = num_tiles;
    auto kern = conv_dwpw_kernel<TH, BN, OUT_F32, NW, TAIL>;
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
        if (e != hipSuccess) return e;
        attr = true;
    }
    static const bool clocks = [] { const char* v = std::getenv("YOLOP_DWPW_CLOCKS"); return v && *v == '1'; }();   // debug: per-phase s_memtime sums
    if (clocks) {
        DwPwParams q = p;
        const size_t n = (size_t)G * ntiles * NW * 4;
        if (hipMalloc((void**)&q.clk, n * 8) != hipSuccess) return hipErrorOutOfMemory;
        hipLaunchKernelGGL(kern, dim3(G * ntiles), dim3(NW * 64), sh, st, q, tiles_h, tiles_w, ntiles, G);
        (void)hipStreamSynchronize(st);
        std::vector<unsigned long long> h(n);
        (void)hipMemcpy(h.data(), q.clk, n * 8, hipMemcpyDeviceToHost);
        (void)hipFree(q.clk);
        static const char* nm[4] = {"wait+barrier", "issue+dw", "mfma", "epilogue"};
        const double steps = (double)num_tiles / G * (p.C / 32);
        for (int w = 0; w < NW; w += NW - 1) {
            fprintf(stderr, "[dwpw clocks] C=%d Cout=%d %dx%d wave %d, s_memtime ticks per chunk step:", p.C, p.Cout, p.H, p.W, w);
            for (int i = 0; i < 4; ++i) {
                double s = 0;
                for (int g = 0; g < G * ntiles; ++g) s += (double)h[((size_t)g * NW + w) * 4 + i];
                fprintf(stderr, " %s %.0f", nm[i], s / (G * ntiles) / steps);
            }
            fprintf(stderr, "\n");
        }
        return hipGetLastError();
    }
    hipLaunchKernelGGL(kern, dim3(G * ntiles), dim3(NW * 64), sh, st, p, tiles_h, tiles_w, ntiles, G);
    return hipGetLastError();
}

hipError_t launch_conv_dwpw(const DwPwParams& p, hipStream_t st) {
    if (dwpw_stream_valid(p)) return launch_dwpw_stream(p, st);
    const int bn = dwpw_bn(p);
    if (dwpw_tail(p)) return bn == 128 ? launch_dwpw_t<128, false, true>(p, st) : launch_dwpw_t<64, false, true>(p, st);
    if (bn == 128) return p.out_f32 ? launch_dwpw_t<128, true>(p, st) : launch_dwpw_t<128, false>(p, st);
    return p.out_f32 ? launch_dwpw_t<64, true>(p, st) : launch_dwpw_t<64, false>(p, st);
}

}  // namespace yp
