// Persistent form of conv_dma.hip (bf16 implicit-GEMM conv on MFMA with an LDS-DMA operand ring): a workgroup keeps its
// output-channel block and walks pixel tiles j, j+G, j+2G, ...; the k-step ring runs straight across tile boundaries, so
// the loads of the next tile are in flight while the current tile finishes and stores. PMC profiles (profiles/r01_*)
// showed the one-tile-per-workgroup form spending most of a wave's life in its prologue / first-load latency / store tail.
//
// vmcnt accounting (see conv_halo_p.hip): per wave the queue holds LPW LDS-DMA loads per k-step and S = FM*FN epilogue
// stores per tile, the stores issued unconditionally through a buffer descriptor (out-of-range offset when a lane has
// nothing to store). Iteration g: wait(g) ; barrier ; issue(g+NS-1) ; compute(g) ; [stores if g ends a tile].
// Ops younger than L(g) at wait(g): L(g+1..g+NS-2) plus the stores of every tile end among the previous NS-1 iterations
// -> s_waitcnt vmcnt((NS-2)*LPW + k*S), k = 0..NS-1.
//
// PIPE form (software-pipelined fragments). PMC on the 128->128 3x3 @40x40 layer: 1650 cycles per k-step for 512 cycles
// of MFMA - with one workgroup per CU every wave passes the barrier at the same time, so the LDS-read burst of a k-step
// and its MFMA burst alternate instead of overlapping. PIPE keeps two sets of fragment registers: iteration g reads the
// fragments of k-step g+1 (whose stage is waited for one iteration earlier) while the MFMAs of k-step g run from the
// registers read in iteration g-1. Ops younger than L(g+1) at wait(g): L(g+2..g+NS-2) plus the stores of the tile ends
// among the previous NS-2 iterations -> vmcnt((NS-3)*LPW + k*S).
#include "common.h"

namespace yp {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ float silu_f2(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
template <int N> __device__ __forceinline__ void wait_vmp() {
    static_assert(N >= 0 && N < 64, "vmcnt range");
    __builtin_amdgcn_s_waitcnt((N & 0xF) | (7 << 4) | (0xF << 8) | (((N >> 4) & 3) << 14));
}
template <int BK> __device__ __forceinline__ int swz_p(int row) {
    return BK == 32 ? (((row >> 2) & 1) << 1) : ((row >> 1) & 7);
}

template <int BM, int BN, int WGM, int WGN, int BK, int NS, bool HAS_RES, bool OUT_F32, bool WRES, bool PIPE, bool PP>
__global__ __launch_bounds__(WGM * WGN * 64) void conv_dma_p_kernel(const ConvParams p, const int mtiles, const int ntiles, const int G) {
    constexpr int NW = WGM * WGN;
    constexpr int CPR = BK / 8;
    constexpr int RB = BK * 2;
    constexpr int A_INSTR = BM * CPR / 64;
    constexpr int W_INSTR = BN * CPR / 64;
    constexpr int A_IPW = A_INSTR / NW;
    constexpr int W_IPW = (W_INSTR + NW - 1) / NW;
    constexpr int LPW = A_IPW + (WRES ? 0 : W_IPW);   // WRES: the weight block is LDS-resident, only pixels stream
    constexpr int SB = (WRES ? BM : (BM + BN)) * RB;
    constexpr int WM = BM / WGM, WN = BN / WGN, FM = WM / 16, FN = WN / 16;
    constexpr int KSUB = BK / 32;
    constexpr int S = FM * FN;
    static_assert((NW == 4 || NW == 8) && A_INSTR % NW == 0 && A_IPW >= 1, "tile/wave layout");
    static_assert((NS - 2) * LPW + (NS - 1) * S < 64, "vmcnt immediate");
    static_assert(!PIPE || NS >= 4, "the pipelined form gives up one stage of prefetch distance");
    constexpr unsigned OOB = 0x80000000u;

    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];

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
    const int HoWo = p.Ho * p.Wo;

    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t xrs2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x2_C > 0 ? p.x2 : p.x), 0, (int)(p.x2_C > 0 ? p.x2_bytes : p.x_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, (int)p.y_bytes, 0x00020000);

    float bias[FN][4];
#pragma unroll
    for (int a = 0; a < FN; ++a) {
        const int co = n0 + wn * WN + a * 16 + fc * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) bias[a][r] = (co + r < p.Cout) ? p.bias[co + r] : 0.f;
    }

    // ---- WRES: the whole [BN][Kpad] weight block of this workgroup goes to LDS once, as nk tiles of [BN][BK] -------
    unsigned char* const Wres = smem + NS * SB + 1024;
    if (WRES) {
        const int nkw = p.Kpad / BK;
        const int ninstr = nkw * BN * CPR / 64;
        for (int ii = wave; ii < ninstr; ii += NW) {
            const int s = ii * 64 + lane;
            const int rowg = s / CPR, pc = s - rowg * CPR;
            const int kt = rowg / BN, n = rowg - kt * BN;
            const int c = pc ^ swz_p<BK>(n);
            const unsigned voff = (unsigned)(((n0 + n) * p.Kpad + kt * BK + c * 8) * 2);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_void*)(Wres + ii * 1024), 16, voff, 0, 0, 0);
        }
    }

    // ---- issue side -----------------------------------------------------------------------------------------
    unsigned aconst[A_IPW], amask[A_IPW], aconst2[A_IPW];   // aconst2: the same row in the folded-upsample source (p.x2)
    auto set_tile = [&](int mt) {
#pragma unroll
        for (int j = 0; j < A_IPW; ++j) {
            const int s = (wave * A_IPW + j) * 64 + lane;
            const int row = s / CPR, pc = s - row * CPR;
            const int c = pc ^ swz_p<BK>(row);
            const int m = mt * BM + row;
            unsigned mask = 0, base = 0;
            if (mt < mtiles && m < p.M && p.dbg != 2) {
                if (p.ks == 1) {
                    base = (unsigned)(m * p.x_stride + p.x_coff) * 2u;
                    mask = 1u;
                    if (p.x2_C > 0) {
                        const int b = m / HoWo, r = m - b * HoWo;
                        const int ho = r / p.Wo, wo = r - ho * p.Wo;
                        aconst2[j] = (unsigned)(((b * p.x2_H + (ho >> 1)) * p.x2_W + (wo >> 1)) * p.x2_stride + p.x2_coff) * 2u + (unsigned)c * 16u;
                    }
                } else {
                    const int b = m / HoWo, r = m - b * HoWo;
                    const int ho = r / p.Wo, wo = r - ho * p.Wo;
                    const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
                    base = (unsigned)(((b * p.H + hi0) * p.W + wi0) * p.x_stride + p.x_coff) * 2u;
                    for (int ky = 0; ky < p.ks; ++ky)
                        for (int kx = 0; kx < p.ks; ++kx)
                            if ((unsigned)(hi0 + ky) < (unsigned)p.H && (unsigned)(wi0 + kx) < (unsigned)p.W)
                                mask |= 1u << (ky * p.ks + kx);
                }
            }
            aconst[j] = base + (unsigned)c * 16u;
            amask[j] = mask;
        }
    };
    unsigned wconst[W_IPW];
#pragma unroll
    for (int j = 0; j < W_IPW; ++j) {
        const int ii = wave * W_IPW + j;
        const int s = ii * 64 + lane;
        const int row = s / CPR, pc = s - row * CPR;
        const int c = pc ^ swz_p<BK>(row);
        wconst[j] = (ii < W_INSTR) ? (unsigned)(((n0 + row) * p.Kpad + c * 8) * 2) : OOB;
    }
    const int nk = p.Kpad / BK;
    int it_tile = j0, it_kt = 0, it_slot = 0;
    int is_tap = 0, is_ky = 0, is_kx = 0, is_kc = 0;
    set_tile(it_tile);
    auto issue_next = [&]() {
        const unsigned tapoff = (unsigned)(((is_ky * p.W + is_kx) * p.x_stride + is_kc) * 2);
        unsigned char* sbase = smem + it_slot * SB;
        const bool src2 = p.x2_C > 0 && is_kc < p.x2_C;      // this k-step's channels come from the low-resolution source
        const __amdgpu_buffer_rsrc_t ars = src2 ? xrs2 : xrs;
#pragma unroll
        for (int j = 0; j < A_IPW; ++j) {
            const unsigned voff = ((amask[j] >> is_tap) & 1u) ? ((src2 ? aconst2[j] : aconst[j]) + tapoff) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ars, (lds_void*)(sbase + (wave * A_IPW + j) * 1024), 16, voff, 0, 0, 0);
        }
        const bool live = it_tile < mtiles;
        if (!WRES)
#pragma unroll
        for (int j = 0; j < W_IPW; ++j) {
            const int ii = wave * W_IPW + j;
            unsigned char* dst = (ii < W_INSTR) ? (sbase + BM * RB + ii * 1024) : (smem + NS * SB);
            const unsigned voff = (wconst[j] == OOB || !live) ? OOB : (wconst[j] + (unsigned)(it_kt * BK) * 2u);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_void*)dst, 16, voff, 0, 0, 0);
        }
        it_slot = (it_slot + 1 == NS) ? 0 : it_slot + 1;
        is_kc += BK;
        if (is_kc >= p.Cin) {
            is_kc = 0;
            ++is_tap;
            if (++is_kx == p.ks) { is_kx = 0; ++is_ky; }
        }
        if (++it_kt == nk) {
            it_kt = 0; is_tap = 0; is_ky = 0; is_kx = 0; is_kc = 0;
            it_tile += G;
            set_tile(it_tile);
        }
    };

    // fragment read offsets
    int aoff[KSUB], woff[KSUB];
#pragma unroll
    for (int ss = 0; ss < KSUB; ++ss) {
        const int ra = wm * WM + fr, rw = wn * WN + fr;
        aoff[ss] = ra * RB + (((ss * 4 + fc) ^ swz_p<BK>(ra)) * 16);
        woff[ss] = (WRES ? 0 : BM * RB) + rw * RB + (((ss * 4 + fc) ^ swz_p<BK>(rw)) * 16);
    }

#pragma unroll
    for (int s = 0; s < NS - 1; ++s) issue_next();
    if (WRES) {                           // one-time drain: resident weights (and the first stages) have landed
        wait_vmp<0>();
        __builtin_amdgcn_s_barrier();
    }

    int rslot = 0;
    unsigned epmask = 0;
    bf16x8 cw[KSUB][FN], cx[KSUB][FM];            // PIPE: fragments of the current k-step
    auto load_frags = [&](int slot, int ktn, bf16x8 (&wf)[KSUB][FN], bf16x8 (&xf)[KSUB][FM]) {
        const unsigned char* sb = smem + slot * SB;
        const unsigned char* wb_ = WRES ? (Wres + ktn * BN * RB) : sb;
#pragma unroll
        for (int ss = 0; ss < KSUB; ++ss) {
#pragma unroll
            for (int a = 0; a < FN; ++a) wf[ss][a] = *(const bf16x8*)(wb_ + woff[ss] + a * 16 * RB);
#pragma unroll
            for (int b = 0; b < FM; ++b) xf[ss][b] = *(const bf16x8*)(sb + aoff[ss] + b * 16 * RB);
        }
    };
    f32x4 acc[FN][FM];
    auto reset_acc = [&]() {
#pragma unroll
        for (int a = 0; a < FN; ++a)
#pragma unroll
            for (int b = 0; b < FM; ++b) acc[a][b] = f32x4{bias[a][0], bias[a][1], bias[a][2], bias[a][3]};   // bias rides in the accumulator
    };
    // epilogue of one tile: exactly S buffer stores per wave
    auto store_tile = [&](int tile) {
    const int m0 = tile * BM;
        uint2 rres[FM][FN];
        if (HAS_RES) {
#pragma unroll
            for (int b = 0; b < FM; ++b) {
                const int m = m0 + wm * WM + b * 16 + fr;
#pragma unroll
                for (int a = 0; a < FN; ++a) {
                    const int co = n0 + wn * WN + a * 16 + fc * 4;
                    rres[b][a] = (m < p.M && co < p.Cout)
                                     ? *(const uint2*)((const __bf16*)p.res + (size_t)m * p.res_stride + p.res_coff + co)
                                     : make_uint2(0u, 0u);
                }
            }
        }
#pragma unroll
        for (int b = 0; b < FM; ++b) {
            const int m = m0 + wm * WM + b * 16 + fr;
#pragma unroll
            for (int a = 0; a < FN; ++a) {
                const int co = n0 + wn * WN + a * 16 + fc * 4;
                const bool ok = (m < p.M) && (co < p.Cout) && (p.dbg != 1);
                float v[4] = {acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]};
                if (p.act == ACT_SILU) silu4_packed(v);
                if (HAS_RES) {
                    const uint2 rr = rres[b][a];
                    v[0] += __uint_as_float(rr.x << 16); v[1] += __uint_as_float(rr.x & 0xffff0000u);
                    v[2] += __uint_as_float(rr.y << 16); v[3] += __uint_as_float(rr.y & 0xffff0000u);
                }
                if (OUT_F32) {
                    const unsigned off = ok ? ((unsigned)m * (unsigned)p.y_stride + (unsigned)(p.y_coff + co)) * 4u : OOB;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, make_float4(v[0], v[1], v[2], v[3])), yrs, off, 0, 0);
                } else {
                    const unsigned off = ok ? ((unsigned)m * (unsigned)p.y_stride + (unsigned)(p.y_coff + co)) * 2u : OOB;
                    __attribute__((aligned(8))) __bf16 o[4] = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                    __builtin_amdgcn_raw_buffer_store_b64(*(const __attribute__((ext_vector_type(2))) unsigned*)o, yrs, off, 0, 0);
                }
            }
        }
    };

    if constexpr (PP) {
        // ---- ping-pong schedule (8 waves: two per SIMD). Waves 0-3 and waves 4-7 alternate roles between barriers: in
        // interval A(g) the first half issues its LDS-DMA pieces of stage g+NS-1 and reads its fragments of stage g while the
        // second half runs the MFMAs of k-step g-1; in B(g) the roles swap. Every SIMD then always has one wave in a matrix
        // burst and its partner in a load segment, instead of both bursting and both loading together (PMC: 32 % matrix pipe).
        // Both halves pass the same two barriers per k-step. Visibility of stage g+1: every wave waits for its own pieces
        // before the barrier that ends B(g); ops younger than L(g+1) there: L(g+2..g+NS-1) (the last one issued in this
        // k-step) plus the stores of the tile ends of the last NS-1 (first half) / NS-2 (second half) k-steps.
        static_assert(NW == 8 && !PIPE, "ping-pong pairs the two waves of each SIMD");
        static_assert((NS - 2) * LPW + (NS - 1) * S < 64, "vmcnt immediate");
        if (!WRES) {
            wait_vmp<(NS - 2) * LPW>();           // stage 0 landed
            __builtin_amdgcn_s_barrier();
        }
        auto mfma_all = [&]() {
#pragma unroll
            for (int ss = 0; ss < KSUB; ++ss)
#pragma unroll
                for (int a = 0; a < FN; ++a)
#pragma unroll
                    for (int b = 0; b < FM; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cw[ss][a], cx[ss][b], acc[a][b], 0, 0, 0);
        };
        auto wait_stage = [&](int k) {
            if (k == 0) wait_vmp<(NS - 2) * LPW>();
            else if (k == 1) wait_vmp<(NS - 2) * LPW + S>();
            else if (k == 2) wait_vmp<(NS - 2) * LPW + 2 * S>();
            else wait_vmp<(NS - 2) * LPW + (NS >= 4 ? 3 : 2) * S>();
        };
        reset_acc();
        if (wave < NW / 2) {
            for (int tile = j0; tile < mtiles; tile += G) {
                for (int kt = 0; kt < nk; ++kt) {
                    issue_next();
                    epmask <<= 1;
                    load_frags(rslot, kt, cw, cx);
                    rslot = (rslot + 1 == NS) ? 0 : rslot + 1;
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();                                  // end of A(g)
                    mfma_all();
                    if (kt == nk - 1) {
                        store_tile(tile);
                        reset_acc();
                        epmask |= 1u;
                    }
                    wait_stage(__builtin_popcount(epmask & ((1u << (NS - 1)) - 1u)));
                    __builtin_amdgcn_s_barrier();                                  // end of B(g)
                }
            }
        } else {
            int prev_tile = -1;
            bool prev_last = false;
            for (int tile = j0; tile < mtiles; tile += G) {
                for (int kt = 0; kt < nk; ++kt) {
                    if (prev_tile >= 0) {
                        mfma_all();
                        if (prev_last) {
                            store_tile(prev_tile);
                            reset_acc();
                            epmask |= 1u;
                        }
                    }
                    __builtin_amdgcn_s_barrier();                                  // end of A(g)
                    issue_next();
                    load_frags(rslot, kt, cw, cx);
                    rslot = (rslot + 1 == NS) ? 0 : rslot + 1;
                    wait_stage(__builtin_popcount(epmask & ((1u << (NS - 2)) - 1u)));
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();                                  // end of B(g)
                    epmask <<= 1;
                    prev_tile = tile;
                    prev_last = (kt == nk - 1);
                }
            }
            if (prev_tile >= 0) {
                mfma_all();
                store_tile(prev_tile);
            }
        }
        wait_vmp<0>();
        return;
    }

    if (PIPE) {
        if (!WRES) {
            wait_vmp<(NS - 2) * LPW>();           // stage 0 landed
            __builtin_amdgcn_s_barrier();
        }
        load_frags(0, 0, cw, cx);
        rslot = 1;
    }
    for (int tile = j0; tile < mtiles; tile += G) {
        reset_acc();
        if (PIPE) {
            for (int kt = 0; kt < nk; ++kt) {
                {
                    const int k = __builtin_popcount(epmask & ((1u << (NS - 2)) - 1u));
                    if (k == 0) wait_vmp<(NS - 3) * LPW>();
                    else if (k == 1) wait_vmp<(NS - 3) * LPW + S>();
                    else wait_vmp<(NS - 3) * LPW + 2 * S>();
                }
                __builtin_amdgcn_s_barrier();
                issue_next();
                epmask <<= 1;
                bf16x8 nw_[KSUB][FN], nx_[KSUB][FM];
                load_frags(rslot, (kt + 1 == nk) ? 0 : kt + 1, nw_, nx_);
#pragma unroll
                for (int ss = 0; ss < KSUB; ++ss)
#pragma unroll
                    for (int a = 0; a < FN; ++a)
#pragma unroll
                        for (int b = 0; b < FM; ++b)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cw[ss][a], cx[ss][b], acc[a][b], 0, 0, 0);
#pragma unroll
                for (int ss = 0; ss < KSUB; ++ss) {
#pragma unroll
                    for (int a = 0; a < FN; ++a) cw[ss][a] = nw_[ss][a];
#pragma unroll
                    for (int b = 0; b < FM; ++b) cx[ss][b] = nx_[ss][b];
                }
                rslot = (rslot + 1 == NS) ? 0 : rslot + 1;
            }
        } else
        for (int kt = 0; kt < nk; ++kt) {
            {
                const int k = __builtin_popcount(epmask & ((1u << (NS - 1)) - 1u));
                if (k == 0) wait_vmp<(NS - 2) * LPW>();
                else if (k == 1) wait_vmp<(NS - 2) * LPW + S>();
                else if (k == 2) wait_vmp<(NS - 2) * LPW + 2 * S>();
                else wait_vmp<(NS - 2) * LPW + (NS >= 4 ? 3 : 2) * S>();
            }
            __builtin_amdgcn_s_barrier();
            issue_next();
            epmask <<= 1;
            const unsigned char* sb = smem + rslot * SB;
            const unsigned char* wb_ = WRES ? (Wres + kt * BN * RB) : sb;
#pragma unroll
            for (int ss = 0; ss < KSUB; ++ss) {
                bf16x8 wf[FN], xf[FM];
#pragma unroll
                for (int a = 0; a < FN; ++a) wf[a] = *(const bf16x8*)(wb_ + woff[ss] + a * 16 * RB);
#pragma unroll
                for (int b = 0; b < FM; ++b) xf[b] = *(const bf16x8*)(sb + aoff[ss] + b * 16 * RB);
#pragma unroll
                for (int a = 0; a < FN; ++a)
#pragma unroll
                    for (int b = 0; b < FM; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[a], xf[b], acc[a][b], 0, 0, 0);
            }
            rslot = (rslot + 1 == NS) ? 0 : rslot + 1;
        }

        store_tile(tile);
        epmask |= 1u;
    }
    wait_vmp<0>();
}

// ---------------------------------------------------------------------------------------------------------------
struct DmaPCfg { int BM, BN, NW, BK, NS; const char* name; int wres; int pipe; int pp; };
static const DmaPCfg kP[] = {
    {128, 32, 4, 32, 4, "conv_dma_p_kernel<128,32,4,1,32,4>"},     // 0
    {128, 64, 4, 32, 4, "conv_dma_p_kernel<128,64,2,2,32,4>"},     // 1
    {128, 128, 4, 32, 4, "conv_dma_p_kernel<128,128,2,2,32,4>"},   // 2
    {64, 64, 4, 32, 4, "conv_dma_p_kernel<64,64,2,2,32,4>"},       // 3
    {256, 64, 8, 32, 4, "conv_dma_p_kernel<256,64,4,2,32,4>"},     // 4
    {256, 128, 8, 32, 4, "conv_dma_p_kernel<256,128,4,2,32,4>"},   // 5
    {128, 128, 4, 64, 3, "conv_dma_p_kernel<128,128,2,2,64,3>"},   // 6
    {128, 64, 4, 64, 3, "conv_dma_p_kernel<128,64,2,2,64,3>"},     // 7
    {128, 256, 8, 32, 4, "conv_dma_p_kernel<128,256,2,4,32,4>"},   // 8
    {64, 128, 4, 64, 3, "conv_dma_p_kernel<64,128,2,2,64,3>"},     // 9
    {64, 64, 4, 64, 4, "conv_dma_p_kernel<64,64,2,2,64,4>"},       // 10
    {128, 64, 8, 64, 4, "conv_dma_p_kernel<128,64,4,2,64,4>"},     // 11
    // weight-resident forms (ids 12..): the [BN][K] weight block stays in LDS, only the pixel tile streams
    {128, 64, 4, 32, 4, "conv_dma_p_kernel<128,64,2,2,32,4,W>", 1},   // 12
    {128, 64, 4, 64, 4, "conv_dma_p_kernel<128,64,2,2,64,4,W>", 1},   // 13
    {64, 64, 4, 32, 4, "conv_dma_p_kernel<64,64,2,2,32,4,W>", 1},     // 14
    {64, 64, 4, 64, 4, "conv_dma_p_kernel<64,64,2,2,64,4,W>", 1},     // 15
    {128, 128, 4, 64, 3, "conv_dma_p_kernel<128,128,2,2,64,3,W>", 1}, // 16
    {128, 128, 8, 64, 4, "conv_dma_p_kernel<128,128,4,2,64,4,W>", 1}, // 17
    {128, 32, 4, 32, 4, "conv_dma_p_kernel<128,32,4,1,32,4,W>", 1},   // 18
    {256, 64, 8, 64, 3, "conv_dma_p_kernel<256,64,4,2,64,3,W>", 1},   // 19
    {128, 256, 8, 64, 3, "conv_dma_p_kernel<128,256,2,4,64,3,W>", 1}, // 20
    // software-pipelined fragment forms (ids 21..)
    {256, 128, 8, 32, 4, "conv_dma_p_kernel<256,128,4,2,32,4,P>", 0, 1},     // 21
    {256, 64, 8, 32, 4, "conv_dma_p_kernel<256,64,4,2,32,4,P>", 0, 1},       // 22
    {128, 256, 8, 32, 4, "conv_dma_p_kernel<128,256,2,4,32,4,P>", 0, 1},     // 23
    {128, 128, 4, 32, 4, "conv_dma_p_kernel<128,128,2,2,32,4,P>", 0, 1},     // 24
    {128, 64, 4, 32, 4, "conv_dma_p_kernel<128,64,2,2,32,4,P>", 0, 1},       // 25
    {64, 64, 4, 32, 4, "conv_dma_p_kernel<64,64,2,2,32,4,P>", 0, 1},         // 26
    {64, 64, 4, 64, 4, "conv_dma_p_kernel<64,64,2,2,64,4,P>", 0, 1},         // 27
    {128, 64, 8, 64, 4, "conv_dma_p_kernel<128,64,4,2,64,4,P>", 0, 1},       // 28
    {128, 128, 8, 64, 4, "conv_dma_p_kernel<128,128,4,2,64,4,W,P>", 1, 1},   // 29
    {64, 64, 4, 64, 4, "conv_dma_p_kernel<64,64,2,2,64,4,W,P>", 1, 1},       // 30
    {128, 64, 4, 64, 4, "conv_dma_p_kernel<128,64,2,2,64,4,W,P>", 1, 1},     // 31
    {128, 128, 8, 64, 4, "conv_dma_p_kernel<128,128,4,2,64,4,P>", 0, 1},     // 32
    // ping-pong forms (ids 33..): the two waves of each SIMD alternate between a load segment and a matrix burst
    {256, 128, 8, 32, 4, "conv_dma_p_kernel<256,128,4,2,32,4,Q>", 0, 0, 1},     // 33
    {256, 64, 8, 32, 4, "conv_dma_p_kernel<256,64,4,2,32,4,Q>", 0, 0, 1},       // 34
    {128, 256, 8, 32, 4, "conv_dma_p_kernel<128,256,2,4,32,4,Q>", 0, 0, 1},     // 35
    {128, 128, 8, 64, 4, "conv_dma_p_kernel<128,128,4,2,64,4,Q>", 0, 0, 1},     // 36
    {128, 64, 8, 64, 4, "conv_dma_p_kernel<128,64,4,2,64,4,Q>", 0, 0, 1},       // 37
    {128, 128, 8, 64, 4, "conv_dma_p_kernel<128,128,4,2,64,4,W,Q>", 1, 0, 1},   // 38
    {256, 64, 8, 64, 3, "conv_dma_p_kernel<256,64,4,2,64,3,W,Q>", 1, 0, 1},     // 39
    {256, 128, 8, 64, 3, "conv_dma_p_kernel<256,128,4,2,64,3,Q>", 0, 0, 1},     // 40 (3 x 48 KiB stages)
};
constexpr int kNumP = (int)(sizeof(kP) / sizeof(kP[0]));
int conv_dma_p_num_cfgs() { return kNumP; }

bool conv_dma_p_cfg_valid(const ConvParams& p, int c) {
    if (c < 0 || c >= kNumP) return false;
    if ((p.Cin % 32) != 0 || (p.Kpad % 32) != 0 || p.ks > 3 || p.up != 1) return false;
    if (p.x_bytes >= (1ull << 31) || p.w_bytes >= (1ull << 31) || p.y_bytes >= (1ull << 31)) return false;
    if ((p.Cout & 3) || (p.y_stride & 3) || (p.y_coff & 3) || (p.res && ((p.res_stride & 3) || (p.res_coff & 3)))) return false;
    if (p.res && p.out_f32) return false;
    const DmaPCfg& k = kP[c];
    if (k.BK == 64 && ((p.Cin % 64) != 0 || (p.Kpad % 64) != 0)) return false;
    if (p.x2_C > 0 && (p.ks != 1 || (p.x2_C % k.BK) != 0 || p.x2_bytes >= (1ull << 31))) return false;
    const int cpad = (p.Cout + 31) / 32 * 32;
    if (k.BN > 32 && k.BN >= 2 * cpad) return false;
    if (k.BN == 32 && p.Cout > 32) return false;
    if (k.wres) {
        const size_t sh = (size_t)k.NS * k.BM * k.BK * 2 + 1024 + (size_t)k.BN * p.Kpad * 2;
        if (sh > 160 * 1024) return false;
    }
    return true;
}
const char* conv_dma_p_kernel_name(int c) { return kP[c].name; }

template <int BM, int BN, int WGM, int WGN, int BK, int NS, bool HAS_RES, bool OUT_F32, bool WRES, bool PIPE, bool PP>
static hipError_t launch_p_var(const ConvParams& p, hipStream_t st) {
    const int mtiles = (p.M + BM - 1) / BM, ntiles = (p.Cout + BN - 1) / BN;
    const size_t sh = WRES ? ((size_t)NS * BM * BK * 2 + 1024 + (size_t)BN * p.Kpad * 2) : ((size_t)NS * (BM + BN) * BK * 2 + 1024);
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(WGM * WGN == 8 ? 2 : 4, (160 * 1024) / sh));
    int G = (256 * per_cu) / ntiles;
    if (G < 1) G = 1;
    if (G > mtiles) G = mtiles;
    auto kern = conv_dma_p_kernel<BM, BN, WGM, WGN, BK, NS, HAS_RES, OUT_F32, WRES, PIPE, PP>;
    static bool attr = false;
    if (!attr && (WRES || sh > 64 * 1024)) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(WRES ? 160 * 1024 : sh));
        if (e != hipSuccess) return e;
        attr = true;
    }
    hipLaunchKernelGGL(kern, dim3(G * ntiles), dim3(WGM * WGN * 64), sh, st, p, mtiles, ntiles, G);
    return hipGetLastError();
}
template <int BM, int BN, int WGM, int WGN, int BK, int NS, bool WRES = false, bool PIPE = false, bool PP = false>
static hipError_t launch_p_one(const ConvParams& p, hipStream_t st) {
    if (p.out_f32) return launch_p_var<BM, BN, WGM, WGN, BK, NS, false, true, WRES, PIPE, PP>(p, st);
    if (p.res) return launch_p_var<BM, BN, WGM, WGN, BK, NS, true, false, WRES, PIPE, PP>(p, st);
    return launch_p_var<BM, BN, WGM, WGN, BK, NS, false, false, WRES, PIPE, PP>(p, st);
}

hipError_t launch_conv_dma_p(const ConvParams& p, int c, hipStream_t st) {
    switch (c) {
        case 0: return launch_p_one<128, 32, 4, 1, 32, 4>(p, st);
        case 1: return launch_p_one<128, 64, 2, 2, 32, 4>(p, st);
        case 2: return launch_p_one<128, 128, 2, 2, 32, 4>(p, st);
        case 3: return launch_p_one<64, 64, 2, 2, 32, 4>(p, st);
        case 4: return launch_p_one<256, 64, 4, 2, 32, 4>(p, st);
        case 5: return launch_p_one<256, 128, 4, 2, 32, 4>(p, st);
        case 6: return launch_p_one<128, 128, 2, 2, 64, 3>(p, st);
        case 7: return launch_p_one<128, 64, 2, 2, 64, 3>(p, st);
        case 8: return launch_p_one<128, 256, 2, 4, 32, 4>(p, st);
        case 9: return launch_p_one<64, 128, 2, 2, 64, 3>(p, st);
        case 10: return launch_p_one<64, 64, 2, 2, 64, 4>(p, st);
        case 11: return launch_p_one<128, 64, 4, 2, 64, 4>(p, st);
        case 12: return launch_p_one<128, 64, 2, 2, 32, 4, true>(p, st);
        case 13: return launch_p_one<128, 64, 2, 2, 64, 4, true>(p, st);
        case 14: return launch_p_one<64, 64, 2, 2, 32, 4, true>(p, st);
        case 15: return launch_p_one<64, 64, 2, 2, 64, 4, true>(p, st);
        case 16: return launch_p_one<128, 128, 2, 2, 64, 3, true>(p, st);
        case 17: return launch_p_one<128, 128, 4, 2, 64, 4, true>(p, st);
        case 18: return launch_p_one<128, 32, 4, 1, 32, 4, true>(p, st);
        case 19: return launch_p_one<256, 64, 4, 2, 64, 3, true>(p, st);
        case 20: return launch_p_one<128, 256, 2, 4, 64, 3, true>(p, st);
        case 21: return launch_p_one<256, 128, 4, 2, 32, 4, false, true>(p, st);
        case 22: return launch_p_one<256, 64, 4, 2, 32, 4, false, true>(p, st);
        case 23: return launch_p_one<128, 256, 2, 4, 32, 4, false, true>(p, st);
        case 24: return launch_p_one<128, 128, 2, 2, 32, 4, false, true>(p, st);
        case 25: return launch_p_one<128, 64, 2, 2, 32, 4, false, true>(p, st);
        case 26: return launch_p_one<64, 64, 2, 2, 32, 4, false, true>(p, st);
        case 27: return launch_p_one<64, 64, 2, 2, 64, 4, false, true>(p, st);
        case 28: return launch_p_one<128, 64, 4, 2, 64, 4, false, true>(p, st);
        case 29: return launch_p_one<128, 128, 4, 2, 64, 4, true, true>(p, st);
        case 30: return launch_p_one<64, 64, 2, 2, 64, 4, true, true>(p, st);
        case 31: return launch_p_one<128, 64, 2, 2, 64, 4, true, true>(p, st);
        case 32: return launch_p_one<128, 128, 4, 2, 64, 4, false, true>(p, st);
        case 33: return launch_p_one<256, 128, 4, 2, 32, 4, false, false, true>(p, st);
        case 34: return launch_p_one<256, 64, 4, 2, 32, 4, false, false, true>(p, st);
        case 35: return launch_p_one<128, 256, 2, 4, 32, 4, false, false, true>(p, st);
        case 36: return launch_p_one<128, 128, 4, 2, 64, 4, false, false, true>(p, st);
        case 37: return launch_p_one<128, 64, 4, 2, 64, 4, false, false, true>(p, st);
        case 38: return launch_p_one<128, 128, 4, 2, 64, 4, true, false, true>(p, st);
        case 39: return launch_p_one<256, 64, 4, 2, 64, 3, true, false, true>(p, st);
        default: return launch_p_one<256, 128, 4, 2, 64, 3, false, false, true>(p, st);
    }
}

}  // namespace yp
