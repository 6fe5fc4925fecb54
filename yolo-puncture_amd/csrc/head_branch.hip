// v10Detect one-to-one head, round 3: the box branch (and the mask-coefficient branch of the seg head) evaluated ONLY where its result is
// used - at the top-k anchors of stage 1.
//
// The reference gathers box distances (and coefficients) for the max_det anchors its top-k keeps and throws the other 8100 of 8400 away
// ([U] v10postprocess: topk on the class scores, then gather; run inside `.predict`, reference yolo_seg/app.py:91). The branch
// that produces them is three small convolutions per level - cv.0 3x3 (Cin -> CMID) + SiLU, cv.1 3x3 (CMID -> CMID) + SiLU, cv.2 1x1
// (CMID -> COUT) - i.e. a 5x5 receptive field on the level's feature map. Dense, that is 73 GFLOP per 32 frames (173 us of the step over
// nine launches). The values computed here are the same numbers: the same bf16 inputs, the same bf16 weights, fp32 accumulation, the same
// rounding points (bias + SiLU + bf16 after the first two convolutions); only the summation order inside an fp32 sum differs - as it does
// between any two tile configurations of the dense kernels.
//
// A winner's cv.1 reads cv.0 at its 3x3 neighbourhood. Neighbouring winners share those positions and a level has only H*W of them, so the
// stage-1 kernel (head.hip) lists the DISTINCT in-frame positions the winners' neighbourhoods cover - per level, all images in one list:
//   head_pos_kernel  cv.0 once per listed position, 64 positions per tile: MFMA pixel rows = positions, so every fragment is full; the
//                    positions' 3x3 input patches arrive per 32-channel chunk by LDS-DMA as [9 taps][64 positions][32 ch] (zero outside
//                    the frame = the padding), weight fragments go straight from L2 to registers one chunk ahead; bias + SiLU + bf16 into a
//                    position-addressed map t0 [level][image][pixel][CMID] (only listed positions are written, only those are read).
//   head_win_kernel  16 winners of one image and level per workgroup: cv.1 at the winners' own pixels gathers its 9 taps from t0 (zero
//                    outside the frame), bias + SiLU + bf16 -> T1 [16][CMID] in LDS, cv.2 + bias -> fp32 row out[b][rank][COUT].
// Work is min(9 * winners, H*W) positions per level and image: never more than the dense branch.
// (First form of this file: ONE kernel, cv.0 at 9 positions per winner from a 5x5 patch. 2700 evaluations per image however the winners
//  lie; on the bench's synthetic network, whose 300 winners all sit on the 20x20 level, that is 2700 where 400 distinct positions exist,
//  and each 16-winner workgroup re-read the level's 590 KB of cv.0 weights: 593 MB through the CUs' L2 ports per launch, 100.9 us, at the
//  L2 -> CU ingest wall of ~30 B/clk/CU whatever the prefetch depth. This form: 35.6 + 16.1 us, +3.6 us in stage 1 for the lists.)
#include "common.h"
#include <cstdlib>
#include <type_traits>

namespace yp {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) bf16x8* gfrag;       // weight fragments: GLOBAL loads (a generic pointer would make them flat_load,
                                                                      // which also counts in lgkmcnt and drags the LDS waits along)

__device__ __forceinline__ void hb_write8(unsigned char* dst, uint2 v) {
    asm volatile("ds_write_b64 %0, %1" ::"v"((unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)dst), "v"(*(const unsigned long long*)&v) : "memory");
}

// phase timestamps (100 MHz) of workgroup 0's first tile in the last head_pos_kernel launch (yp_debug_head_branch_clocks)
__device__ unsigned long long g_hb_clk[8];
#define HB_STAMP(i) do { if (blockIdx.x == 0 && g == 0 && threadIdx.x == 0) g_hb_clk[i] = wall_clock64(); } while (0)

template <int CMID>
__global__ __launch_bounds__(256) void head_pos_kernel(const HeadBranchParams p) {
    constexpr int NCF = CMID / 16, NPG = 4 / NCF, NT = 4 / NPG;        // wave = (channel fragment cf, position group pg); NT 16-position tiles each
    static_assert(NCF * NPG == 4, "4 waves");
    constexpr unsigned OOB = 0x80000000u;
    constexpr int PSLOT = 9 * 64 * 64;                                // one chunk plane: [9 taps][64 positions][32 ch] bf16
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char* const Ps = smem;                                   // 3 slots: chunk c in slot c % 3, two planes in flight
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fc = lane >> 4;
    const int cf = wave % NCF, pg = wave / NCF;
    const int dpc = (lane & 3) ^ (((lane >> 4) & 1) << 1);
    // tiles of the three levels, the level with the most channels first (its tiles are the long ones: a workgroup that starts one last
    // would finish last)
    const int c0 = min(p.pcount[0], p.plist_cap[0]), c1 = min(p.pcount[1], p.plist_cap[1]), c2 = min(p.pcount[2], p.plist_cap[2]);
    const int n2 = (c2 + 63) >> 6, n1 = (c1 + 63) >> 6, n0 = (c0 + 63) >> 6;
    for (int g = blockIdx.x; g < n0 + n1 + n2; g += gridDim.x) {
        const int l = g < n2 ? 2 : g < n2 + n1 ? 1 : 0;
        const int t = g - (l == 2 ? 0 : l == 1 ? n2 : n2 + n1);
        const int cnt = l == 2 ? c2 : l == 1 ? c1 : c0;
        const int nv = min(64, cnt - t * 64);
        HB_STAMP(0);
        // (kernel-argument arrays indexed by a run-time level are re-read with vector loads + a full wait per use: select scalars once)
        const int H = l == 2 ? p.H[2] : l == 1 ? p.H[1] : p.H[0], W = l == 2 ? p.W[2] : l == 1 ? p.W[1] : p.W[0];
        const int Cin = l == 2 ? p.Cin[2] : l == 1 ? p.Cin[1] : p.Cin[0], nchunk = Cin >> 5;
        const int xs = l == 2 ? p.x_stride[2] : l == 1 ? p.x_stride[1] : p.x_stride[0], xc = l == 2 ? p.x_coff[2] : l == 1 ? p.x_coff[1] : p.x_coff[0];
        const void* const xp = l == 2 ? p.x[2] : l == 1 ? p.x[1] : p.x[0];
        const size_t xb = l == 2 ? p.x_bytes[2] : l == 1 ? p.x_bytes[1] : p.x_bytes[0];
        const __bf16* const w0 = (const __bf16*)(l == 2 ? p.w0[2] : l == 1 ? p.w0[1] : p.w0[0]);
        const float* const b0 = l == 2 ? p.b0[2] : l == 1 ? p.b0[1] : p.b0[0];
        const int K0 = 9 * Cin;
        const int* const pl = p.plist + (l == 2 ? p.plist_off[2] : l == 1 ? p.plist_off[1] : p.plist_off[0]) + t * 64;
        const size_t t0o = l == 2 ? p.t0_off[2] : l == 1 ? p.t0_off[1] : p.t0_off[0];

        // this lane's position in its DMA role: row lane >> 2 of the wave's 16-position group (surplus rows repeat the tile's last position)
        const int ent_d = pl[min(wave * 16 + (lane >> 2), nv - 1)];
        const int b_d = ent_d >> 20, loc_d = ent_d & 0xFFFFF;
        const int y_d = loc_d / W, x_d = loc_d - y_d * W;
        if (ent_d >= 0) HB_STAMP(1);
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)xp, 0, (int)xb, 0x00020000);
        // 36 one-KB pieces per plane (tap j, 16-position group): wave w issues the 9 taps of group w - every wave 9 pieces in every call,
        // also for a chunk past the last one (zero fill into the free slot), so that one counted wait fits every iteration
        auto issue_plane = [&](int c, int slot) {
            const bool live = c < nchunk;
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                const int yy = y_d + j / 3 - 1, xx = x_d + j % 3 - 1;
                const bool ok = live && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
                const unsigned voff = ok ? (unsigned)((((b_d * H + yy) * W + xx) * xs + xc + c * 32 + dpc * 8) * 2) : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_void*)(Ps + slot * PSLOT + (j * 4 + wave) * 1024), 16, voff, 0, 0, 0);
            }
        };
        auto load_w0 = [&](int c, bf16x8 (&dst)[9]) {
            const int cc = min(c, nchunk - 1);
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) dst[tp] = *(gfrag)(w0 + (size_t)(cf * 16 + fr) * K0 + tp * Cin + cc * 32 + fc * 8);
        };
        f32x4 acc[NT];
        {
            const float4 bb = *(const float4*)(b0 + cf * 16 + fc * 4);
#pragma unroll
            for (int i = 0; i < NT; ++i) acc[i] = f32x4{bb.x, bb.y, bb.z, bb.w};
        }
        bf16x8 wa[9], wb[9];
        __builtin_amdgcn_s_barrier();                                 // (every wave is done with the previous tile's last plane)
        load_w0(0, wa);
        __builtin_amdgcn_sched_barrier(0);
        issue_plane(0, 0);
        issue_plane(1, 1);
        __builtin_amdgcn_sched_barrier(0);
        for (int c = 0; c < nchunk; ++c) {
            __builtin_amdgcn_s_waitcnt(9 | (7 << 4) | (0xF << 8));    // vmcnt(9): all but the youngest plane - plane c and the fragments of chunk c are there
            __builtin_amdgcn_s_barrier();
            if (c == 0) HB_STAMP(2);
            load_w0(c + 1, wb);
            __builtin_amdgcn_sched_barrier(0);                        // fragments strictly before the plane (in-order counter)
            issue_plane(c + 2, (c + 2) % 3);
            __builtin_amdgcn_sched_barrier(0);
            const unsigned char* ps = Ps + (c % 3) * PSLOT;
#pragma unroll
            for (int tr = 0; tr < 3; ++tr) {                          // three taps at a time: 3 * NT fragments in registers
                bf16x8 xr[3][NT];
#pragma unroll
                for (int q = 0; q < 3; ++q)
#pragma unroll
                    for (int i = 0; i < NT; ++i) xr[q][i] = *(const bf16x8*)(ps + ((tr * 3 + q) * 4 + pg + i * NPG) * 1024 + swz64((unsigned)(fr * 64 + fc * 16)));
#pragma unroll
                for (int q = 0; q < 3; ++q)
#pragma unroll
                    for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[tr * 3 + q], xr[q][i], acc[i], 0, 0, 0);
            }
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) wa[tp] = wb[tp];
        }
        HB_STAMP(3);
        __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (0xF << 8));        // (the zero-fill pieces past the last chunk, before the next tile reuses the slots)
        // bias (in the accumulator) + SiLU + bf16 -> t0[level][image][pixel][cf * 16 + fc * 4 ..]
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int idx = (pg + i * NPG) * 16 + fr;
            const int ent = pl[min(idx, nv - 1)];
            float v[4] = {acc[i][0], acc[i][1], acc[i][2], acc[i][3]};
            if (p.act0 == ACT_SILU) silu4_packed(v);
            __attribute__((aligned(8))) __bf16 o[4] = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
            if (idx < nv)
                *(uint2*)((__bf16*)p.t0 + t0o + ((size_t)(ent >> 20) * (H * W) + (ent & 0xFFFFF)) * CMID + cf * 16 + fc * 4) = *(const uint2*)o;
        }
        HB_STAMP(4);
        if (blockIdx.x == 0 && g == 0 && tid == 0) { g_hb_clk[5] = ((unsigned long long)l << 32) | (unsigned)nchunk; g_hb_clk[6] = (unsigned long long)(n0 + n1 + n2); g_hb_clk[7] = (unsigned long long)(c0 + c1 + c2); }
    }
}

template <int CMID, int COUT>
__global__ __launch_bounds__(256) void head_win_kernel(const HeadBranchParams p) {
    constexpr int NCF = CMID / 16, NKC1 = CMID / 32, NCF2 = COUT / 16;
    static_assert(NCF <= 4 && NCF2 <= 4, "4 waves");
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char* const T0 = smem;                                   // [NKC1][9 taps * 16 winners][64 B]
    unsigned char* const T1 = smem + NKC1 * 9 * 1024;                 // [NKC1][16 winners][64 B]
    const int l = blockIdx.y, b = blockIdx.z, grp = blockIdx.x;
    const int cnt = p.wcount[b * 3 + l];
    if (grp * 16 >= cnt) return;                                      // (workgroup-uniform)
    const int nv = min(16, cnt - grp * 16);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fc = lane >> 4;
    const int H = p.H[l], W = p.W[l];
    const int* const wl = p.wlist + (size_t)(b * 3 + l) * p.maxk + grp * 16;
    const int ent_f = wl[min(fr, nv - 1)], ent_d = wl[min(lane >> 2, nv - 1)];
    const int rank_f = ent_f & 511, loc_d = ent_d >> 9;
    const int y_d = loc_d / W, x_d = loc_d - y_d * W;
    const int dpc = (lane & 3) ^ (((lane >> 4) & 1) << 1);
    // the 9 taps of cv.1 = the first convolution's outputs around the winner: pieces (channel chunk kc, tap t) of 16 rows
    {
        const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc(p.t0, 0, (int)p.t0_bytes, 0x00020000);
        const size_t base = p.t0_off[l] + (size_t)b * (H * W) * CMID;
        for (int q = wave; q < NKC1 * 9; q += 4) {
            const int kc = q / 9, t = q - kc * 9;
            const int yy = y_d + t / 3 - 1, xx = x_d + t % 3 - 1;
            const bool ok = (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
            const unsigned voff = ok ? (unsigned)((base + (size_t)(yy * W + xx) * CMID + kc * 32 + dpc * 8) * 2) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(trs, (lds_void*)(T0 + q * 1024), 16, voff, 0, 0, 0);
        }
    }
    if (wave < NCF) {
        const int cf = wave;
        const __bf16* const w1 = (const __bf16*)p.w1[l];
        const int K1 = p.Kpad1[l];
        const float4 bb = *(const float4*)(p.b1[l] + cf * 16 + fc * 4);
        f32x4 a1 = f32x4{bb.x, bb.y, bb.z, bb.w};
        bf16x8 w1f[9][NKC1];
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int kc = 0; kc < NKC1; ++kc) w1f[t][kc] = *(gfrag)(w1 + (size_t)(cf * 16 + fr) * K1 + t * CMID + kc * 32 + fc * 8);
        __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (0xF << 8));
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int kc = 0; kc < NKC1; ++kc) {
                const bf16x8 xf = *(const bf16x8*)(T0 + (kc * 9 + t) * 1024 + swz64((unsigned)(fr * 64 + fc * 16)));
                a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1f[t][kc], xf, a1, 0, 0, 0);
            }
        float v[4] = {a1[0], a1[1], a1[2], a1[3]};
        if (p.act1 == ACT_SILU) silu4_packed(v);
        __attribute__((aligned(8))) __bf16 o[4] = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
        const int c8 = (2 * (cf & 1) + (fc >> 1)) & 3;
        hb_write8(T1 + (size_t)(cf >> 1) * 16 * 64 + swz64((unsigned)(fr * 64 + c8 * 16)) + (fc & 1) * 8, *(const uint2*)o);
    } else {
        __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (0xF << 8));
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wave < NCF2) {
        const __bf16* const w2 = (const __bf16*)p.w2[l];
        const int K2 = p.Kpad2[l];
        const float4 bb = *(const float4*)(p.b2[l] + wave * 16 + fc * 4);
        f32x4 a2 = f32x4{bb.x, bb.y, bb.z, bb.w};
#pragma unroll
        for (int kc = 0; kc < NKC1; ++kc) {
            const bf16x8 wf = *(gfrag)(w2 + (size_t)(wave * 16 + fr) * K2 + kc * 32 + fc * 8);
            const bf16x8 xf = *(const bf16x8*)(T1 + (size_t)kc * 16 * 64 + swz64((unsigned)(fr * 64 + fc * 16)));
            a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf, a2, 0, 0, 0);
        }
        if (fr < nv) *(float4*)(p.out + ((size_t)b * p.max_det + rank_f) * COUT + wave * 16 + fc * 4) = make_float4(a2[0], a2[1], a2[2], a2[3]);
    }
}

hipError_t head_branch_read_clocks(unsigned long long* out8) { return hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_hb_clk), 8 * sizeof(unsigned long long)); }

bool head_branch_valid(const HeadBranchParams& p) {
    if (!((p.cmid == 64 && p.cout == 64) || (p.cmid == 32 && p.cout == 32))) return false;
    for (int l = 0; l < 3; ++l) {
        if ((p.Cin[l] % 32) != 0 || p.Cin[l] < 32 || (p.x_stride[l] & 7) || (p.x_coff[l] & 7) || p.x_bytes[l] >= (1ull << 31)) return false;
        if (p.Kpad0[l] != 9 * p.Cin[l] || p.Kpad1[l] != 9 * p.cmid || p.Kpad2[l] != (p.cmid + 31) / 32 * 32) return false;   // no padded taps
    }
    if (p.t0_bytes >= (1ull << 31) || p.B >= 2048 || p.H[0] * p.W[0] >= (1 << 20)) return false;   // 32-bit gather offsets, image << 20 | pixel (sizes are plan-time facts: checked before the workspace exists)
    return p.maxk >= p.max_det && p.max_det <= 512;
}

hipError_t launch_head_branch(const HeadBranchParams& p, hipStream_t st) {
    if (!p.plist || !p.pcount || !p.t0) return hipErrorInvalidValue;
    const dim3 grid((unsigned)((p.max_det + 15) / 16), 3, (unsigned)p.B);
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)head_pos_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 9 * 64 * 64);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)head_pos_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 9 * 64 * 64);
        if (e != hipSuccess) return e;
        attr = true;
    }
    if (p.cmid == 64) {
        hipLaunchKernelGGL((head_pos_kernel<64>), dim3((unsigned)p.pos_grid), dim3(256), 3 * 9 * 64 * 64, st, p);
        hipLaunchKernelGGL((head_win_kernel<64, 64>), grid, dim3(256), 2 * 10 * 1024, st, p);
    } else {
        hipLaunchKernelGGL((head_pos_kernel<32>), dim3((unsigned)p.pos_grid), dim3(256), 3 * 9 * 64 * 64, st, p);
        hipLaunchKernelGGL((head_win_kernel<32, 32>), grid, dim3(256), 10 * 1024, st, p);
    }
    return hipGetLastError();
}

}  // namespace yp
