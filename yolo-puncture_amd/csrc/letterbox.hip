// LetterBox on the device (SURVEY.md §8f-1, A.5 step 2 [U]; the step right before `.predict`'s network, reference call
// sites yolo_seg/app.py:86-91): aspect-preserving 8-bit bilinear resize in OpenCV's INTER_LINEAR fixed-point form, centred
// in a frame of constant 114. Integer arithmetic exactly as the oracle restates it (oracle/postprocess_oracle.py:
// resize_bilinear_u8_cv2): 11-bit coefficients a = rint(f * 2048), horizontal pass in int32 scaled by 2048, vertical pass
// ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2. The source coordinate is the float32 cast of the float64
// (d + 0.5) * scale - 0.5, computed here with explicitly rounded operations so that no FMA contraction can change a bit.
#include "common.h"

namespace yp {

struct AxisCoef { int s0, s1, a0, a1; };

__device__ __forceinline__ AxisCoef axis_coef(int d, int n_src, double scale) {
    const double fd = __dsub_rn(__dmul_rn(__dadd_rn((double)d, 0.5), scale), 0.5);
    float f = (float)fd;
    int s = (int)floorf(f);
    f = __fsub_rn(f, (float)s);
    if (s < 0) { s = 0; f = 0.f; }
    if (s >= n_src - 1) { s = n_src - 1; f = 0.f; }
    AxisCoef c;
    c.s0 = s;
    c.s1 = min(s + 1, n_src - 1);
    c.a1 = __float2int_rn(__fmul_rn(f, 2048.f));
    c.a0 = __float2int_rn(__fmul_rn(__fsub_rn(1.0f, f), 2048.f));
    return c;
}

__global__ __launch_bounds__(256) void letterbox_kernel(const uint8_t* __restrict__ src, int h0, int w0, uint8_t* __restrict__ dst,
                                                        int out_h, int out_w, int new_h, int new_w, int top, int left, int pad,
                                                        double sx, double sy, int identity) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= out_w || y >= out_h) return;
    uint8_t* o = dst + ((size_t)y * out_w + x) * 3;
    const int rx = x - left, ry = y - top;
    if ((unsigned)rx >= (unsigned)new_w || (unsigned)ry >= (unsigned)new_h) {
        o[0] = o[1] = o[2] = (uint8_t)pad;
        return;
    }
    if (identity) {                                  // cv2.resize is skipped when the shape already matches
        const uint8_t* s = src + ((size_t)ry * w0 + rx) * 3;
        o[0] = s[0]; o[1] = s[1]; o[2] = s[2];
        return;
    }
    const AxisCoef cx = axis_coef(rx, w0, sx), cy = axis_coef(ry, h0, sy);
    const uint8_t* r0 = src + (size_t)cy.s0 * w0 * 3;
    const uint8_t* r1 = src + (size_t)cy.s1 * w0 * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int h0v = (int)r0[cx.s0 * 3 + c] * cx.a0 + (int)r0[cx.s1 * 3 + c] * cx.a1;
        const int h1v = (int)r1[cx.s0 * 3 + c] * cx.a0 + (int)r1[cx.s1 * 3 + c] * cx.a1;
        int v = (((cy.a0 * (h0v >> 4)) >> 16) + ((cy.a1 * (h1v >> 4)) >> 16) + 2) >> 2;
        v = v < 0 ? 0 : (v > 255 ? 255 : v);
        o[c] = (uint8_t)v;
    }
}

hipError_t launch_letterbox(const uint8_t* src, int h0, int w0, uint8_t* dst, int out_h, int out_w, int new_h, int new_w, int top,
                            int left, int pad, hipStream_t st) {
    const double sx = (double)w0 / (double)new_w, sy = (double)h0 / (double)new_h;
    const int identity = (h0 == new_h && w0 == new_w) ? 1 : 0;
    hipLaunchKernelGGL(letterbox_kernel, dim3((out_w + 63) / 64, (out_h + 3) / 4), dim3(256), 0, st, src, h0, w0, dst, out_h, out_w,
                       new_h, new_w, top, left, pad, sx, sy, identity);
    return hipGetLastError();
}

}  // namespace yp
