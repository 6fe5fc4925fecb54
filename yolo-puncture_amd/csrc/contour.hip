// Masks.xy + the shaft-length rectangle on the GPU: what the reference's video loop does with every frame's best mask,
//     coord_xy = results[0].masks.xy[best]                 (yolo_seg/app.py:101; [U] masks2segments = cv2.findContours RETR_EXTERNAL /
//                                                            CHAIN_APPROX_SIMPLE, then strategy "all" (8.3.x: every contour, concatenated)
//                                                            or "largest" (8.0-8.2: the contour with the most points))
//     rect_len, ratio = get_coord_min_rect_len(coord_xy)    (yolo_seg/app.py:102-103; yolo_seg/utils/mask_tools.py:12-22 = cv2.minAreaRect)
// without the full-resolution mask ever leaving HBM. A pre-pass finds the bounding boxes (32 workgroups per mask), then one workgroup per mask:
//   1. bit image of the box in LDS (one zero word / row around it; masks are zero outside their detection box)
//   2. candidate starts = set pixels whose W, NW, N, NE neighbours are clear (word-parallel bit logic). The raster-first pixel of every
//      8-connected blob is one; other candidates are later local tops of a blob's outer border or local tops on hole borders
//   3. Moore traces (3x3 neighbourhood = three 64-bit LDS windows, next direction by bit rotation + ctz; Jacob's stopping criterion),
//      CHAIN_APPROX_SIMPLE points written while walking into per-candidate lists inside the caller's point buffer:
//      a. up to 1024 candidates: every candidate is a CHECKPOINT of its border, every lane walks only to the next one (trace_segment);
//         borders = cycles of the `next` pointers, led by their raster-first candidate; a cycle whose raster-first pixel is not that
//         candidate (a hole border) is dropped; point counts from the segments + the joints
//      b. otherwise (noise masks): one lane per candidate walks its whole border; a walk that meets a pixel earlier in raster order than
//         its start is not a blob's outer border from its first pixel and is dropped - no connected-component labelling either way.
//      The survivors are the outer borders of the 8-connected blobs.
//   3c. RETR_EXTERNAL: with two or more outer borders, one that lies INSIDE another (a blob in a hole of another blob) is not external.
//      Decided by crossing parity: the walkers go round once more and count, for every border start q, the unit moves that cross the
//      rightward ray from q (half-open rule on the pixel-centre polygon; a start pixel never lies on another blob's polygon); q is nested
//      iff some other outer border crosses its ray an odd number of times. One blob (the needle's usual mask) skips all of this.
//   4. order: contours are listed bottom-up (descending raster order of their start pixels - the order cv2 hands them back in);
//      "largest" keeps the one with the most points (first in that order on a tie), "all" keeps every one. Lists are assembled by parallel
//      copies (a serial re-trace only when a list overflowed), each turned round from the start pixel (the walkers go clockwise, cv2's
//      outer borders run the other way: down first); per-column min/max of all points -> Andrew's monotone chain on at most 2 points per
//      column (exact integer cross products, both chains at once) -> rotating calipers over the hull edges in float64.
// The same definitions, stated on the host: hostops.external_contours / mask_polygon (connected components + hole filling); a third,
// independent statement (Suzuki & Abe's raster labelling as cv2 runs it) checks both in tests/.
#include "common.h"
#include <cstdio>

namespace yp {

constexpr int CT_THREADS = 1024;
constexpr int CT_MAXCAND = 4096;            // candidate starts (cand[] behind the bit image); a multiple of CT_THREADS: the one-lane path keeps its
                                            // candidates' results in registers, CT_MAXCAND / CT_THREADS per thread
constexpr int CT_NLMAX = 64;                // outer borders that take part in the nesting test / the "all" list (more: the host path)
constexpr int CT_BITMAP_BYTES = 126 * 1024;        // a whole 1280x720 frame fits (722 rows x 43 words)
constexpr int CT_MAXCOL = 2048;            // hull column tables

struct ContourParams {
    const uint8_t* masks;      // [n][H][W], non-zero = set
    int n, H, W;
    int max_pts;
    int32_t* pts;              // [n][max_pts][2] (x, y) of the winning contour's run end points
    int32_t* count;            // [n] number of points (0: empty mask ; -1: bounding box too large for the LDS image ; -2: more than max_pts points / candidates / contours)
    int strategy;              // 0 = "largest", 1 = "all"
    int32_t* parts;            // [n][parts_cap] or null: [0] = number of external contours in the list, [1..] = their point counts in list order
    int parts_cap;
    double* rect;              // [n][2] (long side, short side) of the minimum-area rectangle of those points, or null
    int boxg;                  // workgroups per mask of the bounding-box pre-pass (their partial boxes sit at the head of the mask's `pts`)
};

// clockwise from east, as hostops._DIRS: (dy,dx)
// (dy, dx) = {0,1,1,1,0,-1,-1,-1}, {1,1,0,-1,-1,-1,0,1}, each + 1 in two bits per direction: decoded in registers - a `__constant__` table indexed
// per lane is a vector memory load, two of them on the serial path of every trace step (measured: 643 -> see DESIGN us per 720p mask)
__device__ __forceinline__ int c_dy(int d) { return (int)((0x01a9u >> (2 * d)) & 3u) - 1; }
__device__ __forceinline__ int c_dx(int d) { return (int)((0x901au >> (2 * d)) & 3u) - 1; }

// phase stamps of mask 0's workgroup (s_memtime, 100 MHz) + [8] candidates, [9] points of the winner, [10] box width, [11] box height
__device__ unsigned long long g_ct_clk[12];
#define CT_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_ct_clk[i] = wall_clock64(); } while (0)

struct Bitmap {
    const unsigned* w;   // LDS
    int pitch;           // words per row
    // 8-neighbour ring of the pixel in column x whose row ABOVE starts at word `rb` (row y of the image lives at stored row y + 1, column x
    // at bit x + 32): bits E, SE, S, SW, W, NW, N, NE. Three 64-bit windows, one shift each; the row below arrives mirrored through a
    // packed 3-bit reversal table.
    __device__ __forceinline__ unsigned ring(int rb, int x) const {
        const int pos = x + 31;                              // bit position of x-1
        const unsigned* r0 = w + rb + (pos >> 5);
        const int sh = pos & 31;
        const unsigned long long u = ((unsigned long long)r0[1] << 32) | r0[0];
        const unsigned long long m = ((unsigned long long)r0[pitch + 1] << 32) | r0[pitch];
        const unsigned long long d = ((unsigned long long)r0[2 * pitch + 1] << 32) | r0[2 * pitch];
        const unsigned uu = (unsigned)(u >> sh) & 7u, mm = (unsigned)(m >> sh) & 7u, dd = (unsigned)(d >> sh) & 7u;
        return (mm >> 2) | ((0x73516240u >> (4 * dd)) & 7u) << 1 | (mm & 1u) << 4 | uu << 5;
    }
};

constexpr int CT_SLOT_PTS = 1024;           // points per emission slot (and of the list's head the winner is assembled in), when max_pts allows
constexpr int CT_MAXSLOTS = 255;            // candidate k < slots emits into slot k of the mask's own `pts` region (regions 1 .. slots behind the head)
constexpr int CT_BOXG = 32;                 // workgroups per mask of the bounding-box pre-pass (fewer when the point list is shorter than 64)

// One Moore trace from (sy,sx), `max_steps` steps at most.
// EMIT = false: returns the number of CHAIN_APPROX_SIMPLE points (0 = dropped: met an earlier pixel, or longer than the step bound) and
//   whether the start point is one of them (`start_kept`: it is iff the last move differs from the first; _compress lists it first).
// EMIT = true: additionally writes the kept points except the start point to out[0 .. cap) in trace order, bounding-box origin added
//   (`*stored` = how many were kept; more than cap = the list overflowed). The counting pass gives every candidate with a slot its own
//   list this way, so the winner's points are already in memory when it is known - tracing it a second time cost as much as the whole
//   counting pass.
template <bool EMIT>
__device__ int moore_trace(const Bitmap& bm, int bw, int sy, int sx, int max_steps, int32_t* out, int cap, int ox, int oy, int* start_kept, int* stored) {
    // The step is the serial path of the whole kernel (one lane walks the winning border alone), so it carries as little as possible:
    // running row base and linear index instead of multiplications, the first step peeled so that the loop has no "first time" tests,
    // the pending point is always the current pixel.
    const int start_lin = sy * bw + sx;
    int cy = sy, cx = sx, rb = sy * bm.pitch, lin = start_lin;
    *start_kept = 0;
    if (EMIT) *stored = 0;
    unsigned nbm = bm.ring(rb, cx);
    if (nbm == 0) {                               // isolated pixel
        *start_kept = 1;
        return 1;
    }
    // first move: "arrived" from the north-west side, the search starts at north (d = 6)
    int nd = (6 + __builtin_ctz(((nbm >> 6) | (nbm << 2)) & 0xffu)) & 7;
    const int start_d = nd, first_move = nd;
    int prev_move = nd, npts = 2, nkeep = 0;
    {
        const int dy = c_dy(nd), dx = c_dx(nd);
        cy += dy; cx += dx; rb += dy * bm.pitch; lin += dy * bw + dx;
    }
    if (lin < start_lin) return 0;
    int d = (nd + 6 - (nd & 1)) & 7;              // restart at the background pixel examined last
    for (int step = 1; step < max_steps; ++step) {
        nbm = bm.ring(rb, cx);                    // (never empty: the pixel we came from is a neighbour)
        nd = (d + __builtin_ctz(((nbm >> d) | (nbm << (8 - d))) & 0xffu)) & 7;
        if (lin == start_lin && nd == start_d) {
            // closed: pts[:-1] drops the repeated start, n = npts - 1 points, moves m_0..m_{n-1} (the last one returned to the start)
            const int n = npts - 1;
            if (n <= 2) {                         // _compress keeps everything: the start, then (n == 2) the pixel the first move leads to
                *start_kept = 1;
                if (EMIT && n == 2) {
                    if (cap > 0) { out[0] = sx + c_dx(first_move) + ox; out[1] = sy + c_dy(first_move) + oy; }
                    *stored = 1;
                }
                return n;
            }
            if (EMIT) *stored = nkeep;
            // point 0 is kept iff the last move differs from the first one; the pending point (the start, reached by prev_move) is point 0
            if (prev_move != first_move) {
                *start_kept = 1;
                return nkeep + 1;
            }
            return nkeep > 0 ? nkeep : 1;         // (all moves equal cannot happen on a closed trace; _compress would keep pts[0])
        }
        if (nd != prev_move) {                    // the pixel we are leaving had a different move before it: kept
            if (EMIT && nkeep < cap) { out[2 * nkeep] = cx + ox; out[2 * nkeep + 1] = cy + oy; }
            ++nkeep;
        }
        prev_move = nd;
        const int dy = c_dy(nd), dx = c_dx(nd);
        cy += dy; cx += dx; rb += dy * bm.pitch; lin += dy * bw + dx;
        ++npts;
        if (lin < start_lin) return 0;            // not the raster-first pixel of its blob's outer border
        d = (nd + 6 - (nd & 1)) & 7;
    }
    return 0;
}

// One SEGMENT of a border: from candidate start (sy,sx) up to the next candidate start on the same border (possibly itself).
// Every candidate (local top: W, NW, N, NE clear) is a natural checkpoint of the borders it lies on: a walker that stands on such a pixel
// and is about to take the move a fresh trace would take there (first set neighbour clockwise from north) is, from then on, that fresh
// trace - the next state of a Moore walker depends on (pixel, move) only. So instead of one lane walking a whole border while the
// later local tops of the same border re-walk most of it in vain, every candidate's lane walks only to the next checkpoint; the borders
// are then the cycles of the `next` pointers, and CHAIN_APPROX_SIMPLE is decided inside a segment as before and at the joints from
// (last move of one segment, first move of the next).
struct SegInfo { int next_lin, nkeep, nmoves, first_move, last_move, min_lin; };
template <bool EMIT>
__device__ void trace_segment(const Bitmap& bm, const unsigned* cbm, int bw, int sy, int sx, int max_steps, int32_t* out, int cap, int ox, int oy, SegInfo& si) {
    const int start_lin = sy * bw + sx;
    int cy = sy, cx = sx, rb = sy * bm.pitch, lin = start_lin;
    si.next_lin = -1; si.nkeep = 0; si.nmoves = 0; si.first_move = 0; si.last_move = 0; si.min_lin = start_lin;
    int mn = start_lin;
    unsigned nbm = bm.ring(rb, cx);
    if (nbm == 0) { si.next_lin = start_lin; return; }          // isolated pixel: a border of its own, no moves
    int nd = (6 + __builtin_ctz(((nbm >> 6) | (nbm << 2)) & 0xffu)) & 7;
    si.first_move = nd;
    int prev_move = nd, nkeep = 0, nmoves = 1;
    {
        const int dy = c_dy(nd), dx = c_dx(nd);
        cy += dy; cx += dx; rb += dy * bm.pitch; lin += dy * bw + dx;
    }
    int d = (nd + 6 - (nd & 1)) & 7;
    for (int step = 1; step < max_steps; ++step) {
        nbm = bm.ring(rb, cx);
        nd = (d + __builtin_ctz(((nbm >> d) | (nbm << (8 - d))) & 0xffu)) & 7;
        mn = min(mn, lin);
        const unsigned cw = cbm[rb + bm.pitch + 1 + (cx >> 5)];
        if (((cw >> (cx & 31)) & 1u) && nd == ((6 + __builtin_ctz(((nbm >> 6) | (nbm << 2)) & 0xffu)) & 7)) {
            si.next_lin = lin; si.nkeep = nkeep; si.nmoves = nmoves; si.last_move = prev_move; si.min_lin = mn;
            return;
        }
        if (nd != prev_move) {
            if (EMIT && nkeep < cap) { out[2 * nkeep] = cx + ox; out[2 * nkeep + 1] = cy + oy; }
            ++nkeep;
        }
        prev_move = nd;
        const int dy = c_dy(nd), dx = c_dx(nd);
        cy += dy; cx += dx; rb += dy * bm.pitch; lin += dy * bw + dx;
        ++nmoves;
        d = (nd + 6 - (nd & 1)) & 7;
    }
}

// Crossing parity of a border (SEG: of one segment of it) against up to 64 query pixels: bit q of the result toggles for every unit move
// (y1,x1) -> (y2,x2) of the walk that crosses the rightward ray from query q under the half-open rule of the crossing-number test -
// (y1 > qy) != (y2 > qy), and the end point on row qy lies right of qx (moves are unit steps, so the other end point is on row qy + 1).
// Summed over a closed border the bit says whether q lies inside the polygon through the border's pixel centres; pieces walked twice
// (one-pixel-wide parts) cancel. Same walk as trace_segment / moore_trace.
template <bool SEG>
__device__ unsigned long long walk_parity(const Bitmap& bm, const unsigned* cbm, int bw, int sy, int sx, int max_steps, const int* qy, const int* qx, int nq,
                                          const unsigned long long* rowq = nullptr) {
    const int start_lin = sy * bw + sx;
    int cy = sy, cx = sx, rb = sy * bm.pitch, lin = start_lin;
    unsigned long long tog = 0ull;
    unsigned nbm = bm.ring(rb, cx);
    if (nbm == 0) return 0ull;
    int nd = (6 + __builtin_ctz(((nbm >> 6) | (nbm << 2)) & 0xffu)) & 7;
    const int start_d = nd;
    auto cross = [&](int y1, int x1, int y2, int x2) {
        if (y1 == y2) return;
        const int ylo = min(y1, y2), xa = (y1 < y2) ? x1 : x2;      // the end point on the upper row (the row a query must be on)
        if (rowq) {
            // rowq[y] = the queries on row y as a bit set: one LDS read per move instead of two per (move, query) - with 40 borders in a
            // mask that loop was 0.85 of the 0.9 ms the kernel took
            unsigned long long m = rowq[ylo];
            while (m) {
                const int q = __builtin_ctzll(m);
                m &= m - 1ull;
                if (xa > qx[q]) tog ^= 1ull << q;
            }
            return;
        }
        for (int q = 0; q < nq; ++q)
            if (qy[q] == ylo && xa > qx[q]) tog ^= 1ull << q;
    };
    {
        const int dy = c_dy(nd), dx = c_dx(nd);
        cross(cy, cx, cy + dy, cx + dx);
        cy += dy; cx += dx; rb += dy * bm.pitch; lin += dy * bw + dx;
    }
    int d = (nd + 6 - (nd & 1)) & 7;
    for (int step = 1; step < max_steps; ++step) {
        nbm = bm.ring(rb, cx);
        nd = (d + __builtin_ctz(((nbm >> d) | (nbm << (8 - d))) & 0xffu)) & 7;
        if (SEG) {
            const unsigned cw = cbm[rb + bm.pitch + 1 + (cx >> 5)];
            if (((cw >> (cx & 31)) & 1u) && nd == ((6 + __builtin_ctz(((nbm >> 6) | (nbm << 2)) & 0xffu)) & 7)) return tog;
        } else if (lin == start_lin && nd == start_d) return tog;
        const int dy = c_dy(nd), dx = c_dx(nd);
        cross(cy, cx, cy + dy, cx + dx);
        cy += dy; cx += dx; rb += dy * bm.pitch; lin += dy * bw + dx;
        d = (nd + 6 - (nd & 1)) & 7;
    }
    return tog;
}

// Bounding-box pre-pass: CT_BOXG workgroups per mask, each over a contiguous 1/CT_BOXG of the pixels; the partial boxes go to the head
// of the mask's `pts` region (consumed by contour_kernel before it writes anything there). One workgroup scanning a 1280x720 mask
// alone took 93 us of the 600-us kernel.
__global__ __launch_bounds__(256) void contour_bbox_kernel(const ContourParams p) {
    __shared__ int s_box[4];
    const int mi = blockIdx.y, g = blockIdx.x, tid = threadIdx.x;
    const uint8_t* M = p.masks + (size_t)mi * p.H * p.W;
    if (tid == 0) { s_box[0] = p.W; s_box[1] = p.H; s_box[2] = -1; s_box[3] = -1; }
    __syncthreads();
    const size_t npix = (size_t)p.H * p.W;
    const size_t per = ((npix + p.boxg - 1) / p.boxg + 15) & ~(size_t)15;
    const size_t i0 = (size_t)g * per, i1 = min(npix, i0 + per);
    int x0 = p.W, y0 = p.H, x1 = -1, y1 = -1;
    for (size_t i = i0 + (size_t)tid * 16; i < i1; i += (size_t)256 * 16) {
        unsigned any = 0;
        const size_t e = min(i + 16, i1);
        if (((uintptr_t)(M + i) & 15) == 0 && e == i + 16) {
            const uint4 v = *(const uint4*)(M + i);
            any = v.x | v.y | v.z | v.w;
        } else {
            for (size_t k = i; k < e; ++k) any |= M[k];
        }
        if (any)
            for (size_t k = i; k < e; ++k)
                if (M[k]) {
                    const int y = (int)(k / p.W), x = (int)(k - (size_t)y * p.W);
                    x0 = min(x0, x); x1 = max(x1, x); y0 = min(y0, y); y1 = max(y1, y);
                }
    }
    if (x1 >= 0) { atomicMin(&s_box[0], x0); atomicMin(&s_box[1], y0); atomicMax(&s_box[2], x1); atomicMax(&s_box[3], y1); }
    __syncthreads();
    if (tid < 4) p.pts[(size_t)mi * p.max_pts * 2 + 4 * g + tid] = s_box[tid];
}

__global__ __launch_bounds__(CT_THREADS) void contour_kernel(const ContourParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int s_box[4];                 // x0, y0, x1, y1 (inclusive)
    __shared__ int s_ncand;
    __shared__ int s_np, s_nhull;
    const int mi = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint8_t* M = p.masks + (size_t)mi * p.H * p.W;
    if (tid == 0) { s_box[0] = p.W; s_box[1] = p.H; s_box[2] = -1; s_box[3] = -1; s_ncand = 0; s_np = 0; s_nhull = 0; }
    __syncthreads();
    CT_STAMP(0);
    // ---- 1a. bounding box: the partial boxes of contour_bbox_kernel ---------------------------------------------------------
    if (tid < p.boxg) {
        const int32_t* pb = p.pts + (size_t)mi * p.max_pts * 2 + 4 * tid;
        const int x0 = pb[0], y0 = pb[1], x1 = pb[2], y1 = pb[3];
        if (x1 >= 0) { atomicMin(&s_box[0], x0); atomicMin(&s_box[1], y0); atomicMax(&s_box[2], x1); atomicMax(&s_box[3], y1); }
    }
    __syncthreads();
    const int bx0 = s_box[0], by0 = s_box[1], bx1 = s_box[2], by1 = s_box[3];
    if (bx1 < 0) {                            // empty mask
        if (tid == 0) { p.count[mi] = 0; if (p.rect) { p.rect[2 * mi] = 0.0; p.rect[2 * mi + 1] = 0.0; } if (p.parts) p.parts[(size_t)mi * p.parts_cap] = 0; }
        return;
    }
    const int bw = bx1 - bx0 + 1, bh = by1 - by0 + 1;
    const int pitch = (bw + 31) / 32 + 3;     // one zero word left, one right, one for the 64-bit window's overrun
    if ((size_t)(bh + 2) * pitch * 4 > (size_t)CT_BITMAP_BYTES || bw > CT_MAXCOL) {
        if (tid == 0) { p.count[mi] = -1; if (p.rect) { p.rect[2 * mi] = 0.0; p.rect[2 * mi + 1] = 0.0; } if (p.parts) p.parts[(size_t)mi * p.parts_cap] = 0; }
        return;
    }
    unsigned* bmw = (unsigned*)smem;
    int* cand = (int*)(smem + CT_BITMAP_BYTES);
    CT_STAMP(1);
    // ---- 1b. bit image --------------------------------------------------------------------------------------------------
    for (int i = tid; i < (bh + 2) * pitch; i += CT_THREADS) bmw[i] = 0u;
    __syncthreads();
    for (int y = wave; y < bh; y += CT_THREADS / 64) {
        const uint8_t* row = M + (size_t)(by0 + y) * p.W + bx0;
        for (int x = 0; x < bw; x += 64) {
            const int xx = x + lane;
            const unsigned long long b = __ballot(xx < bw && row[xx] != 0);
            if (lane == 0) {
                unsigned* dst = bmw + (size_t)(y + 1) * pitch + 1 + (x >> 5);
                dst[0] = (unsigned)b;
                if (x + 32 < bw) dst[1] = (unsigned)(b >> 32);
            }
        }
    }
    __syncthreads();
    Bitmap bm{bmw, pitch};
    CT_STAMP(2);
    // ---- 2. candidates ----------------------------------------------------------------------------------------------------
    const int words_per_row = (bw + 31) / 32;
    for (int i = tid; i < bh * words_per_row; i += CT_THREADS) {
        const int y = i / words_per_row, j = i - y * words_per_row + 1;
        const unsigned* r = bmw + (size_t)(y + 1) * pitch;
        const unsigned* u = r - pitch;
        const unsigned m = r[j];
        if (!m) continue;
        const unsigned wb = (m << 1) | (r[j - 1] >> 31);
        const unsigned ub = u[j];
        const unsigned nwb = (ub << 1) | (u[j - 1] >> 31);
        const unsigned neb = (ub >> 1) | (u[j + 1] << 31);
        unsigned c = m & ~wb & ~ub & ~nwb & ~neb;
        while (c) {
            const int b = __builtin_ctz(c);
            c &= c - 1;
            const int k = atomicAdd(&s_ncand, 1);
            if (k < CT_MAXCAND) cand[k] = y * bw + (j - 1) * 32 + b;
        }
    }
    __syncthreads();
    const int ncand = s_ncand;
    if (ncand > CT_MAXCAND) {
        if (tid == 0) { p.count[mi] = -2; if (p.rect) { p.rect[2 * mi] = 0.0; p.rect[2 * mi + 1] = 0.0; } if (p.parts) p.parts[(size_t)mi * p.parts_cap] = 0; }
        return;
    }
    CT_STAMP(3);
    if (blockIdx.x == 0 && tid == 0) { g_ct_clk[8] = (unsigned long long)ncand; g_ct_clk[10] = (unsigned long long)bw; g_ct_clk[11] = (unsigned long long)bh; }
    // ---- 3. trace every candidate -------------------------------------------------------------------------------------------
    const int max_steps = 4 * bh * bw + 8;
    int32_t* out = p.pts + (size_t)mi * p.max_pts * 2;
    constexpr int CPT = CT_MAXCAND / CT_THREADS;  // one-lane path: candidate tid + i * CT_THREADS lives in my_np[i] = (points << 1 | start point kept), 0 = dropped
    static_assert(CPT * CT_THREADS == CT_MAXCAND, "candidates per thread");
    int my_np[CPT];
    // the list's head (region 0) takes the result; regions 1 .. nslots behind it are the candidates' own lists
    const int slot_cap = p.max_pts >= 8 * CT_SLOT_PTS ? CT_SLOT_PTS : p.max_pts / 8;
    // the assembled list grows from the region's start; the candidates' slots begin behind head_cap points (a one-mask call has room for a
    // 16-slot head: "all" lists of a few thousand points are then still copied out of the slots in parallel instead of re-traced by one lane)
    const int head_cap = p.max_pts >= 48 * slot_cap ? 16 * slot_cap : slot_cap;
    const int nslots = slot_cap >= 8 ? min(CT_MAXSLOTS, (p.max_pts - head_cap) / slot_cap) : 0;
    // outer borders found (their leaders): candidate index, points, start point kept, start pixel, external?
    __shared__ int s_nl, s_no, s_total;
    __shared__ int l_k[CT_NLMAX], l_np[CT_NLMAX], l_sk[CT_NLMAX], l_y[CT_NLMAX], l_x[CT_NLMAX], l_ext[CT_NLMAX], l_ord[CT_NLMAX], l_base[CT_NLMAX];
    __shared__ unsigned long long l_par[CT_NLMAX];
    constexpr int CT_QROWS = 1024;                // rows the per-row query sets cover (a taller box takes the plain loop)
    __shared__ unsigned long long q_row[CT_QROWS];
    __shared__ int s_stored[CT_MAXSLOTS];         // one-lane path: points a candidate with a slot stored in it
    __shared__ unsigned long long s_key;          // "largest" among more than CT_NLMAX borders: the round's best (points, start, candidate)
    __shared__ int s_in, s_qw[2];                 // ... is it inside another border? its start pixel (y, x)
    if (tid == 0) { s_nl = 0; s_no = 0; s_total = 0; }
    if (tid < CT_NLMAX) l_par[tid] = 0ull;
    auto decline = [&](int code) {                // (uniform: every thread takes the same branch)
        if (tid == 0) { p.count[mi] = code; if (p.rect) { p.rect[2 * mi] = 0.0; p.rect[2 * mi + 1] = 0.0; } if (p.parts) p.parts[(size_t)mi * p.parts_cap] = 0; }
    };
    // ---- 3a. segmented trace (see trace_segment): needs a second bit image (the candidate pixels) + per-candidate tables in LDS ----
    constexpr int CT_SEGMAX = 1024;
    const size_t bm_bytes = (((size_t)(bh + 2) * pitch * 4) + 15) & ~(size_t)15;
    const bool seg_mode = ncand <= CT_SEGMAX && nslots > 0 && 2 * bm_bytes + (size_t)CT_SEGMAX * 32 <= (size_t)CT_BITMAP_BYTES;
    unsigned* cbm = (unsigned*)(smem + bm_bytes);
    int* nxt = (int*)(smem + 2 * bm_bytes);                  // successor candidate (index into the sorted list), -1 = none
    int* nkp = nxt + CT_SEGMAX;                              // interior kept points of the segment
    int* nmv = nkp + CT_SEGMAX;                              // moves of the segment (later: output offset of the segment's points)
    int* mvs = nmv + CT_SEGMAX;                              // first move | last move << 4
    int* cyc = mvs + CT_SEGMAX;                              // assembly: the contour's segments in border order
    int* jkp = cyc + CT_SEGMAX;                              // assembly: 1 if the joint point at the segment's start is kept (before: leader -> slot)
    int* mnl = jkp + CT_SEGMAX;                              // raster-first pixel the segment visits
    int* bid = mnl + CT_SEGMAX;                              // leader (candidate index) of the outer border the segment lies on, -1 = none
    auto find = [&](int lin) {                               // index of a candidate pixel in the sorted list (segmented path)
        int lo = 0, hi = ncand - 1;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (cand[mid] < lin) lo = mid + 1; else hi = mid; }
        return cand[lo] == lin ? lo : -1;
    };
    __syncthreads();
    if (seg_mode) {
        // sort the candidates by raster position (bitonic over the next power of two; pads = INT_MAX)
        int n2 = 1;
        while (n2 < ncand) n2 <<= 1;
        for (int i = ncand + tid; i < n2; i += CT_THREADS) cand[i] = 0x7fffffff;
        for (int i = tid; i < (int)(bm_bytes / 4); i += CT_THREADS) cbm[i] = 0u;
        __syncthreads();
        for (int k2 = 2; k2 <= n2; k2 <<= 1)
            for (int j = k2 >> 1; j > 0; j >>= 1) {
                for (int i = tid; i < n2; i += CT_THREADS) {
                    const int l = i ^ j;
                    if (l > i) {
                        const int a = cand[i], b = cand[l];
                        if (((i & k2) == 0) ? (a > b) : (a < b)) { cand[i] = b; cand[l] = a; }
                    }
                }
                __syncthreads();
            }
        for (int k = tid; k < ncand; k += CT_THREADS) {
            const int lin = cand[k];
            const int y = lin / bw, x = lin - y * bw;
            atomicOr(&cbm[(size_t)(y + 1) * pitch + 1 + (x >> 5)], 1u << (x & 31));
        }
        __syncthreads();
        for (int k = tid; k < ncand; k += CT_THREADS) {           // (ncand <= CT_THREADS: one segment per thread)
            const int lin = cand[k];
            const int sy = lin / bw, sx = lin - sy * bw;
            SegInfo si;
            if (k < nslots) trace_segment<true>(bm, cbm, bw, sy, sx, max_steps, out + 2 * ((size_t)head_cap + (size_t)k * slot_cap), slot_cap, bx0, by0, si);
            else trace_segment<false>(bm, cbm, bw, sy, sx, max_steps, nullptr, 0, bx0, by0, si);
            nxt[k] = si.next_lin >= 0 ? find(si.next_lin) : -1;
            nkp[k] = si.nkeep; nmv[k] = si.nmoves; mvs[k] = si.first_move | (si.last_move << 4); mnl[k] = si.min_lin;
        }
        __threadfence_block();
        __syncthreads();
        // borders = cycles of nxt; the raster-first candidate of a cycle leads it. Every candidate walks its cycle once.
        for (int k = tid; k < ncand; k += CT_THREADS) {
            int j = k, len = 0, kept = 0, moves = 0, lead = k, cmin = 0x7fffffff;
            bool ok = true;
            do {
                const int jn = nxt[j];
                if (jn < 0) { ok = false; break; }
                cmin = min(cmin, mnl[j]);
                kept += nkp[j] + (((mvs[j] >> 4) & 15) != (mvs[jn] & 15) ? 1 : 0);      // interior points + the joint point at jn
                moves += nmv[j];
                lead = min(lead, jn);
                j = jn;
            } while (j != k && ++len < ncand);
            // a border counts from its raster-first PIXEL only (as the one-lane trace drops a walk that meets an earlier pixel): a hole
            // border whose first pixel is no local top has candidates but no survivor
            const bool valid = ok && j == k && cmin == cand[lead];
            bid[k] = valid ? lead : -1;
            cyc[k] = 0;
            if (!valid || lead != k) continue;
            const int pj = [&] { int q = k; while (nxt[q] != k) q = nxt[q]; return q; }();    // the segment that ends at the leader
            const int start_kept = (((mvs[pj] >> 4) & 15) != (mvs[k] & 15)) ? 1 : 0;
            int np;
            if (moves == 0) np = 1;                                // isolated pixel
            else if (moves <= 2) np = moves;                       // _compress keeps everything
            else np = kept > 0 ? kept : 1;
            const int sk = (moves <= 2) ? 1 : start_kept;
            cyc[k] = (np << 1) | sk;                               // (leaders only; the table below holds the first CT_NLMAX of them)
            const int li = atomicAdd(&s_nl, 1);
            if (li < CT_NLMAX) {
                l_k[li] = k; l_np[li] = np; l_sk[li] = sk; l_y[li] = cand[k] / bw; l_x[li] = cand[k] % bw; l_ext[li] = 1;
                jkp[k] = li;
            }
        }
        __syncthreads();
        const int nl = s_nl;
        if (nl > CT_NLMAX) {
            if (p.strategy == 1) { decline(-2); return; }
            // "largest" among more borders than the table holds: take the best, test IT against every other outer border's segments, drop
            // it if it is nested, again
            bool found = false;
            for (int round = 0; round < 16 && !found; ++round) {
                if (tid == 0) { s_key = 0ull; s_in = 0; }
                __syncthreads();
                for (int k = tid; k < ncand; k += CT_THREADS)
                    if (cyc[k]) atomicMax(&s_key, ((unsigned long long)(unsigned)(cyc[k] >> 1) << 40) | ((unsigned long long)(unsigned)cand[k] << 16) | (unsigned long long)k);
                __syncthreads();
                const unsigned long long key = s_key;
                const int kw = (int)(key & 0xffffull), lw = (int)((key >> 16) & 0xffffffull);
                if (tid == 0) { s_qw[0] = lw / bw; s_qw[1] = lw % bw; }
                __syncthreads();
                for (int k = tid; k < ncand; k += CT_THREADS) mnl[k] = 0;      // (free since the cycles were resolved) parity per border, at its leader
                __syncthreads();
                for (int k = tid; k < ncand; k += CT_THREADS) {
                    if (bid[k] < 0 || bid[k] == kw) continue;
                    const int lin = cand[k];
                    if (walk_parity<true>(bm, cbm, bw, lin / bw, lin % bw, max_steps, &s_qw[0], &s_qw[1], 1) & 1ull) atomicXor(&mnl[bid[k]], 1);
                }
                __syncthreads();
                for (int k = tid; k < ncand; k += CT_THREADS)
                    if (mnl[k]) s_in = 1;
                __syncthreads();
                if (s_in == 0) {
                    found = true;
                    if (tid == 0) { l_k[0] = kw; l_np[0] = cyc[kw] >> 1; l_sk[0] = cyc[kw] & 1; l_y[0] = s_qw[0]; l_x[0] = s_qw[1]; l_ext[0] = 1; s_nl = 1; }
                } else if (tid == 0) cyc[kw] = 0;
                __syncthreads();
            }
            if (!found) { decline(-2); return; }
        } else if (nl >= 2) {                                      // 3c. which outer borders lie inside another one?
            const bool use_rows = bh <= CT_QROWS;
            if (use_rows) {
                for (int i = tid; i < bh; i += CT_THREADS) q_row[i] = 0ull;
                __syncthreads();
                if (tid < nl) atomicOr(&q_row[l_y[tid]], 1ull << tid);
                __syncthreads();
            }
            for (int k = tid; k < ncand; k += CT_THREADS) {
                if (bid[k] < 0) continue;
                const int own = jkp[bid[k]];
                const int lin = cand[k];
                unsigned long long t = walk_parity<true>(bm, cbm, bw, lin / bw, lin % bw, max_steps, l_y, l_x, nl, use_rows ? q_row : nullptr);
                t &= ~(1ull << own);
                if (t) atomicXor(&l_par[own], t);
            }
            __syncthreads();
            if (tid < nl) {
                bool inside = false;
                for (int x = 0; x < nl; ++x) inside |= (x != tid) && ((l_par[x] >> tid) & 1ull);
                l_ext[tid] = inside ? 0 : 1;
            }
            __syncthreads();
        }
    } else {
        // ---- 3b. one lane per candidate walks its whole border ------------------------------------------------------------------------
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int k = tid + i * CT_THREADS;
            my_np[i] = 0;
            if (k >= ncand) continue;
            const int lin = cand[k];
            const int sy = lin / bw, sx = lin - sy * bw;
            int kept0 = 0, stored = 0, np;
            if (k < nslots) np = moore_trace<true>(bm, bw, sy, sx, max_steps, out + 2 * ((size_t)head_cap + (size_t)k * slot_cap), slot_cap, bx0, by0, &kept0, &stored);
            else np = moore_trace<false>(bm, bw, sy, sx, max_steps, nullptr, 0, bx0, by0, &kept0, &stored);
            if (k < nslots) s_stored[k] = stored;
            if (np > 0) {
                my_np[i] = (np << 1) | kept0;
                const int li = atomicAdd(&s_nl, 1);
                if (li < CT_NLMAX) { l_k[li] = k; l_np[li] = np; l_sk[li] = kept0; l_y[li] = sy; l_x[li] = sx; l_ext[li] = 1; }
            }
        }
        __threadfence_block();
        __syncthreads();
        const int ns = s_nl;
        if (ns >= 2 && p.strategy == 1) {
            if (ns > CT_NLMAX) { decline(-2); return; }
            const bool use_rows = bh <= CT_QROWS;
            if (use_rows) {
                for (int i = tid; i < bh; i += CT_THREADS) q_row[i] = 0ull;
                __syncthreads();
                if (tid < ns) atomicOr(&q_row[l_y[tid]], 1ull << tid);
                __syncthreads();
            }
#pragma unroll
            for (int i = 0; i < CPT; ++i) {
                const int k = tid + i * CT_THREADS;
                if (my_np[i] == 0) continue;
                int own = 0;
                while (l_k[own] != k) ++own;
                const int lin = cand[k];
                unsigned long long t = walk_parity<false>(bm, nullptr, bw, lin / bw, lin % bw, max_steps, l_y, l_x, ns, use_rows ? q_row : nullptr);
                t &= ~(1ull << own);
                if (t) atomicXor(&l_par[own], t);
            }
            __syncthreads();
            if (tid < ns) {
                bool inside = false;
                for (int x = 0; x < ns; ++x) inside |= (x != tid) && ((l_par[x] >> tid) & 1ull);
                l_ext[tid] = inside ? 0 : 1;
            }
            __syncthreads();
        } else if (ns >= 2) {
            // "largest" among possibly thousands of borders (noise masks): take the best, test IT against the borders that start before it
            // (only those can surround it), drop it if it is nested, again - a handful of rounds at most
            bool found = false;
            for (int round = 0; round < 16 && !found; ++round) {
                if (tid == 0) { s_key = 0ull; s_in = 0; }
                __syncthreads();
#pragma unroll
                for (int i = 0; i < CPT; ++i)                      // most points, then the later start (first in the bottom-up list)
                    if (my_np[i]) atomicMax(&s_key, ((unsigned long long)(unsigned)(my_np[i] >> 1) << 40) | ((unsigned long long)(unsigned)cand[tid + i * CT_THREADS] << 16) | (unsigned long long)(tid + i * CT_THREADS));
                __syncthreads();
                const unsigned long long key = s_key;
                const int kw = (int)(key & 0xffffull), lw = (int)((key >> 16) & 0xffffffull);
                if (tid == 0) { s_qw[0] = lw / bw; s_qw[1] = lw % bw; }
                __syncthreads();
#pragma unroll
                for (int i = 0; i < CPT; ++i) {
                    const int k = tid + i * CT_THREADS;
                    if (my_np[i] == 0 || cand[k] >= lw) continue;
                    const int lin = cand[k];
                    if (walk_parity<false>(bm, nullptr, bw, lin / bw, lin % bw, max_steps, &s_qw[0], &s_qw[1], 1) & 1ull) s_in = 1;
                }
                __syncthreads();
                if (s_in == 0) {
                    found = true;
                    if (tid == 0) { l_k[0] = kw; l_np[0] = (int)(key >> 40); l_y[0] = s_qw[0]; l_x[0] = s_qw[1]; l_ext[0] = 1; s_nl = 1; }
#pragma unroll
                    for (int i = 0; i < CPT; ++i)
                        if (tid + i * CT_THREADS == kw) l_sk[0] = my_np[i] & 1;
                } else {
#pragma unroll
                    for (int i = 0; i < CPT; ++i)
                        if (tid + i * CT_THREADS == kw) my_np[i] = 0;
                }
                __syncthreads();
            }
            if (!found) { decline(-2); return; }
        }
    }
    CT_STAMP(4);
    // ---- 4. order, assemble, turn round ------------------------------------------------------------------------------------------------
    const int nl = min(s_nl, CT_NLMAX);
    if (tid == 0) {
        int no = 0, total = 0;
        if (p.strategy == 1) {                                     // every external contour, bottom-up
            for (int i = 0; i < nl; ++i) if (l_ext[i]) l_ord[no++] = i;
            for (int a = 1; a < no; ++a) {
                const int v = l_ord[a];
                const int key = l_y[v] * bw + l_x[v];
                int b = a - 1;
                while (b >= 0 && l_y[l_ord[b]] * bw + l_x[l_ord[b]] < key) { l_ord[b + 1] = l_ord[b]; --b; }
                l_ord[b + 1] = v;
            }
        } else {                                                   // most points; on a tie the first of the bottom-up list = the later start
            int best = -1;
            for (int i = 0; i < nl; ++i) {
                if (!l_ext[i]) continue;
                if (best < 0 || l_np[i] > l_np[best] || (l_np[i] == l_np[best] && l_y[i] * bw + l_x[i] > l_y[best] * bw + l_x[best])) best = i;
            }
            if (best >= 0) l_ord[no++] = best;
        }
        for (int a = 0; a < no; ++a) { l_base[a] = total; total += l_np[l_ord[a]]; }
        s_no = no; s_total = total;
    }
    __syncthreads();
    const int no = s_no, total = s_total;
    if (no == 0 || total > p.max_pts) { decline(no == 0 ? 0 : -2); return; }
    // the head region (head_cap points) holds the whole list when every contour is to be copied out of the candidates' slots; a longer
    // list is re-traced contour by contour by one lane straight into place (it may then run over the slots: nothing reads them any more)
    const bool copy_ok = total <= head_cap;
    const bool par_all = seg_mode && copy_ok;
    if (par_all) {
        // every contour at once: (A) one lane per contour walks its cycle of segments and fixes each segment's output offset and joint
        // point, (B) a wave per SEGMENT - whatever contour it belongs to - copies the segment's points out of its slot, (C) what (A)
        // declined is re-traced by one lane. (One contour after the other - a serial cycle walk and three barriers each - cost 280 us for
        // the 40 contours of an "all" list.)
        __shared__ int l_ok[CT_NLMAX], l_apos[CT_NLMAX];
        if (tid < nl) l_apos[tid] = -1;
        __syncthreads();
        if (tid < no) l_apos[l_ord[tid]] = tid;
        if (tid < nl) cyc[l_k[tid]] = tid;                          // leader's candidate index -> its row of the table (cyc is free here)
        __syncthreads();
        if (tid < no) {
            const int li = l_ord[tid], kb = l_k[li], npq = l_np[li], rotq = l_sk[li], base = l_base[tid];
            bool ok = true;
            int off = base + rotq, j = kb, moves = 0, prev = -1, len = 0;
            while (ok) {
                if (j >= nslots || nkp[j] > slot_cap) { ok = false; break; }
                const int jn = nxt[j];
                const int joint = (prev >= 0 && (((mvs[prev] >> 4) & 15) != (mvs[j] & 15))) ? 1 : 0;   // the joint point at this segment's start
                jkp[j] = joint;
                off += joint;
                moves += nmv[j];
                const int myoff = off;
                off += nkp[j];
                nmv[j] = myoff;                                    // (moves are no longer needed: the slot becomes the offset)
                prev = j;
                ++len;
                j = jn;
                if (j == kb) break;
                if (len >= ncand) { ok = false; break; }
            }
            if (ok && (moves <= 2 || off != base + npq)) ok = false;      // tiny borders and any disagreement go the serial way
            l_ok[tid] = ok ? 1 : 0;
            if (ok && rotq) { out[2 * base] = l_x[li] + bx0; out[2 * base + 1] = l_y[li] + by0; }
        }
        __syncthreads();
        for (int k = wave; k < ncand; k += CT_THREADS / 64) {
            const int b = bid[k];
            if (b < 0) continue;
            const int li = cyc[b];
            if ((unsigned)li >= (unsigned)nl || l_k[li] != b) continue;
            const int a = l_apos[li];
            if (a < 0 || !l_ok[a]) continue;
            const int o0 = nmv[k];
            if (lane == 0 && jkp[k]) { const int l = cand[k]; const int yy = l / bw, xx = l - yy * bw; out[2 * (o0 - 1)] = xx + bx0; out[2 * (o0 - 1) + 1] = yy + by0; }
            const int32_t* sp = out + 2 * ((size_t)head_cap + (size_t)k * slot_cap);
            for (int q = lane; q < 2 * nkp[k]; q += 64) out[2 * o0 + q] = sp[q];
        }
        if (tid == 0) {
            for (int a = 0; a < no; ++a) {
                if (l_ok[a]) continue;
                const int li = l_ord[a], rotq = l_sk[li], base = l_base[a];
                int kept0 = 0, stored = 0;
                moore_trace<true>(bm, bw, l_y[li], l_x[li], max_steps, out + 2 * (base + rotq), p.max_pts - base - rotq, bx0, by0, &kept0, &stored);
                if (rotq) { out[2 * base] = l_x[li] + bx0; out[2 * base + 1] = l_y[li] + by0; }
            }
        }
        __threadfence_block();
        __syncthreads();
    }
    for (int a = 0; a < (par_all ? 0 : no); ++a) {
        const int li = l_ord[a], kb = l_k[li], npq = l_np[li], rotq = l_sk[li], base = l_base[a];
        const int sy = l_y[li], sx = l_x[li];
        __shared__ int s_cyc_len, s_par_ok;
        if (tid == 0) {
            bool ok = copy_ok;
            int len = 0;
            if (ok && seg_mode) {
                // the contour's segments in order with the output offset of their points
                int off = base + rotq, j = kb, moves = 0;
                while (ok) {
                    if (j >= nslots || nkp[j] > slot_cap) { ok = false; break; }
                    const int jn = nxt[j];
                    int joint = 0;
                    if (j != kb) {                                 // the joint point at this segment's start
                        const int q = cyc[len - 1];
                        joint = (((mvs[q] >> 4) & 15) != (mvs[j] & 15)) ? 1 : 0;
                    }
                    jkp[j] = joint;
                    off += joint;
                    moves += nmv[j];
                    cyc[len++] = j;
                    const int myoff = off;
                    off += nkp[j];
                    nmv[j] = myoff;                                // (moves are no longer needed: the slot becomes the offset)
                    j = jn;
                    if (j == kb) break;
                    if (len >= ncand) { ok = false; break; }
                }
                if (ok && (moves <= 2 || off != base + npq)) ok = false;      // tiny borders and any disagreement go the serial way
            } else if (ok) {
                ok = kb < nslots && s_stored[kb] == npq - rotq && npq - rotq <= slot_cap;
            }
            s_par_ok = ok ? 1 : 0;
            s_cyc_len = len;
        }
        __syncthreads();
        if (s_par_ok && seg_mode) {
            const int len = s_cyc_len;
            for (int i = wave; i < len; i += CT_THREADS / 64) {     // a wave per segment: joint point, then the segment's list
                const int j = cyc[i];
                const int o0 = nmv[j];
                if (lane == 0 && jkp[j]) { const int l = cand[j]; const int yy = l / bw, xx = l - yy * bw; out[2 * (o0 - 1)] = xx + bx0; out[2 * (o0 - 1) + 1] = yy + by0; }
                const int32_t* sp = out + 2 * ((size_t)head_cap + (size_t)j * slot_cap);
                for (int q = lane; q < 2 * nkp[j]; q += 64) out[2 * o0 + q] = sp[q];
            }
            if (tid == 0 && rotq) { out[2 * base] = sx + bx0; out[2 * base + 1] = sy + by0; }
        } else if (s_par_ok) {
            const int32_t* sp = out + 2 * ((size_t)head_cap + (size_t)kb * slot_cap);
            for (int j = tid; j < 2 * (npq - rotq); j += CT_THREADS) out[2 * (base + rotq) + j] = sp[j];
            if (tid == 0 && rotq) { out[2 * base] = sx + bx0; out[2 * base + 1] = sy + by0; }
        } else if (tid == 0) {
            int kept0 = 0, stored = 0;
            moore_trace<true>(bm, bw, sy, sx, max_steps, out + 2 * (base + rotq), p.max_pts - base - rotq, bx0, by0, &kept0, &stored);
            if (rotq) { out[2 * base] = sx + bx0; out[2 * base + 1] = sy + by0; }
        }
        __threadfence_block();
        __syncthreads();
    }
    // the walkers go clockwise (east first); cv2 follows an outer border the other way (down first) from the same start pixel: the kept
    // pixels are the same set, so each contour's list is turned round behind its start point
    for (int a = 0; a < no; ++a) {
        const int li = l_ord[a];
        const int lo = l_base[a] + l_sk[li], hi = l_base[a] + l_np[li] - 1;        // reverse out[lo .. hi]
        for (int i = tid; lo + i < hi - i; i += CT_THREADS) {
            const int u = lo + i, v = hi - i;
            const int ux = out[2 * u], uy = out[2 * u + 1];
            out[2 * u] = out[2 * v]; out[2 * u + 1] = out[2 * v + 1];
            out[2 * v] = ux; out[2 * v + 1] = uy;
        }
    }
    if (tid == 0) {
        p.count[mi] = total;
        s_np = total;
        if (p.parts) {
            int32_t* pp = p.parts + (size_t)mi * p.parts_cap;
            pp[0] = no;
            for (int a = 0; a < no && a + 1 < p.parts_cap; ++a) pp[a + 1] = l_np[l_ord[a]];
        }
        if (blockIdx.x == 0) g_ct_clk[9] = (unsigned long long)total;
    }
    __threadfence_block();
    __syncthreads();                          // (the bit image is dead from here on: its LDS becomes the hull's tables)
    CT_STAMP(5);
    if (!p.rect) return;
    const int np = s_np;
    int* colmin = (int*)smem;
    int* colmax = colmin + CT_MAXCOL;
    int* hull = colmax + CT_MAXCOL;           // up to 2 * CT_MAXCOL + 2 vertices, (x, y) interleaved; behind it the two chains under construction
    static_assert((size_t)(2 * CT_MAXCOL + 2 * (2 * CT_MAXCOL + 2) + 4 * (CT_MAXCOL + 2)) * sizeof(int) <= (size_t)CT_BITMAP_BYTES, "hull tables fit the bit image's LDS");
    for (int i = tid; i < bw; i += CT_THREADS) { colmin[i] = 0x7fffffff; colmax[i] = -1; }
    __syncthreads();
    for (int i = tid; i < np; i += CT_THREADS) {
        const int x = out[2 * i] - bx0, y = out[2 * i + 1];
        atomicMin(&colmin[x], y);
        atomicMax(&colmax[x], y);
    }
    __syncthreads();
    // Andrew's monotone chain over (x, colmin[x]), (x, colmax[x]) in (x, y) order: lower chain left->right, upper chain right->left;
    // pops on cross <= 0 (no collinear vertices) - the points of a column between its extremes are popped by the full algorithm
    // too, so this is exactly hostops._convex_hull on the unique points. The two chains are independent stacks: lane 0 of wave 0
    // builds the lower one while lane 0 of wave 1 builds the upper one (each keeps the top two vertices in registers), then
    // hull = lower[:-1] + upper[:-1].
    __shared__ int s_chain_n[2], s_nuniq, s_first[2], s_last[2];
    int* const chain_lo = hull + 2 * (2 * CT_MAXCOL + 2);     // a chain holds at most one vertex per column + 2 while it is built
    int* const chain_up = chain_lo + 2 * (CT_MAXCOL + 2);
    if (tid == 0) {
        int nuniq = 0, fx = 0, fy = 0, gx = 0, gy = 0;
        for (int x = 0; x < bw; ++x) {
            const int lo = colmin[x], hi = colmax[x];
            if (hi >= 0) {
                if (nuniq == 0) { fx = x; fy = lo; }
                nuniq += (lo != hi) ? 2 : 1;
                gx = x; gy = hi;
            }
        }
        s_nuniq = nuniq; s_first[0] = fx; s_first[1] = fy; s_last[0] = gx; s_last[1] = gy;
    }
    if (tid == 0 || tid == 64) {
        const bool upper = tid == 64;
        int* const st = upper ? chain_up : chain_lo;
        int n = 0;
        int ax = 0, ay = 0, bx = 0, by = 0;                   // st[n-2], st[n-1] (valid when n >= 2 / n >= 1)
        auto push = [&](int qx, int qy) {
            while (n >= 2 && (long long)(bx - ax) * (qy - ay) - (long long)(by - ay) * (qx - ax) <= 0) {
                --n;                                           // pop: the new top is the old second, the new second comes from LDS
                bx = ax; by = ay;
                if (n >= 2) { ax = st[2 * (n - 2)]; ay = st[2 * (n - 2) + 1]; }
            }
            st[2 * n] = qx; st[2 * n + 1] = qy; ++n;
            ax = bx; ay = by; bx = qx; by = qy;
        };
        if (!upper) {
            for (int x = 0; x < bw; ++x) {
                const int lo = colmin[x], hi = colmax[x];
                if (hi >= 0) {
                    push(x, lo);
                    if (hi != lo) push(x, hi);
                }
            }
        } else {
            for (int x = bw - 1; x >= 0; --x) {
                const int lo = colmin[x], hi = colmax[x];
                if (hi >= 0) {
                    if (hi != lo) push(x, hi);
                    push(x, lo);
                }
            }
        }
        s_chain_n[upper ? 1 : 0] = n;
    }
    __syncthreads();
    {
        const int nuniq = s_nuniq;
        if (nuniq <= 2) {
            if (tid == 0) {
                hull[0] = s_first[0]; hull[1] = s_first[1];
                if (nuniq == 2) { hull[2] = s_last[0]; hull[3] = s_last[1]; }
                s_nhull = nuniq < 2 ? 1 : 2;
            }
        } else {
            const int nlo = s_chain_n[0] - 1, nup = s_chain_n[1] - 1;     // lo[:-1], up[:-1]
            for (int i = tid; i < nlo; i += CT_THREADS) { hull[2 * i] = chain_lo[2 * i]; hull[2 * i + 1] = chain_lo[2 * i + 1]; }
            for (int i = tid; i < nup; i += CT_THREADS) { hull[2 * (nlo + i)] = chain_up[2 * i]; hull[2 * (nlo + i) + 1] = chain_up[2 * i + 1]; }
            if (tid == 0) s_nhull = nlo + nup;
        }
    }
    __syncthreads();
    const int nh = s_nhull;
    if (nh <= 2) {
        if (tid == 0) {
            double len = 0.0;
            if (nh == 2) { const double ddx = hull[2] - hull[0], ddy = hull[3] - hull[1]; len = hypot(ddx, ddy); }
            p.rect[2 * mi] = len; p.rect[2 * mi + 1] = 0.0;
        }
        return;
    }
    // rotating calipers: one hull edge per thread, float64 as hostops.min_area_rect_size (u = e/|e|, v = (-u.y, u.x), extents of the
    // projections); the smallest area wins, ties by the lower edge index (the host loop keeps the first minimum)
    __shared__ double s_area[CT_THREADS / 64], s_w[CT_THREADS / 64], s_h[CT_THREADS / 64];
    __shared__ int s_idx[CT_THREADS / 64];
    double barea = 1e300, bwid = 0, bhei = 0;
    int bidx = 0x7fffffff;
    for (int i = tid; i < nh; i += CT_THREADS) {
        const int j = (i + 1 == nh) ? 0 : i + 1;
        const double ex = (double)(hull[2 * j] - hull[2 * i]), ey = (double)(hull[2 * j + 1] - hull[2 * i + 1]);
        const double nrm = hypot(ex, ey);
        const double ux = ex / nrm, uy = ey / nrm;
        const double vx = -uy, vy = ux;
        double amin = 1e300, amax = -1e300, bmin = 1e300, bmax = -1e300;
        for (int k = 0; k < nh; ++k) {
            const double hx = (double)(hull[2 * k] + bx0), hy = (double)hull[2 * k + 1];
            const double a = hx * ux + hy * uy, b = hx * vx + hy * vy;
            amin = fmin(amin, a); amax = fmax(amax, a); bmin = fmin(bmin, b); bmax = fmax(bmax, b);
        }
        const double w = amax - amin, h = bmax - bmin;
        if (w * h < barea || (w * h == barea && i < bidx)) { barea = w * h; bwid = w; bhei = h; bidx = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double oa = __shfl_xor(barea, o), ow = __shfl_xor(bwid, o), oh = __shfl_xor(bhei, o);
        const int oi = __shfl_xor(bidx, o);
        if (oa < barea || (oa == barea && oi < bidx)) { barea = oa; bwid = ow; bhei = oh; bidx = oi; }
    }
    if (lane == 0) { s_area[wave] = barea; s_w[wave] = bwid; s_h[wave] = bhei; s_idx[wave] = bidx; }
    __syncthreads();
    if (tid == 0) {
        for (int k = 1; k < CT_THREADS / 64; ++k)
            if (s_area[k] < barea || (s_area[k] == barea && s_idx[k] < bidx)) { barea = s_area[k]; bwid = s_w[k]; bhei = s_h[k]; bidx = s_idx[k]; }
        p.rect[2 * mi] = fmax(bwid, bhei); p.rect[2 * mi + 1] = fmin(bwid, bhei);
    }
    CT_STAMP(6);
}

hipError_t contour_read_clocks(unsigned long long* out12) { return hipMemcpyFromSymbol(out12, HIP_SYMBOL(g_ct_clk), 12 * sizeof(unsigned long long)); }

hipError_t launch_contours(const uint8_t* masks, int n, int H, int W, int strategy, int max_pts, int32_t* pts, int32_t* count, int32_t* parts, int parts_cap,
                           double* rect, hipStream_t st) {
    if (n == 0) return hipSuccess;
    ContourParams p{masks, n, H, W, max_pts, pts, count, strategy, parts, parts ? parts_cap : 0, rect, 0};
    p.boxg = max_pts / 2 < CT_BOXG ? max_pts / 2 : CT_BOXG;      // a partial box takes two points' worth of the list
    if (p.boxg < 1) return hipErrorInvalidValue;
    const size_t sh = (size_t)CT_BITMAP_BYTES + (size_t)CT_MAXCAND * sizeof(int);
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)contour_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        if (e != hipSuccess) return e;
        attr = true;
    }
    hipLaunchKernelGGL(contour_bbox_kernel, dim3(p.boxg, n), dim3(256), 0, st, p);
    hipLaunchKernelGGL(contour_kernel, dim3(n), dim3(CT_THREADS), sh, st, p);
    return hipGetLastError();
}

}  // namespace yp
