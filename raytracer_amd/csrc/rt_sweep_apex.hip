// rt_sweep_apex.hip -- brute-force sweeps for rays that share a point: the two ray families of
// the reference's pixel loop.
//
//   * primary rays all start at the eye (driver.c:121-133: S = 0, E = (px - W/2, py - H/2, ez));
//   * the shadow rays of light l all end at that light (light_model.c:60-70).
//
// For such rays the tvec x D term of the Cramer solve (triangle_intersection.cu:59-94)
// vanishes when the triangle is expressed relative to the shared point P, and the three
// determinants become plain dot products of the ray direction with per-triangle vectors
// (k_pack_apex, rt_transform.hip):   U = D.p   W' = det - U = D.q   V = D.h   det = D.nn.
// Every (ray, triangle) pair first meets a cheap conservative test -- the triangle's bounding box
// on a plane in front of P: the image plane for the pixel grid, the plane of the rays' dominant
// axis for the shadow rays -- and the rare path behind it finishes the inside test (U, V, det,
// the three edges, the ray parameter) and appends survivors to the ray's candidate list; the
// fp64 pass of rt_exact.hip then decides with the reference's exact arithmetic.
//
//   k_sweep_apex    any-hit, ray by ray, for the shadow rays of up to four lights (the phases too
//                   small for the cell-lattice sweep of rt_sweep_cells.hip, and tile culling).
//                   Directions are scaled to a unit dominant component, D' = D / D_k (rays binned
//                   by that axis, k_bin_rays), i.e. a ray is the point (u, v) on the plane
//                   D_k = 1 through P, and it can only hit a triangle whose central projection
//                   onto that plane contains (u, v).  First test of a pair: the u half of the
//                   projection's bounding square (k_pack_apex writes one per triangle and axis,
//                   allowances included), |u - cu| <= rho: a subtraction per pair and a three-way
//                   minimum per two, 13 VALU instructions per 8 pairs; the wave completes the
//                   square for the few triangles that have one of its rays in their u strip, and
//                   the rare path behind that is the full inside test on the four dot products.
//                   Triangles in the order of their solid angle at the light, triangle ranges
//                   with survivor compaction in between, triangle chunks that share a ray's
//                   state through atomics, an occluder cache in front (host side:
//                   trace_shadow_rays in rt_api.hip).
//   k_sweep_raster  closest hit for the pixel grid, lattice form.  D = (X, Y, ez) with integer X, Y
//                   and a constant ez that k_pack_apex folds into the z components.  Each lane
//                   owns up to 2048 pixels, a lattice of rows and columns of the image.
//                   First test of a pair: the pixel against the bounding box of the
//                   triangle's projection onto the image plane (k_pack_apex writes one per
//                   triangle, allowances included).  A pixel is in the box when its column is
//                   in the box's x extent and its row in the y extent, so a lane evaluates one
//                   interval test per column and per row of its lattice -- 146 VALU instructions
//                   per 2048 pairs (64 columns, 32 rows) -- and every pixel's verdict is the one it
//                   would get on its own.  Primary rays leave P, which fixes the sign of det for a
//                   hit, so k_pack_apex orients every triangle (sign layout) for the rare path's
//                   inside test.
//   k_sweep_raster_wave   the same test one level up, the default for meshes large enough to
//                   fill the machine with it: a workgroup owns a block of 4096 columns by up to
//                   2304 rows, and the block's columns AND rows are dealt to the lanes of a wave
//                   (the block's pixels are the product of the two) -- 98 + 1.5 NL instructions
//                   per wave and triangle for 4096 x 64 NL pixels.
//
// The sweeps stream what their first test reads through LDS in tiles of RT_TILE triangles,
// double-buffered by LDS-DMA loads and read at wave-uniform addresses (broadcast, no bank
// conflicts): the boxes for the pixel grid (the rare path fetches a triangle's planes from
// memory), the squares and the four planes for the any-hit sweep.
#include "rt_sweep_common.h"

#ifndef RT_APEX8_WAVES
#define RT_APEX8_WAVES 6    // waves per SIMD the 8-rays-per-lane any-hit kernel must fit
#endif
#ifndef RT_APEX16_WAVES
#define RT_APEX16_WAVES 5   // ... and the 16-rays-per-lane one (96 registers; measured: six waves spill
                            // into the hot loop, four leave LDS latency exposed)
#endif
#ifndef RT_APEX_POLL
#define RT_APEX_POLL 16     // tiles between looks at the flags other chunks may have set (power of two)
#endif
#ifndef RT_APEX_ROUNDS
#define RT_APEX_ROUNDS 16    // rounds of workgroups an any-hit launch is cut into (triangle chunks)
#endif
#ifndef RT_APEX_UNROLL
#define RT_APEX_UNROLL 4
#endif
#ifndef RT_CULL_GROUP
#define RT_CULL_GROUP 4     // tiles whose bounding squares the culled any-hit sweep stages at a time
#endif
#ifndef RT_RASTER_UNROLL
#define RT_RASTER_UNROLL 4
#endif
#ifndef RT_RASTER_CHECK_INTERVAL
#define RT_RASTER_CHECK_INTERVAL 0   // checking build: count columns and rows that pass the first test outside the rare path's rectangle
#endif
#ifndef RT_RASTER_WAVES
#define RT_RASTER_WAVES 8   // waves per SIMD the pixel-grid sweep must fit (64 registers: the rare path
                            // spills a few, the hot loop gains 3 % over six waves at 76)
#endif

namespace {

using namespace rt_sweep;
constexpr int kRasterUnroll = RT_RASTER_UNROLL;
constexpr int kApexUnroll = RT_APEX_UNROLL;

struct Tri2 { float4 P, Q; };

__device__ __forceinline__ Tri2 load_tri2(const float4* p) { return {p[0], p[RT_TILE]}; }

// ------------------------------------------------------------------------------------------
// primary rays on the pixel grid
// ------------------------------------------------------------------------------------------
// What the sweep keeps per pixel lives in memory, not in registers: cnt[] (candidates listed so
// far) and best_hi[] (upper bound of the true t of the closest certain hit so far).  Only the
// rare path touches them, and the hot loop is left with few enough registers for eight waves
// per SIMD.  The workgroups that sweep different triangle chunks for the same pixels share that
// state: a candidate takes its list slot with an atomic increment, and best_hi only ever
// decreases (atomic minimum on the bit pattern: the values are positive floats), so a certain hit
// found in one chunk prunes the candidates of all the others.  The order of a pixel's list is
// then arbitrary; the exact pass picks the winner by (t, triangle number), which is what the
// reference's ascending loop with its strict '<' keeps (triangle_intersection.cu:167-176).
__device__ __forceinline__ void push_mem(const SweepArgs& A, long slot, int tri)
{
    const int c = atomicAdd(&A.cnt[slot], 1);
    // (slot-major: the pixels of a box are neighbours and list the triangle at the same time, so
    // their entries share cache lines -- pixel-major, every entry dirtied a sector of its own)
    if (c < RT_RASTER_LIST_K) A.list[c * A.list_stride + slot] = tri;
}

// The inside test for one pixel that lies in the bounding box of triangle `tri` (candidate() of
// rt_sweep_common.h with the bookkeeping in memory).  Xs = the pixel's X / RT_RASTER_COLS;
// P, Hh, N = the triangle's planes 0, 2, 3 (sign layout of k_pack_apex), margin = plane 1's w.
__device__ __forceinline__ void finish_raster(const SweepArgs& A, long slot, int tri, const float4& P, float margin,
                                              const float4& Hh, const float4& N, float Xs, float Y)
{
    if (margin == INF) {            // no fp32 verdict for this triangle (see k_pack_apex)
        push_mem(A, slot, tri);
        return;
    }
    const float X = (float)RT_RASTER_COLS * Xs;
    // plane 0 holds -(U + delta)
    const float U = -__builtin_fmaf(Xs, P.x, __builtin_fmaf(Y, P.y, P.z)) - P.w;
    const float V = __builtin_fmaf(X, Hh.x, __builtin_fmaf(Y, Hh.y, Hh.z));
    const float d = __builtin_fmaf(X, N.x, __builtin_fmaf(Y, N.y, N.z));
    const float a = U * d, b = V * d;
    const float w = __builtin_fmaf(d, d, -a) - b;
    const float m3 = __builtin_fminf(__builtin_fminf(a, b), w);
    if (!(__builtin_fmaf(margin, __builtin_fabsf(d), m3) > 0.0f)) return;
    const float ad = __builtin_fabsf(d), en = N.w;
    const float den = ad - en;
    if (!(den > 0.0f)) {            // determinant not resolved: the exact pass decides
        push_mem(A, slot, tri);
        return;
    }
    const float T = Hh.w, eT = U22 * __builtin_fabsf(T);
    const float t = T / d;          // rays start at the apex: t0 = 0
    const float terr = (eT + __builtin_fabsf(t) * en) / den * 1.000001f + U22 * __builtin_fabsf(t);
    // (an agent-scope load: another chunk's workgroup may have lowered the bound; a stale value
    // would only prune less)
    const float best_hi = __int_as_float(__hip_atomic_load(reinterpret_cast<const int*>(A.best_hi + slot),
                                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    if (!(t >= -terr) || !(t - terr <= best_hi)) return;
    push_mem(A, slot, tri);
    // a certain hit (inside by the full margin, t resolved) bounds the closest hit's t from above
    if (m3 >= margin * ad && t > terr && t + terr < best_hi)
        atomicMin(reinterpret_cast<int*>(A.best_hi + slot), __float_as_int(t + terr));
}

// One triangle against the R pixels of a lane, a lattice of RY rows by RX columns (the columns
// RasterLane<R>::SX pixels apart, the rows SY local rows: the lanes of a wave interleave, so that
// the pixels of a small box spread over many lanes and the rare path runs them side by side).
// box = (cx, rx, cy, ry): the triangle's bounding box on the pixel grid, x in units of
// RT_RASTER_COLS pixels (k_pack_apex).  A pixel lies in the box when its column lies in the box's
// x extent and its row in the y extent, so the first test of the lane's R pairs is one interval
// test per column, |Xs - cx| <= rx (Xs - cx steps by a constant from column to column: one
// addition each), one per row, |Y - cy| <= ry (one subtraction each), and a three-way minimum
// per two of either: some pixel of the lattice is in the box exactly when some column and some
// row are.  A lane with a pixel inside fetches the triangle's planes from memory (tri is
// wave-uniform) and runs the inside test on the pixels of the box.
template <int R>
struct RasterLane {
    static constexpr int RY = rt_raster_lane_rows(R), RX = R / RY;
    static constexpr int WH = rt_raster_wave_rows(R), WW = rt_raster_wave_cols(R);   // the wave's pixels
    static constexpr int SY = WH / RY;                          // local rows between a lane's rows
    static constexpr int SX = 64 / SY;                          // pixels between a lane's columns
    static constexpr float KX = (float)SX / (float)RT_RASTER_COLS;   // the same in the boxes' x unit
    static_assert(SX * SY == 64 && SX * RX == WW && SY * RY == WH && SX % RT_RASTER_COLS == 0,
                  "the lanes' lattices tile the wave's pixels");
};

template <int R>
__device__ __forceinline__ void raster_triangle(const SweepArgs& A, const float4& box, int tri, float Xs0,
                                                const float (&Y)[rt_raster_lane_rows(R)], const float* y_of_row,
                                                int px0, int lr0, long slot0)
{
    using L = RasterLane<R>;
    constexpr int RY = L::RY, RX = L::RX;
    constexpr float KX = L::KX;
    const float d0 = Xs0 - box.x;
    float accx = INF;
    if (RX == 1) accx = __builtin_fabsf(d0);
#pragma unroll
    for (int c = 0; c + 1 < RX; c += 2)     // (d0 + 0.0f is not d0 to the compiler: it would keep the addition)
        accx = min3_abs(accx, c == 0 ? d0 : d0 + KX * (float)c, d0 + KX * (float)(c + 1));
    float dy[RY];
#pragma unroll
    for (int j = 0; j < RY; ++j) dy[j] = Y[j] - box.z;      // (a row beyond the band has Y = NaN and never passes)
    float accy = INF;
    if (RY == 1) accy = __builtin_fabsf(dy[0]);
#pragma unroll
    for (int j = 0; j + 1 < RY; j += 2) accy = min3_abs(accy, dy[j], dy[j + 1]);
    if (__builtin_expect(accx <= box.y && accy <= box.w, 0)) {
        // the columns of the box: c with |d0 + KX c| <= rx, one more on either side for the rounding
        // of these two bounds (the inside test decides about every pixel it is given)
        int c_lo = (int)__builtin_floorf(__builtin_fmaxf((-box.y - d0) * (1.0f / KX), 0.0f)) - 1;
        int c_hi = (int)__builtin_ceilf(__builtin_fminf((box.y - d0) * (1.0f / KX), (float)(RX - 1))) + 1;
        c_lo = c_lo < 0 ? 0 : c_lo;
        c_hi = c_hi > RX - 1 ? RX - 1 : c_hi;
#if RT_RASTER_CHECK_INTERVAL
        {
            unsigned long long pass = 0;
#pragma unroll
            for (int k = 0; k < RX; ++k) pass |= (__builtin_fabsf(d0 + KX * (float)k) <= box.y ? 1ull : 0ull) << k;
            const unsigned long long inside =
                c_lo > c_hi ? 0ull : (((c_hi >= 63 ? 0ull : (2ull << c_hi)) - 1ull) & ~((1ull << c_lo) - 1ull));
            if ((pass & ~inside) && A.tests) atomicAdd(A.tests, 1ull << 48);
        }
#endif
        const float4* __restrict__ g = A.planes + tri;
        const float4 P = g[0], Hh = g[2 * A.plane_stride], N = g[3 * A.plane_stride];
        const float margin = g[A.plane_stride].w;
        unsigned rows_in = 0;           // the rows of the box: each row's own first test
#pragma unroll
        for (int j = 0; j < RY; ++j) rows_in |= (__builtin_fabsf(dy[j]) <= box.w ? 1u : 0u) << j;
        while (rows_in) {
            const int j = __builtin_ctz(rows_in);
            rows_in &= rows_in - 1;
            // (Y[j] for a j that differs from lane to lane: one LDS read from the wave's table of
            // its rows instead of a chain of RY selects)
            const float Yj = y_of_row[L::SY * j];
            // (the column's own first test, the hot loop's expression: c_lo and c_hi carry a column
            // of slack on either side, which for a box narrower than the lattice's pitch are two
            // of three; a pixel outside the box cannot be hit and would only be rejected later)
            for (int c = c_lo; c <= c_hi; ++c)
                if (__builtin_fabsf(d0 + KX * (float)c) <= box.y &&
                    px0 + L::SX * c < A.W && lr0 + L::SY * j < A.rows.rows)      // ... and the pixel exists
                    finish_raster(A, slot0 + (long)(j * L::SY) * A.W + L::SX * c, tri, P, margin, Hh, N,
                                  Xs0 + KX * (float)c, Yj);
        }
    }
}

// The same for a lattice whose rows are regular: every aligned group of 16 local rows of the call's
// row set is 16 consecutive image rows (a contiguous band, or interleaved blocks whose height is
// a multiple of 16), so within a group the rows' distances from the box's centre step by a
// constant like the columns', and a lane holds one coordinate per group (YG; NaN for a group
// beyond the band) instead of one per row: few enough registers for eight waves per SIMD, and
// room for lattices of 64 rows.  A row of a partly filled last group that does not exist passes
// like its neighbours would; the rare path asks whether a pixel exists.
template <int R>
struct RasterGroups {
    using L = RasterLane<R>;
    static constexpr int G = L::WH / RT_TILE_H;     // groups of 16 local rows in a wave
    static constexpr int RG = L::RY / G;            // rows of a lane's lattice per group
    static_assert(G >= 1 && RG * G == L::RY, "a lane's rows split evenly over the wave's 16-row groups");
};

// Some pixel of the wave lies in the triangle's box.  The pixels of the wave's block that do are
// the crossing of a run of its columns with those of its rows whose Y lies in the box's y extent,
// and the whole wave works through them together, 64 pixels at a time -- a lane's own pixels
// in a box are a handful of one column of its lattice, and walked lane by lane the wave runs as
// many rounds as the box is tall with a quarter of its lanes at work.  Every one of those pixels
// gets the inside test of finish_raster on its own.
// vbox: the triangle's (cx, rx, cy, ry), wave-uniform; wave_x0, wave_y0: the block's first pixel
// and local row; rows_y: the Y of the block's local rows (LDS, NaN beyond the band); row_list:
// LDS scratch of the wave, one entry per local row of the block.
// The column bounds are rounded inwards to whole pixels: the box was grown by 0.01 pixel and more
// for the roundings of a test on it, a hundred times what the bounds' own roundings come to.
template <int R>
__device__ __forceinline__ void raster_rare_block(const SweepArgs& A, const float4& vbox, int tri, int wave_x0,
                                                  int wave_y0, const float* rows_y, int* row_list, int lane)
{
    using L = RasterLane<R>;
    constexpr float COLS = (float)RT_RASTER_COLS;
    const float bx = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(vbox.x)));
    const float rx = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(vbox.y)));
    const float by = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(vbox.z)));
    const float ry = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(vbox.w)));
    // columns: pixels px with |(px - W/2) / COLS - cx| <= rx inside the block and the image
    const int x_end = wave_x0 + L::WW < A.W ? wave_x0 + L::WW : A.W;
    const float half_w = (float)(A.W / 2);
    const float xl = __builtin_fmaxf(__builtin_fmaf(COLS, bx - rx, half_w), (float)wave_x0);
    const float xh = __builtin_fminf(__builtin_fmaf(COLS, bx + rx, half_w), (float)(x_end - 1));
    if (!(xl <= xh)) return;
    const int x_lo = (int)__builtin_ceilf(xl), nx = (int)__builtin_floorf(xh) - x_lo + 1;
    if (nx <= 0) return;
    // rows: the local rows with |Y - cy| <= ry, listed in LDS (a row beyond the band has Y = NaN)
    int ny = 0;
#pragma unroll
    for (int r0 = 0; r0 < L::WH; r0 += 64) {
        const int r = r0 + lane;
        const bool in = r < L::WH && __builtin_fabsf(rows_y[r < L::WH ? r : 0] - by) <= ry;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(in);
        if (in) row_list[ny + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = r;
        ny += __builtin_popcountll(m);
    }
    wave_lds_sync();
    if (ny == 0) return;
    const float4* __restrict__ g4 = A.planes + tri;
    const float4 P = g4[0], Hh = g4[2 * A.plane_stride], N = g4[3 * A.plane_stride];
    const float margin = g4[A.plane_stride].w;
    const int npix = nx * ny;
    const float inv_nx = 1.0f / (float)nx;
    for (int k = lane; k < npix; k += 64) {
        // (k + 1/2) / nx: the quotient's rounding stays far below 1 / (2 nx) for k < 2^20, nx <= 2^12
        const int r = (int)__builtin_floorf(((float)k + 0.5f) * inv_nx);
        const int c = k - r * nx;
        const int row = row_list[r];        // local row within the block
        const int px = x_lo + c;
        finish_raster(A, (long)(wave_y0 + row) * A.W + px, tri, P, margin, Hh, N, (float)(px - A.W / 2) * (1.0f / COLS),
                      rows_y[row]);
    }
}

template <int R>
__device__ __forceinline__ void raster_triangle_reg(const SweepArgs& A, const float4& box, int tri, float Xs0,
                                                    const float (&YG)[RasterGroups<R>::G], const float* rows_y,
                                                    int* row_list, int wave_x0, int wave_y0, int lane)
{
    using L = RasterLane<R>;
    constexpr int RX = L::RX, G = RasterGroups<R>::G, RG = RasterGroups<R>::RG;
    constexpr float KX = L::KX, KY = (float)L::SY;
    const float d0 = Xs0 - box.x;
    float accx = INF;
    if (RX == 1) accx = __builtin_fabsf(d0);
#pragma unroll
    for (int c = 0; c + 1 < RX; c += 2)     // (d0 + 0.0f is not d0 to the compiler: it would keep the addition)
        accx = min3_abs(accx, c == 0 ? d0 : d0 + KX * (float)c, d0 + KX * (float)(c + 1));
    float e[G];
    float accy = INF;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        e[g] = YG[g] - box.z;
        if (RG == 1) accy = __builtin_fminf(accy, __builtin_fabsf(e[g]));
#pragma unroll
        for (int j = 0; j + 1 < RG; j += 2)
            accy = min3_abs(accy, j == 0 ? e[g] : e[g] + KY * (float)j, e[g] + KY * (float)(j + 1));
    }
    const bool mine = accx <= box.y && accy <= box.w;       // some pixel of this lane's lattice is in the box
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(mine) != 0 && A.debug_skip != 1, 0)) {
#if RT_RASTER_CHECK_INTERVAL
        // every pixel of the lane's lattice that passes the test above must lie in the rectangle
        // the wave is about to work through (computed here the same way)
        {
            constexpr float COLS = (float)RT_RASTER_COLS;
            const int x_end = wave_x0 + L::WW < A.W ? wave_x0 + L::WW : A.W;
            const float half_w = (float)(A.W / 2);
            const float xl = __builtin_fmaxf(__builtin_fmaf(COLS, box.x - box.y, half_w), (float)wave_x0);
            const float xh = __builtin_fminf(__builtin_fmaf(COLS, box.x + box.y, half_w), (float)(x_end - 1));
            const int x_lo = (int)__builtin_ceilf(xl), x_hi = (int)__builtin_floorf(xh);
            const int px0 = wave_x0 + lane % L::SX, lr0 = wave_y0 + lane / L::SX;
            unsigned long long missed = 0;
#pragma unroll
            for (int c = 0; c < RX; ++c) {
                const int px = px0 + L::SX * c;
                if (__builtin_fabsf(d0 + KX * (float)c) <= box.y && px < A.W && !(xl <= xh && px >= x_lo && px <= x_hi)) ++missed;
            }
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int j = 0; j < RG; ++j) {
                    const int lrow = lane / L::SX + RT_TILE_H * g + L::SY * j;       // local row within the block
                    if (!(__builtin_fabsf(e[g] + KY * (float)j) <= box.w) || lr0 + RT_TILE_H * g + L::SY * j >= A.rows.rows) continue;
                    if (!(__builtin_fabsf(rows_y[lrow] - box.z) <= box.w)) ++missed;      // the rare path's own row test
                }
            if (missed && A.tests) atomicAdd(A.tests, missed << 48);
        }
#endif
        raster_rare_block<R>(A, box, tri, wave_x0, wave_y0, rows_y, row_list, lane);
    }
}

// The pixel rectangle of a wave, for the tile-culling variant: centre and half extents, x in
// units of RT_RASTER_COLS pixels like the boxes.  Exact in fp32.
struct WaveRect { float xs_c, xs_h, y_c, y_h; };

// Can any pixel of the rectangle lie in the triangle's box?  Interval overlap in x and y.
__device__ __forceinline__ bool rect_meets_box(const float4& box, const WaveRect& w)
{
    return __builtin_fabsf(w.xs_c - box.x) <= w.xs_h + box.y && __builtin_fabsf(w.y_c - box.z) <= w.y_h + box.w;
}

// Stages the bounding boxes of `count` (<= RT_CULL_GROUP) consecutive tiles from tile t on.
__device__ __forceinline__ void stage_boxes(const float4* __restrict__ boxes, int t, int count, float4* buf,
                                            int wave, int lane)
{
    const float4* src = boxes + (long)t * RT_TILE;
#pragma unroll
    for (int c = 0; c < (RT_CULL_GROUP * (RT_TILE / 64) + 3) / 4; ++c) {
        const int part = c * 4 + wave;
        if (part < count * (RT_TILE / 64))
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(src + part * 64 + lane),
                (__attribute__((address_space(3))) void*)&buf[part * 64], 16, 0, 0);
    }
}

// Only the boxes travel through LDS, RT_CULL_GROUP tiles at a time, double-buffered by LDS-DMA
// loads and read at wave-uniform addresses (broadcast, no bank conflicts).
// CULL: every lane first tests ONE triangle's box against the wave's whole pixel rectangle, and
// only the survivors meet the pixels one by one.  The candidates that reach the exact pass are
// the same either way; what changes is that most (pixel, triangle) pairs are then decided 64 R
// at a time instead of one by one.
// REG: the rows of the call's row set are regular (raster_triangle_reg).
template <int R, bool CULL, bool REG>
__global__ void __launch_bounds__(RT_WG, CULL ? RT_MIN_WAVES : (R >= 2048 && !REG ? RT_RASTER_WAVES - 2 : RT_RASTER_WAVES))
k_sweep_raster(const SweepArgs A)
{
    static_assert(!(CULL && REG), "the tile-culling variant keeps a coordinate per row");
    // (one more entry: the read-ahead of the all-pairs loop touches the one behind the second buffer)
    __shared__ float4 tile_lds[2 * RT_CULL_GROUP * RT_TILE + 1];
    auto tile = reinterpret_cast<float4 (*)[RT_CULL_GROUP * RT_TILE]>(tile_lds);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (scalar: triangle numbers stay out of the vector unit)

    // wave = (COLS R) x (64 / COLS) pixels; workgroup = its four waves side by side ((16 R) x 16
    // pixels), two by two or stacked.  (The squarer a wave's pixels, the fewer triangles have one
    // of them in their box: the rare path is paid per wave.)  A lane's R pixels are a lattice of
    // RY rows SY apart by RX columns SX apart, and the wave's SX x SY = 64 lanes are the
    // lattice's phases: neighbouring pixels belong to different lanes.
    constexpr int COLS = RT_RASTER_COLS;
    using L = RasterLane<R>;
    constexpr int RY = L::RY, SX = L::SX, SY = L::SY;
    constexpr int WW = L::WW, WH = L::WH;
    // the four waves: waves_x side by side, the rest below (whichever wastes fewer pixels beyond
    // the image's right and bottom edges)
    // ... or, waves_x = 0, the wave-sized blocks of the band numbered row by row and dealt four
    // to a workgroup: no idle waves but in the last workgroup
    // ... or, waves_x = -1, one wave-sized block per workgroup: its four waves own the same pixels
    // and sweep a quarter of every tile of triangles each (they share the pixels' state through
    // the same atomics as the workgroups of different triangle chunks do).  No idle waves at all,
    // which matters for a row part of a few hundred rows: 270 rows are five blocks of 64.
    const bool shared_block = A.waves_x < 0;
    int wave_x0, wave_y0;
    if (shared_block) {
        const int cols = (A.W + WW - 1) / WW;
        const int w = blockIdx.x, wy = w / cols;
        wave_x0 = (w - wy * cols) * WW;
        wave_y0 = wy * WH;
    } else if (A.waves_x == 0) {
        const int cols = (A.W + WW - 1) / WW;
        const int w = blockIdx.x * 4 + wave, wy = w / cols;
        wave_x0 = (w - wy * cols) * WW;
        wave_y0 = wy * WH;
    } else {
        const int wx = A.waves_x, tile_w = wx * WW, tile_h = WH * (4 / wx);
        const int tiles_x = (A.W + tile_w - 1) / tile_w;
        const int tyi = blockIdx.x / tiles_x, txi = blockIdx.x - tyi * tiles_x;
        wave_x0 = txi * tile_w + (wave & (wx - 1)) * WW;
        wave_y0 = tyi * tile_h + (wave / wx) * WH;
    }
    const int px0 = wave_x0 + lane % SX;
    const int lr0 = wave_y0 + lane / SX;                            // first local row of the call's row set
    // a wave that lies wholly beyond the image's right or bottom edge only helps to stage tiles
    const bool wave_on = __builtin_amdgcn_readfirstlane((wave_x0 < A.W && wave_y0 < A.rows.rows) ? 1 : 0) != 0;
    // driver.c:132 per local row of the wave, in LDS, for the rare path (which indexes it by a row
    // number of its own in every lane); NaN for a row beyond the band
    __shared__ float row_y[RT_WG / 64][WH];
    __shared__ int row_list[RT_WG / 64][REG ? WH : 1];     // the rare path's list of a box's rows (regular-row form)
    constexpr int NY = REG ? 1 : RY, NG = REG ? RasterGroups<R>::G : 1;
    float Y[NY];                                                    // every row of the lane's lattice, or ...
    float YG[NG];                                                   // ... the first one of every 16-row group
    if constexpr (REG) {
        for (int r = lane; r < WH; r += 64)
            row_y[wave][r] = wave_y0 + r < A.rows.rows ? (float)(A.rows.image_row(wave_y0 + r) - A.H / 2)
                                                       : __builtin_nanf("");
#pragma unroll
        for (int g = 0; g < NG; ++g)
            YG[g] = lr0 + RT_TILE_H * g < A.rows.rows ? (float)(A.rows.image_row(lr0 + RT_TILE_H * g) - A.H / 2)
                                                      : __builtin_nanf("");
    } else {
#pragma unroll
        for (int j = 0; j < RY; ++j)
            Y[j] = lr0 + SY * j < A.rows.rows ? (float)(A.rows.image_row(lr0 + SY * j) - A.H / 2) : __builtin_nanf("");
        // (the lanes of a wave that share a row write the same value)
#pragma unroll
        for (int j = 0; j < RY; ++j) row_y[wave][lane / SX + SY * j] = Y[j];
    }
    const float* y_of_row = &row_y[wave][lane / SX];
    // ray number: local pixel (the triangle chunks of a pixel share its count, list and bound)
    const long slot0 = (long)lr0 * A.W + px0;

    // column c of the lane has X = px0 + SX c - W/2 (driver.c:131); the planes and boxes are in
    // units of COLS pixels, so the sweep works with X / COLS = Xs0 + (SX / COLS) c (exact in fp32)
    const float Xs0 = (float)(px0 - A.W / 2) * (1.0f / COLS);
    WaveRect rect{};
    if constexpr (CULL && !REG) {
        // columns wave_x0 .. the wave's last one on the image; rows: the image rows of the wave's
        // local rows that exist (an interleaved row part maps them anywhere)
        const int x_last = wave_x0 + WW - 1 < A.W - 1 ? wave_x0 + WW - 1 : A.W - 1;
        const float xa = (float)(wave_x0 - A.W / 2), xb = (float)(x_last - A.W / 2);
        float ya = INF, yb = -INF;              // (the minimum and maximum pass over a missing row's NaN)
#pragma unroll
        for (int j = 0; j < RY; ++j) {
            ya = __builtin_fminf(ya, Y[j]);
            yb = __builtin_fmaxf(yb, Y[j]);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            ya = __builtin_fminf(ya, __shfl_xor(ya, off));
            yb = __builtin_fmaxf(yb, __shfl_xor(yb, off));
        }
        rect.xs_c = 0.5f * (xa + xb) * (1.0f / COLS);
        rect.xs_h = 0.5f * (xb - xa) * (1.0f / COLS);
        rect.y_c = 0.5f * (ya + yb);
        rect.y_h = 0.5f * (yb - ya);
    }

    // this workgroup's share of the triangle range: chunk blockIdx.y of gridDim.y
    const int t_first = A.tri_begin / RT_TILE, t_count = A.tri_end / RT_TILE - t_first;
    const int t_begin = t_first + (int)((long)t_count * blockIdx.y / gridDim.y);
    const int t_end = t_first + (int)((long)t_count * (blockIdx.y + 1) / gridDim.y);
    const float4* __restrict__ boxes = A.planes + (long)RT_APEX_PLANES * A.plane_stride;
    auto group = [&](int t) { return t_end - t < RT_CULL_GROUP ? t_end - t : RT_CULL_GROUP; };
    unsigned long long tiles_done = 0;
    unsigned kept = 0;              // tile culling: triangles that went on to the per-pixel tests
    if (t_begin < t_end) {
        stage_boxes(boxes, t_begin, group(t_begin), tile[0], wave, lane);
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
    }
    int round = 0;
    for (int t = t_begin; t < t_end; t += RT_CULL_GROUP, ++round) {
        const int buf = round & 1, count = group(t);
        if (t + RT_CULL_GROUP < t_end)
            stage_boxes(boxes, t + RT_CULL_GROUP, group(t + RT_CULL_GROUP), tile[buf ^ 1], wave, lane);
        const float4* __restrict__ base = tile[buf];
        const int tri0 = t * RT_TILE;
        if constexpr (CULL) {
            for (int h = 0; wave_on && h < count * (RT_TILE / 64); ++h) {
                unsigned long long keep = __ballot(rect_meets_box(base[h * 64 + lane], rect));
                kept += __popcll(keep);
                while (keep) {
                    const int j = h * 64 + __builtin_ctzll(keep);
                    keep &= keep - 1;
                    if constexpr (!REG) raster_triangle<R>(A, base[j], tri0 + j, Xs0, Y, y_of_row, px0, lr0, slot0);
                }
            }
        } else {
            // LDS reads are issued one triangle ahead by hand (the rare path splits the body into
            // basic blocks, which keeps the compiler from hoisting them)
            // this wave's triangles of a tile: all of them, or its quarter
            const int q0 = shared_block ? wave * (RT_TILE / 4) : 0, qn = shared_block ? RT_TILE / 4 : RT_TILE;
            for (int sub = 0; wave_on && sub < count; ++sub) {
                // (the address in a vector register the compiler cannot see through, advanced once
                // per kRasterUnroll triangles, the reads at immediate offsets: formed in scalar
                // registers it would be copied to a vector register before every read; the last
                // read of a tile or quarter touches the entry behind it and is not used)
                unsigned box_at = lds_address(base + sub * RT_TILE + q0);
                asm volatile("" : "+v"(box_at));
                float4 next = lds_read_float4(box_at);
                for (int j4 = q0; j4 < q0 + qn; j4 += kRasterUnroll) {
#pragma unroll
                    for (int k = 0; k < kRasterUnroll; ++k) {
                        const float4 box = next;
                        next = lds_read_float4(box_at + 16u * (unsigned)(k + 1));
                        if constexpr (REG)
                            raster_triangle_reg<R>(A, box, tri0 + sub * RT_TILE + j4 + k, Xs0, YG, row_y[wave], row_list[wave], wave_x0, wave_y0, lane);
                        else
                            raster_triangle<R>(A, box, tri0 + sub * RT_TILE + j4 + k, Xs0, Y, y_of_row, px0, lr0, slot0);
                    }
                    box_at += 16u * kRasterUnroll;
                }
            }
        }
        tiles_done += count;
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
    }

    // pairs tested one by one: all of them, or with tile culling those of the surviving triangles
    if (A.tests && lane == 0 && wave_on)
        atomicAdd(A.tests, (CULL ? (unsigned long long)kept
                                 : tiles_done * (unsigned long long)(shared_block ? RT_TILE / 4 : RT_TILE)) * (64 * R));
}

// ------------------------------------------------------------------------------------------
// primary rays, wave-block form
// ------------------------------------------------------------------------------------------
// In the lattice form above with 64 rows and more per lane, every lane of a wave has the same
// rows (its 64 lanes are 64 column phases): the wave's pixels are the product of its columns and
// its rows, and some pixel of the WAVE lies in a triangle's box exactly when some column lies in
// the box's x extent and some row in its y extent.  So the row half of the test need not be
// repeated in every lane: this form deals the block's rows to the lanes as it deals the columns
// (lane l: columns l, l + 64, ... of 4096; local rows l, l + 64, ... of 64 NL), every lane tests
// its own, and two ballots combine them -- 64 + NL additions and 32 + NL / 2 minima per wave and
// triangle for 4096 x 64 NL pixels (104 instructions for 4096 x 256, where a lattice of 256 rows in
// every lane took 482).  Every column and every row is tested against every triangle, and a
// pixel's verdict is its column's and its row's, as before.
//
// A workgroup owns one block; its four waves sweep a quarter of every staged tile each and note
// which of the quarter's 32 triangles passed.  Behind that, the wave works through the passing
// triangles' pixel rectangles together (raster_rare_batch): lane k takes triangle k's box, finds
// its columns (an interval, from the box) and its rows (the block's rows are in ascending order
// of Y: two binary searches with the hot loop's own comparison) and fetches its planes into LDS
// -- one wait for memory per quarter tile -- and then the rows of all the rectangles are dealt to the
// lanes, each cut down to the span its triangle can occupy, and the pixels of the spans dealt to
// the lanes for the inside test of finish_raster (raster_rare_batch has the details).
constexpr int kWaveBlockCols = 4096;
constexpr int kWaveGroup = 2;           // tiles of boxes staged at a time

template <int NL>
__device__ __forceinline__ void raster_rare_batch(const SweepArgs& A, const float4* qbox, int tri0, unsigned passed,
                                                  int wave_x0, int wave_y0, const float* rows_y, float4 (*qp)[4],
                                                  unsigned qp_at, int* flags, float Xs0, const float (&YL)[NL], int lane)
{
    constexpr int WW = kWaveBlockCols, WH = 64 * NL;
    constexpr float COLS = (float)RT_RASTER_COLS;
    int x_lo = 0, nx = 0, r_lo = 0, ny = 0;
    const bool mine = lane < 32 && ((passed >> lane) & 1u);
    if (mine) {
        const float4 box = qbox[lane];
        // columns: pixels px with |(px - W/2) / COLS - cx| <= rx inside the block and the image.  The
        // bounds are rounded inwards to whole pixels: the box was grown by 0.01 pixel and more for
        // the roundings of a test on it, a hundred times what the bounds' own roundings come to.
        const int x_end = wave_x0 + WW < A.W ? wave_x0 + WW : A.W;
        const float half_w = (float)(A.W / 2);
        const float xl = __builtin_fmaxf(__builtin_fmaf(COLS, box.x - box.y, half_w), (float)wave_x0);
        const float xh = __builtin_fminf(__builtin_fmaf(COLS, box.x + box.y, half_w), (float)(x_end - 1));
        if (xl <= xh) {
            x_lo = (int)__builtin_ceilf(xl);
            nx = (int)__builtin_floorf(xh) - x_lo + 1;
        }
        // rows: the local rows r with |Y_r - cy| <= ry, i.e. -ry <= Y_r - cy <= ry in the hot loop's
        // own arithmetic.  Y ascends with r (rows beyond the band: NaN, at the end), and so does the
        // rounded difference, so they are the rows from the first one that is not below the extent
        // to the first one above it.
        int lo = 0, hi = WH;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (rows_y[mid] - box.z < -box.w) lo = mid + 1;
            else hi = mid;
        }
        r_lo = lo;
        hi = WH;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (rows_y[mid] - box.z <= box.w) lo = mid + 1;
            else hi = mid;
        }
        ny = lo - r_lo;
    }
#if RT_RASTER_CHECK_INTERVAL
    // every column and row of the wave that passes the hot loop's test for one of these triangles
    // must lie in the rectangle found here
    {
        unsigned long long missed = 0;
        for (unsigned m = passed; m; m &= m - 1) {
            const int k = __builtin_ctz(m);
            const float4 box = qbox[k];
            const int kx = __shfl(x_lo, k), knx = __shfl(nx, k), kr = __shfl(r_lo, k), kny = __shfl(ny, k);
            const float d0 = Xs0 - box.x;
#pragma unroll
            for (int c = 0; c < 64; ++c) {
                const int px = wave_x0 + lane + 64 * c;
                if (__builtin_fabsf(d0 + (64.0f / COLS) * (float)c) <= box.y && px < A.W && !(px >= kx && px < kx + knx)) ++missed;
            }
#pragma unroll
            for (int j = 0; j < NL; ++j) {
                const int lrow = lane + 64 * j;
                if (__builtin_fabsf(YL[j] - box.z) <= box.w && wave_y0 + lrow < A.rows.rows && !(lrow >= kr && lrow < kr + kny)) ++missed;
            }
        }
        if (missed && A.tests) atomicAdd(A.tests, missed << 48);
    }
#endif
    if (nx <= 0 || ny <= 0) nx = ny = 0;
    const int npix = nx * ny;
    unsigned long long todo = __builtin_amdgcn_ballot_w64(npix > 0);
    if (todo == 0) return;
    // the planes of the triangles that have pixels here (a part of interleaved row blocks sees every
    // triangle of the image in its block of rows, and most of them between its rows)
    if (npix > 0) {
        const float4* __restrict__ g4 = A.planes + (tri0 + lane);
        qp[lane][0] = g4[0];
        qp[lane][1] = g4[A.plane_stride];           // (.w: the margin)
        qp[lane][2] = g4[2 * A.plane_stride];
        qp[lane][3] = g4[3 * A.plane_stride];
    }
    wave_lds_sync();
    // A bounding box is mostly empty -- of the pixels of the million-triangle frame's rectangles one
    // in eighteen is a candidate -- so the rows of a rectangle are first cut down to the span the
    // triangle can occupy: U, V, det and W = det - U - V are linear in a pixel's X for a given row, a
    // hit needs U, V, W >= 0 and det > 0, and the sweep's own allowances say how far the fp32 values
    // of a true hit can lie below that (by at most delta = margin / 32, and en for det).  The span
    // of a row keeps every X with U, V, W > -margin / 2 and det > -2 en -- four half-lines, so one
    // interval -- rounded outwards and widened by a pixel on either side; a pixel outside it cannot
    // be a hit, and the pixels inside it get the inside test of finish_raster as before.
    // The rows of ALL the rectangles are dealt to the lanes 64 at a time (a lane looks up its
    // rectangle's corner and planes through the lane that owns it), each lane works out its row's
    // span, and the pixels of those 64 spans are dealt to the lanes 64 at a time in turn.  Small
    // triangles, a few rows of a few pixels each, fill the lanes that way; one rectangle after the
    // other, each of them cost a round of its own for its rows and one for its pixels, and a wait
    // for memory in each.  A.debug_skip == 4: whole rows (for the tests).
    if (A.debug_skip == 2) return;
    const float half_w = (float)(A.W / 2);
    // (plane j of rectangle k, by its LDS address qp_at + ...: the rectangle differs from lane to lane)
    auto plane = [&](int k, int j) { return lds_read_float4(qp_at + (unsigned)(k * 64 + j * 16)); };
    const int rows_k = npix > 0 ? ny : 0;
    const int inclR = wave_scan(rows_k, lane);
    const int totalR = __builtin_amdgcn_readlane(inclR, 63);
    const int exclR = inclR - rows_k;
    int carryR = -1;
    for (int R0 = 0; R0 < totalR; R0 += 64) {
        const int k = wave_owner(exclR, rows_k, R0, flags, carryR, lane);    // the rectangle of row item R0 + lane
        carryR = __builtin_amdgcn_readlane(k, 63);
        const int kx = __shfl(x_lo, k), knx = __shfl(nx, k);
        const int row = __shfl(r_lo - exclR, k) + R0 + lane;                 // its row of the block
        int c_lo = kx, n_r = 0;
        if (R0 + lane < totalR) {
            int c_hi = kx + knx - 1;
            const float margin = plane(k, 1).w;
            if (margin != INF && A.debug_skip != 4) {       // (INF: no fp32 verdict, every pixel of the box is listed)
                const float4 kP = plane(k, 0), kH = plane(k, 2), kN = plane(k, 3);
                const float Y = rows_y[row];
                // f(Xs) = g Xs + c for U, V, det and W (Xs = X / COLS as in finish_raster)
                const float gU = -kP.x, cU = -__builtin_fmaf(Y, kP.y, kP.z) - kP.w;
                const float gV = COLS * kH.x, cV = __builtin_fmaf(Y, kH.y, kH.z);
                const float gD = COLS * kN.x, cD = __builtin_fmaf(Y, kN.y, kN.z);
                const float gW = gD - gU - gV, cW = cD - cU - cV;
                float lo = -INF, hi = INF;
                bool none = false;
                auto keep = [&](float g, float c, float least) {     // the X with g X + c > least
                    const float bound = (least - c) / g;
                    if (g > 0.0f) lo = __builtin_fmaxf(lo, bound);
                    else if (g < 0.0f) hi = __builtin_fminf(hi, bound);
                    else if (!(c > least)) none = true;
                };
                const float least = -0.5f * margin;
                keep(gU, cU, least);
                keep(gV, cV, least);
                keep(gW, cW, least);
                keep(gD, cD, -2.0f * kN.w);
                // to pixel columns, outwards, and a pixel more on either side
                const float pl = __builtin_fmaf(COLS, lo, half_w), ph = __builtin_fmaf(COLS, hi, half_w);
                if (none || !(pl <= ph)) {
                    c_hi = c_lo - 1;
                } else {
                    if (pl > (float)c_lo) c_lo = (int)__builtin_floorf(pl) - 1 > c_lo ? (int)__builtin_floorf(pl) - 1 : c_lo;
                    if (ph < (float)c_hi) c_hi = (int)__builtin_ceilf(ph) + 1 < c_hi ? (int)__builtin_ceilf(ph) + 1 : c_hi;
                }
            }
            n_r = c_hi - c_lo + 1;
            if (n_r < 0) n_r = 0;
        }
        // the pixels of the batch's spans, dealt to the lanes 64 at a time (a running sum over the
        // rows' span lengths; dealing L lanes to a row, L the power of two that holds the longest
        // span, was measured: one long span idles most lanes, 2.2 ms against 1.6)
        const int incl = wave_scan(n_r, lane);
        const int total = __builtin_amdgcn_readlane(incl, 63);
        const int excl = incl - n_r;
        const int base = c_lo - excl;                   // pixel p of the row's span is column base + p
        const int row_k = (row << 5) | k;               // (k < 32)
        int carry = -1;
        for (int p0 = 0; p0 < total; p0 += 64) {
            const int p = p0 + lane;
            const int q = wave_owner(excl, n_r, p0, flags, carry, lane);     // the lane whose row's span holds pixel p
            carry = __builtin_amdgcn_readlane(q, 63);
            const int px = __shfl(base, q) + p;
            const int rk = __shfl(row_k, q);
            if (p < total) {
                const int prow = rk >> 5, pk = rk & 31;
                finish_raster(A, (long)(wave_y0 + prow) * A.W + px, tri0 + pk, plane(pk, 0), plane(pk, 1).w, plane(pk, 2), plane(pk, 3),
                              (float)(px - A.W / 2) * (1.0f / COLS), rows_y[prow]);
            }
        }
    }
}

template <int NL>
__global__ void __launch_bounds__(RT_WG, RT_RASTER_WAVES)
k_sweep_raster_wave(const SweepArgs A)
{
    constexpr int WW = kWaveBlockCols, WH = 64 * NL;
    constexpr float COLS = (float)RT_RASTER_COLS, KX = 64.0f / COLS;
    // (one more entry: the read-ahead touches the one behind the second buffer)
    __shared__ float4 tile_lds[2 * kWaveGroup * RT_TILE + 1];
    auto tile = reinterpret_cast<float4 (*)[kWaveGroup * RT_TILE]>(tile_lds);
    __shared__ float row_y[WH];                                 // driver.c:132 per local row of the block, NaN beyond the band
    __shared__ float4 plane_lds[RT_WG / 64][RT_TILE / 4][4];    // the rare path's planes of a quarter tile, per wave
    __shared__ int owner_lds[RT_WG / 64][64];                   // ... and its scratch for dealing pixels to lanes

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int cols = (A.W + WW - 1) / WW;
    const int wy = (int)blockIdx.x / cols;
    const int wave_x0 = ((int)blockIdx.x - wy * cols) * WW, wave_y0 = wy * WH;
    auto y_of = [&](int r) {
        return wave_y0 + r < A.rows.rows ? (float)(A.rows.image_row(wave_y0 + r) - A.H / 2) : __builtin_nanf("");
    };
    for (int r = tid; r < WH; r += RT_WG) row_y[r] = y_of(r);
    float YL[NL];
#pragma unroll
    for (int j = 0; j < NL; ++j) YL[j] = y_of(lane + 64 * j);
    // column c of the lane has X = wave_x0 + lane + 64 c - W/2 (driver.c:131); the boxes are in units
    // of COLS pixels, so the sweep works with X / COLS = Xs0 + (64 / COLS) c (exact in fp32)
    const float Xs0 = (float)(wave_x0 + lane - A.W / 2) * (1.0f / COLS);

    // this workgroup's share of the triangle range: chunk blockIdx.y of gridDim.y
    const int t_first = A.tri_begin / RT_TILE, t_count = A.tri_end / RT_TILE - t_first;
    const int t_begin = t_first + (int)((long)t_count * blockIdx.y / gridDim.y);
    const int t_end = t_first + (int)((long)t_count * (blockIdx.y + 1) / gridDim.y);
    const float4* __restrict__ boxes = A.planes + (long)RT_APEX_PLANES * A.plane_stride;
    auto group = [&](int t) { return t_end - t < kWaveGroup ? t_end - t : kWaveGroup; };
    auto stage = [&](int t, int count, float4* buf) {
        const float4* src = boxes + (long)t * RT_TILE;
        // (RT_TILE / 64 parts of 64 boxes per tile, dealt to the four waves)
#pragma unroll
        for (int c = 0; c < (kWaveGroup * (RT_TILE / 64) + 3) / 4; ++c) {
            const int part = c * 4 + wave;
            if (part < count * (RT_TILE / 64))
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(src + part * 64 + lane),
                    (__attribute__((address_space(3))) void*)&buf[part * 64], 16, 0, 0);
        }
    };
    unsigned long long tiles_done = 0;
    if (t_begin < t_end) stage(t_begin, group(t_begin), tile[0]);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();            // (the staged boxes and row_y)
    int round = 0;
    for (int t = t_begin; t < t_end; t += kWaveGroup, ++round) {
        const int buf = round & 1, count = group(t);
        if (t + kWaveGroup < t_end) stage(t + kWaveGroup, group(t + kWaveGroup), tile[buf ^ 1]);
        const float4* __restrict__ base = tile[buf];
        for (int sub = 0; sub < count; ++sub) {
            // this wave's quarter of the tile (the address in a vector register the compiler cannot see
            // through, advanced once per kRasterUnroll triangles, the reads at immediate offsets, one
            // triangle ahead; the last read of a quarter touches the entry behind it and is not used)
            const float4* qbox = base + sub * RT_TILE + wave * (RT_TILE / 4);
            const int tri0 = (t + sub) * RT_TILE + wave * (RT_TILE / 4);
            unsigned box_at = lds_address(qbox);
            asm volatile("" : "+v"(box_at));
            float4 next = lds_read_float4(box_at);
            unsigned passed = 0;
            for (int j4 = 0; j4 < RT_TILE / 4; j4 += kRasterUnroll) {
#pragma unroll
                for (int k = 0; k < kRasterUnroll; ++k) {
                    const float4 box = next;
                    next = lds_read_float4(box_at + 16u * (unsigned)(k + 1));
                    const float d0 = Xs0 - box.x;
                    float accx = INF;
#pragma unroll
                    for (int c = 0; c + 1 < 64; c += 2)     // (d0 + 0.0f is not d0 to the compiler: it would keep the addition)
                        accx = min3_abs(accx, c == 0 ? d0 : d0 + KX * (float)c, d0 + KX * (float)(c + 1));
                    float accy = NL == 1 ? __builtin_fabsf(YL[0] - box.z) : INF;
#pragma unroll
                    for (int j = 0; j + 1 < NL; j += 2) accy = min3_abs(accy, YL[j] - box.z, YL[j + 1] - box.z);
                    const bool in = __builtin_amdgcn_ballot_w64(accx <= box.y) != 0 &&
                                    __builtin_amdgcn_ballot_w64(accy <= box.w) != 0;
                    if (in) passed |= 1u << (j4 + k);
                }
                box_at += 16u * kRasterUnroll;
            }
            if (__builtin_expect(passed != 0 && A.debug_skip != 1, 0))
                raster_rare_batch<NL>(A, qbox, tri0, passed, wave_x0, wave_y0, row_y, plane_lds[wave],
                                      lds_address(&plane_lds[0][0][0]) + (unsigned)wave * (unsigned)sizeof(plane_lds[0]),
                                      owner_lds[wave], Xs0, YL, lane);
        }
        tiles_done += count;
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
    }
    // pairs tested: this wave's quarter of every tile against the block's pixels
    if (A.tests && lane == 0)
        atomicAdd(A.tests, tiles_done * (unsigned long long)(RT_TILE / 4) * ((unsigned long long)WW * WH));
}

// ------------------------------------------------------------------------------------------
// any-hit sweep for rays that end at the apex (shadow rays of one light)
// ------------------------------------------------------------------------------------------
// The bounding-square and inside tests do not change when the direction is scaled by any non-zero
// number, so every ray uses D' = D / D_k with k its dominant axis: D'_k = 1 exactly and the
// other two components, (u, v), are at most 1 in magnitude.  Rays are binned by k beforehand
// (k_bin_rays), a workgroup handles rays of one k, and a dot product D'.p costs two FMAs
// instead of a multiply and two:  U = fma(u, p_i, fma(v, p_j, p_k)).
struct ApexRay {
    float u, v;         // the two minor components of D'; NaN once the ray has retired
};

// What only the rare path needs of a ray is not kept in registers (it would cost the hot loop a
// wave of occupancy) but left in memory by the kernel's prologue (SweepArgs::tails): its number,
//   len = |apex - S|, the ray parameter of the apex along the unit direction, and
//   sig = 1 / (unit direction)_k, so that D' = sig * unit direction.
template <int AX>
__device__ __forceinline__ float dot_ax(float u, float v, const float4& p)
{
    if (AX == 0) return __builtin_fmaf(u, p.y, __builtin_fmaf(v, p.z, p.x));     // D' = (1, u, v)
    if (AX == 1) return __builtin_fmaf(u, p.x, __builtin_fmaf(v, p.z, p.y));     // D' = (u, 1, v)
    return __builtin_fmaf(u, p.x, __builtin_fmaf(v, p.y, p.z));                  // D' = (u, v, 1)
}

// The rest of the inside test for one ray (u, v = minor components of D') that passed the slab
// test of triangle `tri`; Hh, N = the triangle's planes 2 and 3.  Returns true when the ray
// retires.  What only this path needs of a ray -- its number, len and sig -- is read again from
// memory rather than kept in registers: one 16-byte entry of A.tails, written by the prologue of
// k_sweep_apex (read through the ray list and recomputed from the ray's end points here, it was
// two dependent fetches and an fp64 square root and division for every ray that passed a
// triangle's inside test).
template <int AX>
__device__ __forceinline__ bool finish_apex(const SweepArgs& A, float u, float v, long jr, int tri,
                                            const Tri2& q, const float4& Hh, const float4& N)
{
    const float margin = q.Q.w;
    if (margin != INF) {            // INF: nearly degenerate triangle, no fp32 verdict
        const float U = dot_ax<AX>(u, v, q.P);
        const float V = dot_ax<AX>(u, v, Hh);
        const float d = dot_ax<AX>(u, v, N);
        const float a = U * d, b = V * d;
        const float w = __builtin_fmaf(d, d, -a) - b;
        const float m3 = __builtin_fminf(__builtin_fminf(a, b), w);
        if (!(__builtin_fmaf(margin, __builtin_fabsf(d), m3) > 0.0f)) return false;
        // (the ray's number, len and sig: from the entry the kernel's prologue left for this ray)
        const float4 tl = A.tails[jr];
        const int sl = __float_as_int(tl.z);
        const float len = tl.x, sig = tl.y;
        // The triangle's plane meets the line apex + tau * D' at tau = k / (D'.nn).  Along the
        // unit direction, from S: t = len + sig * tau (the reference's t times len; only its
        // sign matters for an any-hit query, triangle_intersection.cu:169).
        const float T = sig * Hh.w;
        // `tri` is the entry of the plane arrays; lists and the occluder cache hold triangle numbers
        return candidate_shared(A.cnt + sl, A.list + (long)sl * RT_LIST_K,
                                A.occ_id ? A.occ_id + sl : nullptr, A.tri_order ? A.tri_order[tri] : tri,
                                a, b, w, d, T, len, margin, N.w, U22 * __builtin_fabsf(T));
    }
    const int sl = __float_as_int(A.tails[jr].z);
    const int at = atomicAdd(A.cnt + sl, 1) & RT_CNT_MASK;
    if (at < RT_LIST_K) A.list[(long)sl * RT_LIST_K + at] = A.tri_order ? A.tri_order[tri] : tri;
    return false;
}

// The tile loop of one workgroup for rays of dominant axis AX.  A ray is retired when its u is
// NaN: its slab value is NaN too and never passes.
//
// Shadow rays that survive a few ranges are scattered over the image, so for almost every
// triangle SOME ray of a wave lies in its slab; what is rare is that one lane has more than one
// such ray.  The slab passes are therefore handled lane-wise: every lane picks its first passing
// ray (a chain of selects), the wave finishes those together, and the loop repeats only while
// some lane has another one.
// The (u, v) bounding box of a wave's live rays, for the tile-culling variant.
struct WaveBox { float u_c, u_h, v_c, v_h; };

template <int R>
__device__ __forceinline__ WaveBox wave_box(const ApexRay (&ray)[R])
{
    // retired and unused rays carry NaN, which fmin / fmax pass over
    float u0 = INF, u1 = -INF, v0 = INF, v1 = -INF;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        u0 = __builtin_fminf(u0, ray[r].u);
        u1 = __builtin_fmaxf(u1, ray[r].u);
        v0 = __builtin_fminf(v0, ray[r].u == ray[r].u ? ray[r].v : INF);
        v1 = __builtin_fmaxf(v1, ray[r].u == ray[r].u ? ray[r].v : -INF);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        u0 = __builtin_fminf(u0, __shfl_xor(u0, off));
        u1 = __builtin_fmaxf(u1, __shfl_xor(u1, off));
        v0 = __builtin_fminf(v0, __shfl_xor(v0, off));
        v1 = __builtin_fmaxf(v1, __shfl_xor(v1, off));
    }
    // (a wave without live rays gets infinite centres and NaN extents: every comparison is
    // false and nothing is kept, but then no ray could pass a test either)
    return {0.5f * (u0 + u1), 0.5f * (u1 - u0), 0.5f * (v0 + v1), 0.5f * (v1 - v0)};
}

// Can any ray of the box lie in the triangle's bounding square sq = (cu, cv, rho, -)?  Interval
// overlap in u and in v; the box is the exact hull of the rays' fp32 (u, v) and the few
// roundings here are covered by the 1e-6 (the square's own allowance covers the per-ray test).
__device__ __forceinline__ bool box_meets_square(const float4& sq, const WaveBox& b)
{
    const float reach_u = b.u_h + sq.z + 1e-6f, reach_v = b.v_h + sq.z + 1e-6f;
    return __builtin_fabsf(b.u_c - sq.x) <= reach_u && __builtin_fabsf(b.v_c - sq.y) <= reach_v;
}

// Stages the planes of tile t of a light for rays of dominant axis AX: the four apex planes and
// the bounding squares of that axis (plane RT_APEX_PLANES + AX of the light's set) as plane 4.
template <int AX>
__device__ __forceinline__ void stage_light_tile(const float4* __restrict__ planes, long plane_stride, int t,
                                                 float4 (*buf)[RT_TILE], int wave, int lane)
{
    constexpr int NP = RT_APEX_PLANES + 1;
#pragma unroll
    for (int c = 0; c < (NP * (RT_TILE / 64) + 3) / 4; ++c) {
        const int chunk = c * 4 + wave;                 // chunk = plane * (RT_TILE/64) + part
        if (chunk < NP * (RT_TILE / 64)) {
            const int plane = chunk / (RT_TILE / 64), part = chunk % (RT_TILE / 64);
            const int src = plane < RT_APEX_PLANES ? plane : RT_APEX_PLANES + AX;
            const long g = (long)src * plane_stride + (long)t * RT_TILE + part * 64 + lane;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(planes + g),
                (__attribute__((address_space(3))) void*)&buf[plane][part * 64], 16, 0, 0);
        }
    }
}

// Stages the bounding squares of `count` (<= RT_CULL_GROUP) consecutive tiles from tile t on.
template <int AX>
__device__ __forceinline__ void stage_squares(const float4* __restrict__ planes, long plane_stride, int t,
                                              int count, float4* buf, int wave, int lane)
{
    const float4* src = planes + (long)(RT_APEX_PLANES + AX) * plane_stride + (long)t * RT_TILE;
#pragma unroll
    for (int c = 0; c < (RT_CULL_GROUP * (RT_TILE / 64) + 3) / 4; ++c) {
        const int part = c * 4 + wave;
        if (part < count * (RT_TILE / 64))
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(src + part * 64 + lane),
                (__attribute__((address_space(3))) void*)&buf[part * 64], 16, 0, 0);
    }
}

// One triangle against the lane's R rays.  A ray can only hit if its direction (u, v) lies
// inside the triangle's bounding square on the plane of the dominant axis (k_pack_apex),
//     max(|u - cu|, |v - cv|) <= rho.
// First test of every pair: the u half of that, |u - cu| <= rho -- one subtraction per pair, a
// three-way minimum per two pairs and one comparison per triangle: 13 VALU instructions per 8
// pairs.  The rays of a wave are a narrow bundle, so few triangles have any of them in their
// u strip; for those the wave completes the square test, and a lane with a ray inside the
// square goes on to the full inside test with its allowances (finish_apex), from the
// triangle's four apex planes P, Q, Hh, N, which `planes` fetches (from the LDS tile, or from
// memory).
template <int AX, int R, class Planes>
__device__ __forceinline__ void apex_triangle(const SweepArgs& A, ApexRay (&ray)[R], long j0, const float4& sq,
                                              int tri, Planes&& planes)
{
    const float qnan = __builtin_nanf("");
    float du[R];
#pragma unroll
    for (int r = 0; r < R; ++r) du[r] = ray[r].u - sq.x;
    // (a retired ray's NaN is passed over by the minimum)
    float mn = INF;
    if (R == 1) mn = __builtin_fabsf(du[0]);
#pragma unroll
    for (int r = 0; r + 1 < R; r += 2) mn = min3_abs(mn, du[r], du[r + 1]);
    if (__builtin_expect(mn <= sq.z, 0)) {
        // second level, for the lanes with a ray in the u strip: the square test completed the
        // same way (a subtraction and a maximum per ray, a three-way minimum per two rays, one
        // comparison); which rays are inside is only worked out when one is
        float m[R];
#pragma unroll
        for (int r = 0; r < R; ++r) m[r] = max_abs(du[r], ray[r].v - sq.y);
        float mn2 = INF;
        if (R == 1) mn2 = m[0];
#pragma unroll
        for (int r = 0; r + 1 < R; r += 2) mn2 = min3_plain(mn2, m[r], m[r + 1]);
        if (mn2 <= sq.z) {
            unsigned pass = 0;
#pragma unroll
            for (int r = 0; r < R; ++r) pass |= (m[r] <= sq.z ? 1u : 0u) << r;
            Tri2 cur;
            float4 Hh, N;
            planes(cur, Hh, N);
            while (pass) {
                const int rs = __builtin_ctz(pass);
                pass &= pass - 1;
                float u = ray[0].u, v = ray[0].v;
#pragma unroll
                for (int r = 1; r < R; ++r) {
                    u = rs == r ? ray[r].u : u;
                    v = rs == r ? ray[r].v : v;
                }
                if (finish_apex<AX>(A, u, v, j0 + (long)rs * 64, tri, cur, Hh, N)) {
                    // (both components: the maximum above passes over a single NaN)
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        ray[r].u = rs == r ? qnan : ray[r].u;
                        ray[r].v = rs == r ? qnan : ray[r].v;
                    }
                }
            }
        }
    }
}

// After a tile: retire the workgroup once every ray in it is certainly occluded or unused; with
// several chunks per ray the flag another chunk's workgroup may have set is polled every few
// tiles (two dependent loads per ray: not free).  Returns true when the workgroup is done.
template <int R>
__device__ __forceinline__ bool apex_all_retired(const SweepArgs& A, ApexRay (&ray)[R], long j0, bool poll)
{
    const float qnan = __builtin_nanf("");
    bool done = true;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (poll && ray[r].u == ray[r].u &&
            (__hip_atomic_load(&A.cnt[A.active[j0 + r * 64]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) &
             RT_CNT_OCCLUDED))
            ray[r].u = ray[r].v = qnan;
        done = done && !(ray[r].u == ray[r].u);
    }
    return __syncthreads_and(done);
}

// The tile loop of one workgroup for rays of dominant axis AX, every pair tested.  A ray is
// retired when its u is NaN: its distance from any square is NaN too and never passes.
//
// Shadow rays that survive a few ranges are scattered over the image, so what is rare is that
// one lane has more than one ray inside a triangle's square.  The passes are therefore handled
// lane-wise: every lane picks its first passing ray (a chain of selects), the wave finishes
// those together, and the loop repeats only while some lane has another one.
template <int AX, int R>
__device__ __forceinline__ void apex_tiles(const SweepArgs& A, float4 (*tile)[RT_APEX_PLANES + 1][RT_TILE],
                                           ApexRay (&ray)[R], long j0, int wave, int lane,
                                           unsigned long long& tiles_done)
{
    // this workgroup's share of the triangle range: chunk blockIdx.z of gridDim.z
    const int t_first = A.tri_begin / RT_TILE, t_count = A.tri_end / RT_TILE - t_first;
    const int t_begin = t_first + (int)((long)t_count * blockIdx.z / gridDim.z);
    const int t_end = t_first + (int)((long)t_count * (blockIdx.z + 1) / gridDim.z);
    if (t_begin < t_end) {
        stage_light_tile<AX>(A.planes, A.plane_stride, t_begin, tile[0], wave, lane);
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
    }
    for (int t = t_begin; t < t_end; ++t) {
        const int buf = (t - t_begin) & 1;
        if (t + 1 < t_end) stage_light_tile<AX>(A.planes, A.plane_stride, t + 1, tile[buf ^ 1], wave, lane);
        const float4* __restrict__ base = tile[buf][0];
        const int tri0 = t * RT_TILE;
        // The squares are read at wave-uniform addresses, which the compiler would form in scalar
        // registers and copy to a vector register before every read (one VALU instruction per
        // triangle beside the loop's 13).  Held in a vector register the compiler cannot see
        // through, the address advances once per kApexUnroll triangles and the reads use
        // immediate offsets.  Reads are issued one triangle ahead by hand (two ahead was
        // measured: no gain, more spills); the last one of a tile reads the entry behind the
        // plane (the next buffer's first, or the padding behind the tile buffers) and is not
        // used.
        unsigned sq_at = lds_address(tile[buf][RT_APEX_PLANES]);
        asm volatile("" : "+v"(sq_at));
        float4 next = lds_read_float4(sq_at);
        for (int j4 = 0; j4 < RT_TILE; j4 += kApexUnroll) {
#pragma unroll
            for (int k = 0; k < kApexUnroll; ++k) {
                const int j = j4 + k;
                const float4 sq = next;
                next = lds_read_float4(sq_at + 16u * (unsigned)(k + 1));
                apex_triangle<AX, R>(A, ray, j0, sq, tri0 + j, [&](Tri2& cur, float4& Hh, float4& N) {
                    cur = load_tri2(base + j);
                    Hh = base[j + 2 * RT_TILE];
                    N = base[j + 3 * RT_TILE];
                });
                // (the record's unused fourth component stays allocated until here: its register,
                // reused as a temporary, would have to wait for the read-ahead to land first)
                asm volatile("" : : "v"(sq.w));
            }
            sq_at += 16u * kApexUnroll;
        }
        ++tiles_done;
        __builtin_amdgcn_s_waitcnt(0);
        const bool poll = gridDim.z > 1 && ((t - t_begin) & (RT_APEX_POLL - 1)) == RT_APEX_POLL - 1;
        if (apex_all_retired<R>(A, ray, j0, poll)) break;
    }
}

// The same with tile culling: only the bounding squares travel through LDS, RT_CULL_GROUP tiles
// at a time.  Every lane first tests one triangle's square against the bounding box of the
// wave's whole bundle of rays (box_meets_square); the survivors meet the rays one by one, and a
// lane with a ray inside the square fetches the triangle's planes from memory.
template <int AX, int R>
__device__ __forceinline__ void apex_tiles_culled(const SweepArgs& A, float4 (*tile)[RT_CULL_GROUP * RT_TILE],
                                                  ApexRay (&ray)[R], long j0, int wave, int lane,
                                                  unsigned& kept)
{
    WaveBox box = wave_box<R>(ray);
    const int t_first = A.tri_begin / RT_TILE, t_count = A.tri_end / RT_TILE - t_first;
    const int t_begin = t_first + (int)((long)t_count * blockIdx.z / gridDim.z);
    const int t_end = t_first + (int)((long)t_count * (blockIdx.z + 1) / gridDim.z);
    auto group = [&](int t) { return t_end - t < RT_CULL_GROUP ? t_end - t : RT_CULL_GROUP; };
    if (t_begin < t_end) {
        stage_squares<AX>(A.planes, A.plane_stride, t_begin, group(t_begin), tile[0], wave, lane);
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
    }
    int round = 0;
    for (int t = t_begin; t < t_end; t += RT_CULL_GROUP, ++round) {
        const int buf = round & 1, count = group(t);
        if (t + RT_CULL_GROUP < t_end)
            stage_squares<AX>(A.planes, A.plane_stride, t + RT_CULL_GROUP, group(t + RT_CULL_GROUP), tile[buf ^ 1],
                              wave, lane);
        const float4* __restrict__ squares = tile[buf];
        if ((round & 3) == 3) box = wave_box<R>(ray);      // the bundle shrinks as rays retire
        for (int h = 0; h < count * (RT_TILE / 64); ++h) {
            unsigned long long keep = __ballot(box_meets_square(squares[h * 64 + lane], box));
            kept += __popcll(keep);
            while (keep) {
                const int j = h * 64 + __builtin_ctzll(keep);
                keep &= keep - 1;
                const long tri = (long)t * RT_TILE + j;
                apex_triangle<AX, R>(A, ray, j0, squares[j], (int)tri, [&](Tri2& cur, float4& Hh, float4& N) {
                    const float4* __restrict__ g = A.planes + tri;
                    cur.P = g[0];
                    cur.Q = g[A.plane_stride];
                    Hh = g[2 * A.plane_stride];
                    N = g[3 * A.plane_stride];
                });
            }
        }
        __builtin_amdgcn_s_waitcnt(0);
        const bool poll = gridDim.z > 1 && (round & 3) == 3;
        if (apex_all_retired<R>(A, ray, j0, poll)) break;
    }
}

// blockIdx.y = ray set = light * 3 + dominant axis; blockIdx.z = triangle chunk.  Set s works on
// the n = live_counts[s] ray numbers active[s * light_ray_stride ...]; the rays, counts and
// lists of light l start l * light_ray_stride rays into their arrays, its planes
// l * light_plane_stride float4 into theirs.
template <int R, bool CULL>
__global__ void __launch_bounds__(RT_WG, R >= 16 ? RT_APEX16_WAVES : (R >= 8 ? RT_APEX8_WAVES : RT_MIN_WAVES))
k_sweep_apex(SweepArgs A)
{
    // all pairs: two buffers of one tile's five planes; tile culling: two of a group's squares
    constexpr int kTileFloat4 = CULL ? RT_CULL_GROUP * RT_TILE : (RT_APEX_PLANES + 1) * RT_TILE;
    // (one more entry: the read-ahead of the all-pairs loop touches the one behind the second buffer)
    __shared__ float4 lds[2 * kTileFloat4 + 1];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;

    const long set = blockIdx.y, light = set / 3;
    const int axis = (int)(set - 3 * light);
    A.planes += light * A.light_plane_stride;
    A.S += 3 * light * A.light_ray_stride;
    A.E += 3 * light * A.light_ray_stride;
    A.cnt += light * A.light_ray_stride;
    A.list += light * A.light_ray_stride * RT_LIST_K;
    if (A.occ_id) A.occ_id += light * A.light_ray_stride;
    if (A.tri_order) A.tri_order += light * A.tri_order_stride;
    A.active += set * A.light_ray_stride;
    A.tails += set * A.light_ray_stride;
    A.n_rays = A.live_counts[set];
    if ((long)blockIdx.x * (RT_WG * R) >= A.n_rays) return;     // the grid is sized for the largest set

    ApexRay ray[R];
    const float qnan = __builtin_nanf("");
    // a wave owns 64 R consecutive rays of the list (neighbours on the screen, or in direction
    // once sorted: the fewer triangles have any of them in their square, the fewer rare-path
    // entries the wave pays for); ray r of this lane: active[j0 + r * 64]
    const long j0 = (long)blockIdx.x * (RT_WG * R) + wave * (64 * R) + lane;

    // The rays' numbers and states first, all in flight together (entry 0 stands in beyond the
    // list's end: a fetch under a condition is waited for inside its branch), then ray after ray
    // its end points, the next ray's on their way while this one's direction is worked out.
    int sl_[R], state_[R];
#pragma unroll
    for (int r = 0; r < R; ++r) sl_[r] = A.active[j0 + r * 64 < A.n_rays ? j0 + r * 64 : 0];
#pragma unroll
    for (int r = 0; r < R; ++r) state_[r] = A.cnt[sl_[r]];
    double nS[3], nE[3];
    {
        const double* s = A.S + 3 * (long)sl_[0];
        const double* e = A.E + 3 * (long)sl_[0];
        nS[0] = s[0]; nS[1] = s[1]; nS[2] = s[2];
        nE[0] = e[0]; nE[1] = e[1]; nE[2] = e[2];
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const long j = j0 + r * 64;
        bool valid = j < A.n_rays;
        const double cS[3] = {nS[0], nS[1], nS[2]}, cE[3] = {nE[0], nE[1], nE[2]};
        if (r + 1 < R) {
            const double* s = A.S + 3 * (long)sl_[r + 1 < R ? r + 1 : r];
            const double* e = A.E + 3 * (long)sl_[r + 1 < R ? r + 1 : r];
            nS[0] = s[0]; nS[1] = s[1]; nS[2] = s[2];
            nE[0] = e[0]; nE[1] = e[1]; nE[2] = e[2];
        }
        double D[3] = {cE[0] - cS[0], cE[1] - cS[1], cE[2] - cS[2]};
        {
            const double len = __builtin_sqrt(D[0] * D[0] + D[1] * D[1] + D[2] * D[2]);
            const double sig = len / (axis == 0 ? D[0] : (axis == 1 ? D[1] : D[2]));
            if (valid) A.tails[j] = make_float4((float)len, (float)sig, __int_as_float(sl_[r]), 0.0f);     // for finish_apex
            // listed rays have a finite, non-zero direction (k_bin_rays); be safe regardless
            if (!(len > 0) || !(sig == sig) || sig > 1e300 || sig < -1e300) valid = false;
            // the ray's count and flags live in cnt[] (zeroed by the host before the first range)
            if (state_[r] & (RT_CNT_OCCLUDED | RT_CNT_RETIRED)) valid = false;
        }
        const double du = (axis == 0 ? D[1] : D[0]), dv = (axis == 2 ? D[1] : D[2]);
        const double dk = axis == 0 ? D[0] : (axis == 1 ? D[1] : D[2]);
        // a retired ray carries a NaN direction, which never passes a test
        ray[r].u = valid ? (float)(du / dk) : qnan;
        ray[r].v = valid ? (float)(dv / dk) : qnan;
        // one ray at a time: interleaving the fp64 set-up of all rays would make this prologue,
        // not the hot loop, decide the kernel's register count and with it the occupancy
        __builtin_amdgcn_sched_barrier(0);
    }

    if constexpr (CULL) {
        unsigned kept = 0;      // triangles that met this wave's rays one by one
        auto tile = reinterpret_cast<float4 (*)[RT_CULL_GROUP * RT_TILE]>(lds);
        switch (axis) {         // uniform over the workgroup
        case 0: apex_tiles_culled<0, R>(A, tile, ray, j0, wave, lane, kept); break;
        case 1: apex_tiles_culled<1, R>(A, tile, ray, j0, wave, lane, kept); break;
        default: apex_tiles_culled<2, R>(A, tile, ray, j0, wave, lane, kept); break;
        }
        // pairs tested one by one: those of each wave's survivors
        if (A.tests && lane == 0) atomicAdd(A.tests, (unsigned long long)kept * (64 * R));
    } else {
        unsigned long long tiles_done = 0;
        auto tile = reinterpret_cast<float4 (*)[RT_APEX_PLANES + 1][RT_TILE]>(lds);
        switch (axis) {
        case 0: apex_tiles<0, R>(A, tile, ray, j0, wave, lane, tiles_done); break;
        case 1: apex_tiles<1, R>(A, tile, ray, j0, wave, lane, tiles_done); break;
        default: apex_tiles<2, R>(A, tile, ray, j0, wave, lane, tiles_done); break;
        }
        if (A.tests && tid == 0) atomicAdd(A.tests, tiles_done * (unsigned long long)RT_TILE * (RT_WG * R));
    }
}

// Bins the shadow rays of every light by the dominant axis of their direction: ray numbers
// (local pixel numbers) of set light * 3 + axis go to out[set * stride ...], their number to
// out_counts[set].  Rays are visited in the order of tiles of 64 x 16 pixels of the step_in grid, and
// every workgroup writes its slice in that order, so the rays of a sweep workgroup stay
// neighbours on the screen (they tend to retire together).  Unused slots (NaN origin) are left
// out, and step_in / step_out select the rays of one level of the occluder cache.
// A workgroup takes kBinRounds x RT_WG consecutive entries of that order, a lane one of every
// round: all its look-ups are in flight together, one pass, two barriers (a pass to count and a
// pass to place, each round waiting for its own loads and two barriers, took 100-130 us for a
// 3840 x 2160 frame where the bytes moved are worth 20).
constexpr int kBinRounds = 16;
// The workgroups of a launch all add to the same few counters.  Counters that share a cache line
// are served one after the other, whichever of them an update is for: thousands of workgroups
// times three sets of a light in one line took 80 us of a launch whose data is worth 20.  So
// k_bin_rays counts into a scratch array with one counter per 256 bytes, and k_bin_counts copies
// the totals to where everything else reads them.
constexpr int kBinCountStride = RT_BIN_COUNT_STRIDE;
__global__ void k_bin_counts(const int* __restrict__ padded, int sets, int* __restrict__ out_counts)
{
    if ((int)threadIdx.x < sets) out_counts[threadIdx.x] = padded[threadIdx.x * kBinCountStride];
}

__global__ void __launch_bounds__(RT_WG)
k_bin_rays(const double* __restrict__ S, const double* __restrict__ E, const int8_t* __restrict__ axis,
           const int32_t* __restrict__ hit, int W, int rows, long stride, int step_in, int step_out,
           int32_t* __restrict__ out, int* __restrict__ out_counts)
{
    constexpr int WAVES = RT_WG / 64;
    static_assert(kBinRounds * WAVES == 64, "one wave scans the (round, wave) counts");
    __shared__ int s_cnt[3][kBinRounds * WAVES];
    __shared__ int s_base[3];
    const long light = blockIdx.y;
    S += 3 * light * stride;
    E += 3 * light * stride;
    if (axis) axis += light * stride;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // the entries: the pixels of the step_in grid (one level of the occluder cache), in tiles of
    // 64 x 16 of them
    const int Ws = (W + step_in - 1) / step_in, Rs = (rows + step_in - 1) / step_in;
    const int tiles_x = (Ws + 63) / 64, tiles_y = (Rs + 15) / 16;
    const long total = (long)tiles_x * tiles_y * 1024;
    const long lo = (long)blockIdx.x * (kBinRounds * RT_WG);
    int id[kBinRounds], ax[kBinRounds];
    // (no division per entry: a workgroup's entries are kBinRounds / 4 whole tiles, and the sample
    // steps are powers of two in practice)
    static_assert(RT_WG == 256 && kBinRounds % 4 == 0, "a tile is four rounds");
    const int out_mask = step_out > 0 && (step_out & (step_out - 1)) == 0 ? step_out - 1 : -1;
    auto on_grid = [](int v, int step, int mask) { return mask >= 0 ? (v & mask) == 0 : v % step == 0; };
    const long t0 = lo >> 10;
    int tx = (int)(t0 % tiles_x), ty = (int)(t0 / tiles_x);
#pragma unroll
    for (int r = 0; r < kBinRounds; ++r) {
        const int w = (r & 3) * RT_WG + (int)threadIdx.x;
        const int px = step_in * (tx * 64 + (w & 63)), lr = step_in * (ty * 16 + (w >> 6));
        bool ok = lo + (long)r * RT_WG + threadIdx.x < total && px < W && lr < rows;
        // ... and not on the coarser step_out grid, whose rays went before
        ok = ok && !(step_out > 0 && on_grid(px, step_out, out_mask) && on_grid(lr, step_out, out_mask));
        id[r] = ok ? lr * W + px : -1;
        if ((r & 3) == 3 && ++tx == tiles_x) { tx = 0; ++ty; }
    }
    if (axis) {
        // (the classification made when the ray was written; a pixel that hit nothing has no ray
        // and its slots were never written: hit[] decides)
        // (unconditional loads, entry 0 standing in where there is no pixel: a load under a
        // condition is waited for inside its branch, sixteen round trips one after the other)
        int h[kBinRounds];
        if (hit) {
#pragma unroll
            for (int r = 0; r < kBinRounds; ++r) h[r] = hit[id[r] < 0 ? 0 : id[r]];
        } else {
#pragma unroll
            for (int r = 0; r < kBinRounds; ++r) h[r] = 0;
        }
#pragma unroll
        for (int r = 0; r < kBinRounds; ++r) ax[r] = (int)axis[id[r] < 0 ? 0 : id[r]];
#pragma unroll
        for (int r = 0; r < kBinRounds; ++r) ax[r] = (id[r] < 0 || h[r] < 0) ? -1 : ax[r];
    } else {
#pragma unroll
        for (int r = 0; r < kBinRounds; ++r) {
            ax[r] = -1;
            if (id[r] < 0 || (hit && hit[id[r]] < 0)) continue;
            const double* s = S + 3 * (long)id[r];
            const double* en = E + 3 * (long)id[r];
            const double sx = s[0];
            if (!(sx == sx)) continue;                  // unused slot
            const double dx = fabs(en[0] - sx), dy = fabs(en[1] - s[1]), dz = fabs(en[2] - s[2]);
            if (!(dx + dy + dz > 0) || !(dx + dy + dz < 1e300)) continue;      // zero-length or non-finite: hits nothing
            ax[r] = dx >= dy && dx >= dz ? 0 : (dy >= dz ? 1 : 2);
        }
    }
    // a ray's place: the rays of its set in the rounds and waves before it, then in the lanes before it
    int before[kBinRounds];
#pragma unroll
    for (int r = 0; r < kBinRounds; ++r) {
        before[r] = 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const unsigned long long m = __ballot(ax[r] == k);
            if (lane == 0) s_cnt[k][r * WAVES + wave] = __popcll(m);
            if (ax[r] == k) before[r] = __popcll(m & ((1ull << lane) - 1));
        }
    }
    __syncthreads();
    if (wave == 0) {
        int total_k[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int c = s_cnt[k][lane];
            const int incl = rt_sweep::wave_scan(c, lane);
            total_k[k] = __builtin_amdgcn_readlane(incl, 63);
            s_cnt[k][lane] = incl - c;
        }
        // the three reservations in one instruction (one after the other, each waited for the
        // counter's answer before the next left)
        if (lane < 3) {
            const int t = lane == 0 ? total_k[0] : (lane == 1 ? total_k[1] : total_k[2]);
            s_base[lane] = t ? atomicAdd(&out_counts[(light * 3 + lane) * kBinCountStride], t) : 0;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kBinRounds; ++r) {
        const int k = ax[r];
        if (k >= 0) out[(light * 3 + k) * stride + s_base[k] + s_cnt[k][r * WAVES + wave] + before[r]] = id[r];
    }
}

template <class K>
int launch(K kernel, const SweepArgs& a, dim3 grid, hipStream_t st, const char* what)
{
    hipLaunchKernelGGL(kernel, grid, dim3(RT_WG), 0, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        rt_set_error("%s launch failed: %s", what, hipGetErrorString(e));
        return RT_E_HIP;
    }
    return RT_OK;
}

int check_common(const SweepArgs& a)
{
    if (!a.planes || !a.cnt || !a.list) {
        rt_set_error("apex sweep: null buffer");
        return RT_E_ARG;
    }
    if (a.num_tris_padded <= 0 || a.num_tris_padded % RT_TILE != 0 ||
        a.num_tris > a.num_tris_padded || a.plane_stride < a.num_tris_padded) {
        rt_set_error("apex sweep: triangle planes must be padded to a multiple of %d", RT_TILE);
        return RT_E_ARG;
    }
    if (a.tri_begin < 0 || a.tri_end > a.num_tris_padded || a.tri_begin > a.tri_end ||
        a.tri_begin % RT_TILE != 0 || a.tri_end % RT_TILE != 0) {
        rt_set_error("apex sweep: triangle range [%d,%d) must be tile-aligned inside [0,%d]",
                     a.tri_begin, a.tri_end, a.num_tris_padded);
        return RT_E_ARG;
    }
    return RT_OK;
}

}  // namespace

int rt_device_cus()
{
    static int cus = 0;         // the device query is slow: once per process
    if (cus == 0) {
        hipDeviceProp_t prop;
        int dev = 0;
        cus = 256;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
            prop.multiProcessorCount > 0)
            cus = prop.multiProcessorCount;
    }
    return cus;
}

// Launch shape of the pixel-grid sweep.  The most sharing per lane (RT_RASTER_RPL pixels) is
// kept as long as the image is wide enough.  Every workgroup of a launch takes the same time,
// so the machine drains in rounds and the last, partly filled round is wasted time; cutting the
// triangle range into chunks that run as separate workgroups makes the rounds shorter and fills
// small bands.  The chunks of a pixel share its candidate list and closest-hit bound (atomics in
// the rare path), so their number costs neither memory nor pruning: about 24 rounds of
// workgroups: the cap of RT_RASTER_MAX_CHUNKS for a 3840x2160 frame of a million triangles at 2048
// pixels per lane (17 workgroups of pixels) and for an eighth of it.  A chunk may be as short as one tile (measured: a 6 k-triangle mesh at 3840x2160 sweeps in
// 2.7 ms with 49 one-tile chunks, 3.3 ms with 6 chunks of 8); a mesh too small to fill the machine
// even so gives up pixels per lane instead.  RT_AMD_RASTER_FORCE="<pixels per lane>,<chunks>[,<waves side by
// side, 0 = dealt row by row>]" overrides the choice (read at every call: tests use it to run
// every kernel shape on small inputs).
void rt_raster_shape(int W, int rows, int tiles, bool regular, int* pixels_per_lane, int* chunks, int* waves_x)
{
    const bool wave_form = regular && !rt_tile_culling();       // the wave-block form is available
    const long slots = (regular ? 8L : 5L) * rt_device_cus();      // resident workgroups
    int max_chunks = tiles;                            // a chunk may be a single tile
    if (max_chunks < 1) max_chunks = 1;
    // (the chunks of a pixel share its candidate list, so their number costs no memory)
    static const int env_cap = getenv("RT_AMD_RASTER_MAX_CHUNKS") ? atoi(getenv("RT_AMD_RASTER_MAX_CHUNKS")) : 0;
    const long cap = env_cap > 0 ? env_cap : RT_RASTER_MAX_CHUNKS;
    if (max_chunks > cap) max_chunks = (int)cap;
    // pixels the working waves cover: what the arithmetic is paid for
    auto swept = [&](int r) { return rt_raster_pairs(W, rows, r, 1); };
    // the lattice form in regular rows gives every wave-sized block a workgroup of its own (no idle
    // wave at all); otherwise the waves are dealt row by row, which leaves at most three idle in the
    // whole launch; the rectangular layouts remain for forced shapes
    auto best_layout = [&](int) { return wave_form ? -1 : 0; };
    static const int rounds = getenv("RT_AMD_RASTER_ROUNDS") ? atoi(getenv("RT_AMD_RASTER_ROUNDS")) : 24;
    if (const char* force = getenv("RT_AMD_RASTER_FORCE")) {
        int fr = 0, fc = 0, fw = 0;
        const int n = sscanf(force, "%d,%d,%d", &fr, &fc, &fw);
        const bool lattice = fr >= 1 && fr <= RT_RASTER_RPL && (fr & (fr - 1)) == 0;
        const bool block = wave_form && rt_wave_block_shape(fr);
        if (n >= 2 && fc >= 1 && fc <= (tiles < 1 ? 1 : tiles) && (lattice || block)) {
            *pixels_per_lane = fr;
            *chunks = fc;
            *waves_x = block ? -1 : ((n == 3 && (fw == -1 || fw == 0 || fw == 1 || fw == 2 || fw == 4)) ? fw : best_layout(fr));
            return;
        }
    }
    // Wave-block form: a workgroup per block of 4096 columns by 64 NL rows, its hot loop 98 + 1.5 NL
    // instructions per wave and triangle whatever the image's width.  The taller the block, the fewer
    // blocks and the cheaper the hot loop, and the more triangles go on to the rare path (all of
    // them, with one block for the image) -- which works through them 32 at a time and is what the
    // launch's time then is.  Measured at 3840x2160, a million triangles: 4.9 ms per launch as a
    // lattice of 256 rows in every lane, 2.25 ms as 4096 x 256 blocks.  The launch needs two rounds of
    // workgroups from blocks x chunks; a mesh too small for that even with 64-row blocks keeps the
    // lattice form below, where the pixels supply the workgroups.
    if (wave_form) {
        static const int shapes[] = {36, 16, 8, 4, 2, 1};
        int best = 0;
        double best_cost = 0;
        for (int nl : shapes) {
            const int r = RT_WAVE_BLOCK_RPL * nl;
            const long wgs = rt_raster_tiles(W, rows, r, -1);
            if (wgs * max_chunks < 2 * slots) continue;
            const double cost = (double)wgs * (98.0 + 1.5 * nl);
            if (!best || cost < best_cost) {
                best = r;
                best_cost = cost;
            }
        }
        if (best) {
            const long wgs = rt_raster_tiles(W, rows, best, -1);
            // (the chunks themselves may be twice as many as the cap the shapes are compared under: a
            // million triangles at 3840x2160 sweep in 1.40 ms with 1024 chunks, 1.23 with 2048, 1.20
            // with 3072, 1.16 with 4096 and 1.18 with one tile per chunk)
            const long most = tiles < 2 * cap ? (tiles < 1 ? 1 : tiles) : 2 * cap;
            long c = (rounds * slots + wgs - 1) / wgs;
            if (c > most) c = most;
            *pixels_per_lane = best;
            *chunks = c < 1 ? 1 : (int)c;
            *waves_x = -1;
            return;
        }
    }
    // Lattice form.  (With tile culling a wave's pixel rectangle should not be too wide: the narrower,
    // the fewer triangles survive its test; 32 pixels per lane measured best on a 1 M-triangle 4K frame.)
    int r = rt_tile_culling() && RT_RASTER_RPL > 32 ? 32 : RT_RASTER_RPL;
    // The wider a wave's pixels (a lane of 1024 pixels has 64 columns of 16 rows, the wave is 4096 x 16
    // pixels), the fewer instructions a pair costs, and the more often the wave has a pixel inside a
    // triangle's box and enters the rare path.  How often depends on the size of the triangles on
    // the screen, for which their number stands in.  Measured at 3840x2160 in the regular-row form
    // (the wave walks a passing box's pixels together): 6 000 triangles sweep in 0.56 ms with 128
    // pixels per lane, 0.50 (512), 0.57 (1024), 0.95 (2048).  In the other forms every lane walks its
    // own columns and rows: a million triangles in 12.7 ms with 2048 pixels per lane, 17.4 with 1024,
    // 20.6 with 512; 100 000 in 3.4 (1024), 2.7 (512), 3.9 ms (256); 6 000 fastest with 128.
    const int widest = wave_form ? (tiles >= 32 ? 1024 : 128) : (tiles >= 4096 ? 2048 : (tiles >= 256 ? 512 : 128));
    if (r > widest) r = widest;
    while (r > 1 && rt_raster_wave_cols(r) > W + rt_raster_wave_cols(r) / 2) r /= 2;    // no wave mostly off the image
    // more pixels per lane cost fewer instructions per pair, but their wider tiles may cover
    // more pixels beyond the image's edges: take the cheaper product
    while (r > 8) {
        const RasterMix m = rt_raster_mix(r), h = rt_raster_mix(r / 2);
        const double cost = (double)(m.fma + m.add + m.logic + m.compare) / r * swept(r);
        const double cost_half = (double)(h.fma + h.add + h.logic + h.compare) / (r / 2) * swept(r / 2);
        // ... unless the wider lanes leave too few workgroups for two rounds of the machine with
        // chunks of at least two tiles (a workgroup of wide lanes has more to set up, and a
        // one-tile chunk does not repay it: the 6 k-triangle mesh at 4K sweeps in 0.6 ms with 128
        // pixels per lane and in 1.2 ms with 256)
        const long long_chunks = tiles / 2 < 1 ? 1 : (tiles / 2 < max_chunks ? tiles / 2 : max_chunks);
        const bool enough = rt_raster_tiles(W, rows, r, best_layout(r)) * long_chunks >= 2 * slots;
        if (cost <= cost_half && enough) break;
        r /= 2;
    }
    for (;;) {
        const int wx = best_layout(r);
        const long wgs = rt_raster_tiles(W, rows, r, wx);
        if (wgs * max_chunks >= slots || r == 1) {
            long c = (rounds * slots + wgs - 1) / wgs;  // about 24 rounds are fine-grained enough
            if (c > max_chunks) c = max_chunks;
            // (a workgroup of wide lanes has more to set up: its chunks keep two tiles)
            if (r >= 256 && c > (tiles / 2 < 1 ? 1 : tiles / 2)) c = tiles / 2 < 1 ? 1 : tiles / 2;
            if (c < 1) c = 1;
            *pixels_per_lane = r;
            *chunks = (int)c;
            *waves_x = wx;
            return;
        }
        r /= 2;
    }
}

int rt_launch_sweep_raster(const SweepArgs& a, hipStream_t st)
{
    int rc = check_common(a);
    if (rc) return rc;
    if (a.W <= 0 || a.H <= 0 || a.rows.rows <= 0 || a.rows.block <= 0 || a.rows.parts <= 0) {
        rt_set_error("raster sweep: empty band");
        return RT_E_ARG;
    }
    if (!a.best_hi) {
        rt_set_error("raster sweep: null buffer");
        return RT_E_ARG;
    }
    int r = 1, chunks = 1, wx = 4;
    const bool regular = a.regular != 0;
    if (regular && !rt_rows_regular(a.rows)) {
        rt_set_error("raster sweep: the row set is not regular");
        return RT_E_ARG;
    }
    rt_raster_shape(a.W, a.rows.rows, (a.tri_end - a.tri_begin) / RT_TILE, regular, &r, &chunks, &wx);
    if (a.chunks != chunks || a.waves_x != wx) {
        rt_set_error("raster sweep: caller and launcher disagree on the launch shape (%d chunks, %d waves "
                     "side by side vs %d, %d)", a.chunks, a.waves_x, chunks, wx);
        return RT_E_ARG;
    }
    const dim3 grid((unsigned)rt_raster_tiles(a.W, a.rows.rows, r, wx), chunks);
#define RT_RASTER_CASE(n)                                                                                  \
    case n:                                                                                                \
        return a.cull ? launch(k_sweep_raster<n, true, false>, a, grid, st, "raster sweep (tile culling)")  \
                      : (regular ? launch(k_sweep_raster<n, false, true>, a, grid, st, "raster sweep")     \
                                 : launch(k_sweep_raster<n, false, false>, a, grid, st, "raster sweep"));
    if (r >= RT_WAVE_BLOCK_RPL) {       // wave-block form: 4096 columns by 64 NL rows per workgroup
        if (!regular || a.cull || wx != -1) {
            rt_set_error("raster sweep: %d pixels per lane exist in the wave-block form only", r);
            return RT_E_ARG;
        }
        switch (r / RT_WAVE_BLOCK_RPL) {
        case 1: return launch(k_sweep_raster_wave<1>, a, grid, st, "raster sweep");
        case 2: return launch(k_sweep_raster_wave<2>, a, grid, st, "raster sweep");
        case 4: return launch(k_sweep_raster_wave<4>, a, grid, st, "raster sweep");
        case 8: return launch(k_sweep_raster_wave<8>, a, grid, st, "raster sweep");
        case 16: return launch(k_sweep_raster_wave<16>, a, grid, st, "raster sweep");
        case 36: return launch(k_sweep_raster_wave<36>, a, grid, st, "raster sweep");
        }
        rt_set_error("raster sweep: unsupported pixels per lane %d", r);
        return RT_E_ARG;
    }
    switch (r) {
        RT_RASTER_CASE(1)
        RT_RASTER_CASE(2)
        RT_RASTER_CASE(4)
        RT_RASTER_CASE(8)
#if RT_RASTER_RPL >= 16
        RT_RASTER_CASE(16)
#endif
#if RT_RASTER_RPL >= 32
        RT_RASTER_CASE(32)
#endif
#if RT_RASTER_RPL >= 64
        RT_RASTER_CASE(64)
#endif
#if RT_RASTER_RPL >= 128
        RT_RASTER_CASE(128)
#endif
#if RT_RASTER_RPL >= 256
        RT_RASTER_CASE(256)
#endif
#if RT_RASTER_RPL >= 512
        RT_RASTER_CASE(512)
#endif
#if RT_RASTER_RPL >= 1024
        RT_RASTER_CASE(1024)
#endif
#if RT_RASTER_RPL >= 2048
        RT_RASTER_CASE(2048)
#endif
    }
#undef RT_RASTER_CASE
    rt_set_error("raster sweep: unsupported pixels per lane %d", r);
    return RT_E_ARG;
}

// Triangle chunks an any-hit launch over `tiles` tiles of a mesh of `mesh_tiles` may use: each at
// least 8 tiles long, 2 where the whole mesh is too small to fill the machine otherwise (measured:
// the shadow rays of a 6 k-triangle mesh at 640x480 take 0.19 ms instead of 0.33 ms; on meshes of
// 70 k triangles and more the shorter chunks cost 3 to 5 %).  (The chunks of a ray share its
// count and list, so their number costs no memory.)
static int apex_max_chunks(int tiles, int mesh_tiles)
{
    static const int cap = getenv("RT_AMD_APEX_MAX_CHUNKS") ? atoi(getenv("RT_AMD_APEX_MAX_CHUNKS")) : 256;
    int c = tiles / (mesh_tiles < 128 ? 2 : 8);
    if (c > cap) c = cap;
    return c < 1 ? 1 : c;
}

// Rays per lane of an any-hit launch over n rays.  The work per pair is the same for every
// choice, but the two LDS reads per triangle and the sign-test reduction are shared by more
// pairs with more rays per lane: below 4 the loop is bound by LDS bandwidth.  So the machine is
// filled with triangle chunks first, and rays per lane are given up only when the chunks do
// not suffice: 8 (at 4 waves per SIMD) wants six workgroups per CU, the others two to three.
static int apex_rays_per_lane(long n, int max_chunks, bool long_late_range)
{
    static const long kApex16MinRays = getenv("RT_AMD_APEX16_MIN_RAYS") ? atol(getenv("RT_AMD_APEX16_MIN_RAYS")) : 1000000;
    int r = RT_APEX_RPL;
    // (16 rays per lane pay where a million rays and more sweep a long range of triangles after
    // the occluded ones have gone -- the late ranges of the 1 M-triangle 4K frame: 60.3 ms of
    // shadow sweeps against 62.8 -- and cost a few percent everywhere else measured, where rays
    // still retire and a wave of 1024 keeps going for its last one: first ranges, the 70-100 k
    // closed meshes, the 6 k-triangle one, an eighth of the frame)
    if (r > 8 && !(long_late_range && n >= kApex16MinRays)) r = 8;
    while (r >= 8 && (n / (RT_WG * r)) * max_chunks < 6L * rt_device_cus()) r /= 2;
    while (r > 1 && r <= 4 && 2 * ((n + RT_WG * r - 1) / (RT_WG * r)) * max_chunks < 5L * rt_device_cus()) r /= 2;
    return r;
}

// Triangle chunks of an any-hit launch of `wgs` workgroups in total: enough to give the machine
// about RT_APEX_ROUNDS rounds of work.
static int apex_chunks(long wgs, int max_chunks)
{
    const long slots = 5L * rt_device_cus();
    long c = (RT_APEX_ROUNDS * slots + wgs - 1) / wgs;
    if (c > max_chunks) c = max_chunks;
    return c < 1 ? 1 : (int)c;
}

// a.lights = number of ray sets (lights * 3), a.n_rays = size of the largest set, a.total_rays =
// rays in all sets together (both known to the host from the last compaction)
int rt_launch_sweep_apex(const SweepArgs& a, hipStream_t st)
{
    int rc = check_common(a);
    if (rc) return rc;
    if (!a.S || !a.E || !a.active || !a.live_counts || a.lights <= 0 || a.lights % 3 != 0) {
        rt_set_error("apex sweep: needs binned ray lists (3 sets per light)");
        return RT_E_ARG;
    }
    if (a.n_rays <= 0) return RT_OK;
    const int tiles = (a.tri_end - a.tri_begin) / RT_TILE;
    const int max_chunks = apex_max_chunks(tiles, a.num_tris_padded / RT_TILE);
    // shape from the host's estimate of the live rays where it has one (the bound may be a range stale)
    const long shape_rays = a.est_rays > 0 ? a.est_rays : (a.total_rays > 0 ? a.total_rays : (long)a.n_rays * a.lights);
    int r = apex_rays_per_lane(shape_rays, max_chunks, a.resume && 8 * tiles >= a.num_tris_padded / RT_TILE);
    if (a.cull && r > 8) r = 8;     // (tile culling tests a wave's whole bundle first: the narrower, the better)
    int forced_chunks = 0;
    // RT_AMD_APEX_FORCE="<rays per lane>,<chunks>": tests run every kernel shape on small inputs
    if (const char* force = getenv("RT_AMD_APEX_FORCE")) {
        int fr = 0, fc = 0;
        if (sscanf(force, "%d,%d", &fr, &fc) == 2 && fc >= 1 &&
            (fr == 1 || fr == 2 || fr == 4 || fr == 8 || fr == 16) && fr <= RT_APEX_RPL) {
            r = fr;
            forced_chunks = fc > tiles ? (tiles < 1 ? 1 : tiles) : fc;
        }
    }
    const int per_wg = RT_WG * r;
    const unsigned wgs = (unsigned)((a.n_rays + per_wg - 1) / per_wg);
    const long all_wgs = (shape_rays + per_wg - 1) / per_wg + a.lights;
    const dim3 grid(wgs, a.lights, forced_chunks ? forced_chunks : apex_chunks(all_wgs, max_chunks));
#define RT_APEX_CASE(n)                                                                        \
    case n:                                                                                    \
        return a.cull ? launch(k_sweep_apex<n, true>, a, grid, st, "apex sweep (tile culling)") \
                      : launch(k_sweep_apex<n, false>, a, grid, st, "apex sweep");
    switch (r) {
        RT_APEX_CASE(1)
        RT_APEX_CASE(2)
        RT_APEX_CASE(4)
#if RT_APEX_RPL >= 8
        RT_APEX_CASE(8)
#endif
#if RT_APEX_RPL >= 16
        RT_APEX_CASE(16)
#endif
    }
#undef RT_APEX_CASE
    rt_set_error("apex sweep: unsupported rays per lane %d", r);
    return RT_E_ARG;
}

int rt_launch_bin_rays(const double* S, const double* E, const int8_t* axis, const int32_t* hit, int W, int rows,
                       int lights, long stride, int step_in, int step_out, int32_t* out, int* out_counts,
                       int* padded_counts, hipStream_t st)
{
    if (W <= 0 || rows <= 0 || lights <= 0) return RT_OK;
    if (!padded_counts || lights * 3 > RT_APEX_SETS) {
        rt_set_error("bin_rays: bad arguments");
        return RT_E_ARG;
    }
    if (hipMemsetAsync(padded_counts, 0, sizeof(int) * RT_BIN_COUNT_STRIDE * RT_APEX_SETS, st) != hipSuccess) {
        rt_set_error("bin_rays: clearing the counters failed");
        return RT_E_HIP;
    }
    if (step_in < 1 || step_out < 0) {
        rt_set_error("bin_rays: bad sample steps");
        return RT_E_ARG;
    }
    const int Ws = (W + step_in - 1) / step_in, Rs = (rows + step_in - 1) / step_in;
    const long total = (long)((Ws + 63) / 64) * ((Rs + 15) / 16) * 1024;
    const long wgs = (total + (long)kBinRounds * RT_WG - 1) / ((long)kBinRounds * RT_WG);
    hipLaunchKernelGGL(k_bin_rays, dim3((unsigned)wgs, lights), dim3(RT_WG), 0, st, S, E, axis, hit, W, rows,
                       stride, step_in, step_out, out, padded_counts);
    hipLaunchKernelGGL(k_bin_counts, dim3(1), dim3(64), 0, st, padded_counts, lights * 3, out_counts);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        rt_set_error("bin_rays launch failed: %s", hipGetErrorString(e));
        return RT_E_HIP;
    }
    return RT_OK;
}
