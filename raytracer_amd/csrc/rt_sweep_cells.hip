// rt_sweep_cells.hip -- any-hit sweep of the shadow rays over lattices of direction cells.
//
// The shadow rays of a light all end at it (light_model.c:60-70), so a ray is a point (u, v) on
// the plane D_k = 1 through the light (k = the dominant axis of its direction, |u|, |v| <= 1),
// and it can only hit a triangle whose central projection onto that plane contains the point --
// hence only one whose bounding square (cu, cv, rho) there, written by k_pack_apex, does.  The
// pixel-grid sweep of the primary rays (rt_sweep_apex.hip) is cheap because a lane's pixels are
// a lattice of rows and columns that share the two halves of a separable box test.  Shadow rays
// lie anywhere, so the plane of a ray set (light, axis) is cut into a regular grid of cells, a
// ray belongs to the cell its fp32 (u, v) falls into, and a lane owns a lattice of CELLS:
//
//   first test of every (cell, triangle) pair: the cell's centre against the triangle's square
//       grown by half a cell, |uc - cu| <= rho + h/2 and |vc - cv| <= rho + h/2, one interval test
//       per column and per row of the lane's lattice (an addition each, a three-way minimum per
//       two): 146 VALU instructions per 2048 pairs.  It is conservative for every ray of the cell
//       (|u - uc| <= h/2), and every cell gets the verdict it would get on its own.
//   behind it, for the cells of a lane whose column and row both passed: every ray of the cell
//       against the square itself, max(|u - cu|, |v - cv|) <= rho, and then the inside test on
//       the four dot products with its allowances and the candidate append, exactly as in
//       k_sweep_apex -- each (ray, triangle) pair gets its own verdict, and what reaches the fp64
//       pass is decided by the same arithmetic.
//
// Every cell meets every triangle; nothing is binned or culled hierarchically.  The cost of the
// hot loop follows the number of cells, which the host chooses from the number of rays.
//
// Boxes are handed to the sweep in cell units (k_cell_boxes: a cell's centre is its integer
// index), so the lattice steps by compile-time constants; only they travel through LDS, a
// triangle's planes are fetched from memory by the few waves that need them.
#include "rt_sweep_common.h"

#include <rocprim/device/device_scan.hpp>

#ifndef RT_CELLS_WAVES
#define RT_CELLS_WAVES 8    // waves per SIMD the kernel must fit (64 registers)
#endif
#ifndef RT_CELLS_UNROLL
#define RT_CELLS_UNROLL 4
#endif
#ifndef RT_CELLS_GROUP
#define RT_CELLS_GROUP 2    // tiles of boxes staged at a time
#endif

namespace {

using namespace rt_sweep;
constexpr int kGroup = RT_CELLS_GROUP;
constexpr int kUnroll = RT_CELLS_UNROLL;
static_assert(RT_TILE % (4 * kUnroll) == 0, "a wave sweeps a quarter of every tile");

template <int R>
struct CellLane {
    static constexpr int RY = rt_cells_lane_rows(R), RX = R / RY;   // rows and columns of a lane's lattice
    static constexpr int SX = RT_CELLS_SX, SY = RT_CELLS_SY;        // cells between them
    static constexpr int WW = rt_cells_wave_cols(R), WH = rt_cells_wave_rows(R);     // a wave's cells
    static_assert(SX * SY == 64 && RX >= 2 && RY >= 2 && RX % 2 == 0 && RY % 2 == 0, "lattice shape");
};

// What the rare path needs of the launch and of the workgroup's ray set
struct CellCtx {
    const float4* planes;   // of the set's light
    long plane_stride;
    long square_plane;      // the plane of that light's squares for the set's axis
    int axis, gu_log2, cell_base, fine_log2;
    const int32_t* tri_order;   // of the set's light, or nullptr
};

// The rest for one ray that lies in the triangle's square: finish_apex of rt_sweep_apex.hip with
// the planes' components already permuted for the set's axis (D' . p = u pu + v pv + pk).
__device__ __forceinline__ bool finish_cell(const CellArgs& A, const CellCtx& X, const float2& rec, int i, int tri,
                                            const float (&Pp)[3], float margin, const float (&Hp)[3], float Hw,
                                            const float (&Np)[3], float en)
{
    const int sl = A.slot[i];
    const int real = X.tri_order ? X.tri_order[tri] : tri;     // lists and the occluder cache hold triangle numbers
    if (margin != INF) {            // INF: nearly degenerate triangle, no fp32 verdict
        const float u = rec.x, v = rec.y;
        const float U = __builtin_fmaf(u, Pp[0], __builtin_fmaf(v, Pp[1], Pp[2]));
        const float V = __builtin_fmaf(u, Hp[0], __builtin_fmaf(v, Hp[1], Hp[2]));
        const float d = __builtin_fmaf(u, Np[0], __builtin_fmaf(v, Np[1], Np[2]));
        const float a = U * d, b = V * d;
        const float w = __builtin_fmaf(d, d, -a) - b;
        const float m3 = __builtin_fminf(__builtin_fminf(a, b), w);
        if (!(__builtin_fmaf(margin, __builtin_fabsf(d), m3) > 0.0f)) return false;
        // len = |light - S|, sig = 1 / (unit direction)_k: t = len + sig * k / (D'.nn)
        const float2 ls = A.len_sig[i];
        const float T = ls.y * Hw;
        return candidate_shared(A.cnt + sl, A.list + (long)sl * RT_LIST_K, A.occ_id ? A.occ_id + sl : nullptr, real,
                                a, b, w, d, T, ls.x, margin, en, U22 * __builtin_fabsf(T));
    }
    const int at = atomicAdd(A.cnt + sl, 1) & RT_CNT_MASK;
    if (at < RT_LIST_K) A.list[(long)sl * RT_LIST_K + at] = real;
    return false;
}

// The rare path, for the triangles of a quarter tile (32 of them) that had some cell of the wave
// inside their grown box -- `passed`, one bit each.  The cells of the wave's footprint whose
// centre is in a triangle's box form a rectangle of the grid, and because the rays are sorted by
// cell, row by row, the rays of one row of the rectangle are one run of the ray array: from the
// first ray at or behind the row's first cell to the first ray behind its last (the cell table
// holds, for every cell, the number of rays in the cells before it).  The whole wave works
// through all of the quarter's rectangles together:
//   * lane k < 32 takes triangle k's box and works out its rectangle;
//   * the (triangle, row) pairs of all rectangles are dealt to the lanes, 64 at a time (a running
//     sum over the rectangles' row counts and a search in it, through lane permutes), and every
//     lane looks up its row's run of rays;
//   * the rays of those runs are dealt to the lanes the same way, and every lane tests its ray
//     against its triangle: first against the triangle's square (the box less the half cell it
//     was grown by), then the inside test with the triangle's planes fetched from memory.
// Every ray of every passing cell is tested against the cell's triangle on its own; what the
// batching buys is that the dependent memory accesses of a path taken for every few triangles
// (row look-up, ray, planes, the ray's state) are waited for once per quarter tile instead of
// once per triangle, with all 64 lanes at work.
// qbox: the quarter's 32 boxes in LDS; tri0: plane entry of its first triangle.
template <int R>
__device__ __forceinline__ void cells_rare(const CellArgs& A, const CellCtx& X, const float4* qbox, int tri0,
                                           unsigned passed, int fx0, int fy0, float su, float sv, int lane,
                                           float4 (*qplanes)[3], int* flags, unsigned& tested, unsigned& visited)
{
    using L = CellLane<R>;
    // lane k: the rectangle of triangle k (no rows when the triangle did not pass, or the box
    // misses the footprint after all), and the triangle's planes on their way into LDS (qplanes:
    // the components in the order of the set's axis, D' . p = u p0 + v p1 + p2, as three records
    // (P, margin), (H, Hw), (N, en)), so that the tests below wait for memory once per quarter tile
    // and not once per batch of rays
    int x_lo = 0, nx = 0, y_lo = 0, ny = 0;
    const int fl = X.fine_log2;
    const float F = (float)(1 << fl);
    const int ax = X.axis;
    const bool mine = lane < 32 && ((passed >> lane) & 1u);
    float4 P, Hh, N;
    float margin = 0.0f;
    if (mine) {
        const float4* __restrict__ g = X.planes + (tri0 + lane);
        P = g[0];
        margin = g[X.plane_stride].w;
        Hh = g[2 * X.plane_stride];
        N = g[3 * X.plane_stride];
        const float4 box = qbox[lane];
        // The fine cells to visit.  A fine cell's centre has the coordinate j in fine units,
        // (j + 1/2) / F - 1/2 in cell units; its rays lie within half a fine cell of it, so the fine
        // cells that can hold a ray of the square are the integers j with |j - xc| <= rho hu F + 1/2,
        // xc = (cx + 1/2) F - 1/2.  The stored extent is rho hu + 1/2 + 2^-9 (cell units): the 2^-9 F
        // fine cells left over exceed the roundings of the stored centre, of xc and of the two bounds
        // here (each at most F 2^-13 on coordinates below 4096 F), so the integers between the
        // computed bounds include them.
        const float xc = __builtin_fmaf(box.x, F, 0.5f * F - 0.5f), hx = __builtin_fmaf(box.y - 0.5f, F, 0.5f);
        const float yc = __builtin_fmaf(box.z, F, 0.5f * F - 0.5f), hy = __builtin_fmaf(box.w - 0.5f, F, 0.5f);
        const float xl = __builtin_fmaxf(xc - hx, (float)(fx0 << fl)), xh = __builtin_fminf(xc + hx, (float)(((fx0 + L::WW) << fl) - 1));
        const float yl = __builtin_fmaxf(yc - hy, (float)(fy0 << fl)), yh = __builtin_fminf(yc + hy, (float)(((fy0 + L::WH) << fl) - 1));
        if (xl <= xh && yl <= yh) {
            x_lo = (int)__builtin_ceilf(xl);
            y_lo = (int)__builtin_ceilf(yl);
            nx = (int)__builtin_floorf(xh) - x_lo + 1;
            ny = (int)__builtin_floorf(yh) - y_lo + 1;
            if (nx <= 0 || ny <= 0) nx = ny = 0;
        }
    }
    const int rows_incl = wave_scan(ny, lane);
    const int rows_total = __builtin_amdgcn_readlane(rows_incl, 63);
    const int rows_excl = rows_incl - ny;
    if (rows_total == 0) return;
    if (mine) {
        // dot_ax of rt_sweep_apex.hip: D' = (1, u, v), (u, 1, v) or (u, v, 1)
        qplanes[lane][0] = make_float4(ax == 0 ? P.y : P.x, ax == 2 ? P.y : P.z, ax == 0 ? P.x : (ax == 1 ? P.y : P.z), margin);
        qplanes[lane][1] = make_float4(ax == 0 ? Hh.y : Hh.x, ax == 2 ? Hh.y : Hh.z, ax == 0 ? Hh.x : (ax == 1 ? Hh.y : Hh.z), Hh.w);
        qplanes[lane][2] = make_float4(ax == 0 ? N.y : N.x, ax == 2 ? N.y : N.z, ax == 0 ? N.x : (ax == 1 ? N.y : N.z), N.w);
    }
    wave_lds_sync();
    const int* __restrict__ table = A.table + ((long)X.cell_base << (2 * fl));
    const float qnan = __builtin_nanf("");
    int carry_e = -1;
    for (int p0 = 0; p0 < rows_total; p0 += 64) {
        // this lane's (triangle, row) pair
        const int p = p0 + lane;
        const int e = wave_owner(rows_excl, ny, p0, flags, carry_e, lane);
        carry_e = __builtin_amdgcn_readlane(e, 63);
        const int j = p - __shfl(rows_excl, e);
        const int ex = __shfl(x_lo, e), enx = __shfl(nx, e), ey = __shfl(y_lo, e);
        int first = 0, n = 0;
        if (p < rows_total) {
            const int at = ((ey + j) << (X.gu_log2 + fl)) + ex;
            first = table[at];
            n = table[at + enx] - first;        // (behind a row's last cell comes the next row's first; the table has one more entry than cells)
            ++visited;
        }
        const int rays_incl = wave_scan(n, lane);
        const int rays_total = A.debug_skip == 2 ? 0 : __builtin_amdgcn_readlane(rays_incl, 63);
        const int rays_excl = rays_incl - n;
        // the ray of lane `lane` in the batch from r0 on: its number and its pair's triangle
        const int ray_base = first - rays_excl;             // ray r of the batch is ray ray_base + r of its pair's array
        int carry_q = -1;
        auto deal = [&](int r0, int& i, int& k) {           // (called for r0 = 0, 64, 128, ... in turn)
            const int q = wave_owner(rays_excl, n, r0, flags, carry_q, lane);     // the pair whose run holds ray r0 + lane
            carry_q = __builtin_amdgcn_readlane(q, 63);
            i = __shfl(ray_base, q) + r0 + lane;
            k = __shfl(e, q);
        };
        // (the next batch's rays are fetched while this batch is tested)
        int i_next = 0, k_next = 0;
        float2 rec_next = make_float2(qnan, 0.0f);
        if (rays_total > 0) {
            deal(0, i_next, k_next);
            if (lane < rays_total) rec_next = A.rays[i_next];
        }
        for (int r0 = 0; r0 < rays_total; r0 += 64) {
            const int i = i_next, k = k_next;
            const float2 rec = rec_next;
            const bool live = r0 + lane < rays_total;
            if (r0 + 64 < rays_total) {
                deal(r0 + 64, i_next, k_next);
                rec_next = r0 + 64 + lane < rays_total ? A.rays[i_next] : make_float2(qnan, 0.0f);
            }
            if (live) {
                const float4 box = qbox[k];
                ++tested;
                // The ray against the triangle's square, in cell units: the box less the half cell it
                // was grown by (what is left of the growth, 2^-9 cell, covers the roundings here).
                // A retired ray's u is NaN and never passes.
                const float xr = __builtin_fmaf(rec.x, su, su - 0.5f), yr = __builtin_fmaf(rec.y, sv, sv - 0.5f);
                if (__builtin_fabsf(xr - box.x) <= box.y - 0.5f && __builtin_fabsf(yr - box.z) <= box.w - 0.5f &&
                    A.debug_skip != 3) {
                    const float4 p0 = qplanes[k][0], p1 = qplanes[k][1], p2 = qplanes[k][2];
                    const float Pp[3] = {p0.x, p0.y, p0.z}, Hp[3] = {p1.x, p1.y, p1.z}, Np[3] = {p2.x, p2.y, p2.z};
                    if (finish_cell(A, X, rec, i, tri0 + k, Pp, p0.w, Hp, p1.w, Np, p2.w)) A.rays[i].x = qnan;
                }
            }
        }
    }
}

// One triangle against the wave's cells.  The 64 lanes of a wave are the 8 x 8 phases of a lattice
// of RX x RY cells each, so together they own every combination of the footprint's WW = 8 RX
// columns and WH = 8 RY rows, and some cell of the wave has its centre in the box exactly when
// some column lies in the box's x extent and some row in its y extent.  Columns and rows are
// therefore dealt to the lanes (lane l: columns l, l + 64, ...; rows l, l + 64, ...; a lane beyond
// a short footprint's columns or rows holds NaN, which never passes), every lane tests its own --
// an addition each, a three-way minimum per two -- and two ballots combine the halves:
// 20 instructions per wave and triangle for 512 x 256 cells, where testing every lane's own
// lattice took 146.  box = (cx, rx, cy, ry) in cell units; X0, Y0: the lane's first column and row.
template <int R>
__device__ __forceinline__ bool cells_triangle(const float4& box, float X0, float Y0)
{
    using L = CellLane<R>;
    constexpr int NC = L::WW >= 64 ? L::WW / 64 : 1, NR = L::WH >= 64 ? L::WH / 64 : 1;
    const float d0 = X0 - box.x;
    float accx = NC == 1 ? __builtin_fabsf(d0) : INF;
#pragma unroll
    for (int c = 0; c + 1 < NC; c += 2)      // (d0 + 0.0f is not d0 to the compiler: it would keep the addition)
        accx = min3_abs(accx, c == 0 ? d0 : d0 + 64.0f * (float)c, d0 + 64.0f * (float)(c + 1));
    const float e0 = Y0 - box.z;
    float accy = NR == 1 ? __builtin_fabsf(e0) : INF;
#pragma unroll
    for (int j = 0; j + 1 < NR; j += 2)
        accy = min3_abs(accy, j == 0 ? e0 : e0 + 64.0f * (float)j, e0 + 64.0f * (float)(j + 1));
    return __builtin_amdgcn_ballot_w64(accx <= box.y) != 0 && __builtin_amdgcn_ballot_w64(accy <= box.w) != 0;
}

// Stages the boxes of `count` (<= kGroup) tiles: tile t and the ones `stride` tiles further on.
__device__ __forceinline__ void stage_cell_boxes(const float4* __restrict__ boxes, int t, int stride, int count,
                                                 float4* buf, int wave, int lane)
{
#pragma unroll
    for (int c = 0; c < (kGroup * (RT_TILE / 64) + 3) / 4; ++c) {
        const int part = c * 4 + wave;
        if (part < count * (RT_TILE / 64)) {
            const int s = part / (RT_TILE / 64), h = part % (RT_TILE / 64);
            const float4* src = boxes + (long)(t + s * stride) * RT_TILE + h * 64 + lane;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)src,
                (__attribute__((address_space(3))) void*)&buf[part * 64], 16, 0, 0);
        }
    }
}

// blockIdx.x = footprint (the WW x WH cells one wave's lattice covers) over all sets of the
// launch; blockIdx.y = triangle chunk.  The four waves of a workgroup own the same cells and
// share the staged boxes; each sweeps a quarter of every tile.
template <int R>
__global__ void __launch_bounds__(RT_WG, RT_CELLS_WAVES)
k_sweep_cells(const CellArgs A)
{
    // (one more entry: the read-ahead touches the one behind the second buffer)
    __shared__ float4 tile_lds[2 * kGroup * RT_TILE + 1];
    auto tile = reinterpret_cast<float4 (*)[kGroup * RT_TILE]>(tile_lds);
    __shared__ float4 plane_lds[RT_WG / 64][RT_TILE / 4][3];       // the rare path's planes of a quarter tile, per wave
    __shared__ int owner_lds[RT_WG / 64][64];                      // ... and its scratch for dealing rows and rays to lanes
    using L = CellLane<R>;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (scalar: triangle numbers stay out of the vector unit)

    int s = 0;
#pragma unroll
    for (int k = 1; k < RT_APEX_SETS; ++k)
        if (k < A.sets && (int)blockIdx.x >= A.set[k].fp_base) s = k;
    const CellSet cs = A.set[s];
    CellCtx X;
    X.planes = A.planes + (long)cs.light * A.light_plane_stride;
    X.plane_stride = A.plane_stride;
    X.square_plane = (long)(RT_APEX_PLANES + cs.axis) * A.plane_stride;
    X.axis = cs.axis;
    X.gu_log2 = cs.gu_log2;
    X.cell_base = cs.cell_base;
    X.fine_log2 = A.fine_log2;
    X.tri_order = A.tri_order ? A.tri_order + (long)cs.light * A.tri_order_stride : nullptr;

    const int f = (int)blockIdx.x - cs.fp_base;
    const int fx = f & ((1 << cs.fp_cols_log2) - 1), fy = f >> cs.fp_cols_log2;
    const int fx0 = fx * L::WW, fy0 = fy * L::WH;               // the footprint's first cell
    // the lane's first column and row of the footprint (none: NaN)
    const float X0 = lane < L::WW ? (float)(fx0 + lane) : __builtin_nanf("");
    const float Y0 = lane < L::WH ? (float)(fy0 + lane) : __builtin_nanf("");
    const float su = (float)(1 << (cs.gu_log2 - 1)), sv = (float)(1 << (cs.gv_log2 - 1));   // cells per unit of u, v
    const float4* __restrict__ boxes = A.boxes + (long)s * A.num_tris_padded;

    // This workgroup's share of the triangle range: the tiles blockIdx.y, blockIdx.y + gridDim.y, ...
    // of it.  The triangles come in the order of their solid angle at the light, largest first,
    // and a large triangle has many rays inside its square: dealt out tile by tile every chunk
    // gets its share of those, and every chunk meets its largest triangles first (rays that
    // are occluded retire early).
    // With few tiles (a mesh of 70 000 triangles has 540) and few footprints, a chunk of one tile
    // still leaves a wave 32 triangles whose rays it tests batch after batch, each batch waiting
    // for memory: the launch then lasts as long as its unluckiest wave.  tile_parts = 2 or 4 deals
    // every wave's 32 triangles of a tile to as many workgroups.
    const int parts = A.tile_parts, chunk = (int)blockIdx.y / parts, part = (int)blockIdx.y - chunk * parts;
    const int per_wave = (RT_TILE / 4) / parts, wave_first = wave * (RT_TILE / 4) + part * per_wave;
    const int t_first = A.tri_begin / RT_TILE, t_count = A.tri_end / RT_TILE - t_first;
    const int stride = (int)gridDim.y / parts;
    const int t0 = t_first + chunk;
    const int mine = chunk < t_count ? (t_count - chunk + stride - 1) / stride : 0;
    auto group = [&](int g) { return mine - g < kGroup ? mine - g : kGroup; };
    unsigned tested = 0, visited = 0, entered = 0;
    if (mine > 0) {
        stage_cell_boxes(boxes, t0, stride, group(0), tile[0], wave, lane);
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
    }
    int round = 0;
    for (int g = 0; g < mine; g += kGroup, ++round) {
        const int buf = round & 1, count = group(g);
        if (g + kGroup < mine)
            stage_cell_boxes(boxes, t0 + (g + kGroup) * stride, stride, group(g + kGroup), tile[buf ^ 1], wave, lane);
        const float4* __restrict__ base = tile[buf];
        for (int sub = 0; sub < count; ++sub) {
            const int tri0 = (t0 + (g + sub) * stride) * RT_TILE + wave_first;
            // (the address in a vector register the compiler cannot see through, advanced once per
            // kUnroll triangles, the reads at immediate offsets, one triangle ahead; the last read
            // of a quarter touches the entry behind it and is not used)
            const float4* qbox = base + sub * RT_TILE + wave_first;
            unsigned box_at = lds_address(qbox);
            asm volatile("" : "+v"(box_at));
            float4 next = lds_read_float4(box_at);
            unsigned passed = 0;            // the quarter's triangles with a cell of the wave in their box
            for (int j4 = 0; j4 < per_wave; j4 += kUnroll) {
#pragma unroll
                for (int k = 0; k < kUnroll; ++k) {
                    const float4 box = next;
                    next = lds_read_float4(box_at + 16u * (unsigned)(k + 1));
                    if (cells_triangle<R>(box, X0, Y0)) passed |= 1u << (j4 + k);
                }
                box_at += 16u * kUnroll;
            }
            if (__builtin_expect(passed != 0, 0)) {
                entered += __builtin_popcount(passed);
                if (A.debug_skip != 1) cells_rare<R>(A, X, qbox, tri0, passed, fx0, fy0, su, sv, lane, plane_lds[wave], owner_lds[wave], tested, visited);
            }
        }
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
    }
    if (A.ray_pairs) {          // [0]: (ray, triangle) pairs tested one by one; [2]: cell-table look-ups
        // (statistics, only when the library is profiling: summed over the workgroup first -- tens of
        // thousands of waves adding to the same three counters one by one showed up in the launch
        // time they were there to measure)
        __shared__ unsigned stat_lds[RT_WG / 64][3];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            tested += __shfl_xor(tested, off);
            visited += __shfl_xor(visited, off);
        }
        if (lane == 0) { stat_lds[wave][0] = tested; stat_lds[wave][1] = visited; stat_lds[wave][2] = entered; }
        __syncthreads();
        if (tid < 3) {
            unsigned long long sum = 0;
            for (int w = 0; w < RT_WG / 64; ++w) sum += stat_lds[w][tid];
            if (sum) atomicAdd(A.ray_pairs + (tid == 0 ? 0 : tid + 1), sum);
        }
    }
}

// ------------------------------------------------------------------------------------------
// preparation: rays into cells, boxes into cell units
// ------------------------------------------------------------------------------------------
struct CellKeySets {
    CellSet set[RT_APEX_SETS];
    int sets;
    int fine_log2;
};

// blockIdx.y = set.  Ray i of the set: its (u, v) and (len, sig) exactly as k_sweep_apex forms
// it, its slot, and the number of its fine cell, counted in hist.  The rays are then put in the
// order of their fine cells by a counting sort: a running sum over hist gives the table (rays
// in the cells before each one), and k_cell_place hands every ray a place among its cell's.
__global__ void __launch_bounds__(RT_WG)
k_cell_keys(const double* __restrict__ S, const double* __restrict__ E, const int32_t* __restrict__ lists,
            long stride, const int32_t* __restrict__ cnt, CellKeySets ks, unsigned* __restrict__ keys,
            int* __restrict__ hist, int* __restrict__ ranks, float2* __restrict__ rec, float2* __restrict__ len_sig,
            int32_t* __restrict__ slot)
{
    const CellSet cs = ks.set[blockIdx.y];
    const int axis = cs.axis;
    const long ray0 = (long)cs.light * stride;
    const int32_t* ids = lists + (long)cs.list * stride;
    const float qnan = __builtin_nanf("");
    // the grid of fine cells (a cell is 2^fine x 2^fine of them, so a ray's cell is its fine cell's)
    const int fl = ks.fine_log2;
    const float hu = (float)(1 << (cs.gu_log2 + fl - 1)), hv = (float)(1 << (cs.gv_log2 + fl - 1));    // fine cells per unit
    const int gu = 1 << (cs.gu_log2 + fl), gv = 1 << (cs.gv_log2 + fl);
    const unsigned key_base = (unsigned)cs.cell_base << (2 * fl);
    for (int i = blockIdx.x * RT_WG + threadIdx.x; i < cs.rays; i += gridDim.x * RT_WG) {
        const long sl = ray0 + ids[i];
        const double* s = S + 3 * sl;
        const double* e = E + 3 * sl;
        const double D[3] = {e[0] - s[0], e[1] - s[1], e[2] - s[2]};
        const double len = __builtin_sqrt(D[0] * D[0] + D[1] * D[1] + D[2] * D[2]);
        const double dk = axis == 0 ? D[0] : (axis == 1 ? D[1] : D[2]);
        const double sig = len / dk;
        bool valid = len > 0 && sig == sig && sig <= 1e300 && sig >= -1e300;
        if (cnt[sl] & (RT_CNT_OCCLUDED | RT_CNT_RETIRED)) valid = false;
        const double du = (axis == 0 ? D[1] : D[0]), dv = (axis == 2 ? D[1] : D[2]);
        float u = (float)(du / dk), v = (float)(dv / dk);
        if (!(u == u) || !(v == v)) valid = false;
        // fine cell: u * 2^(gu+fine-1) is exact in fp32, and so are the floor and the integer offset;
        // u = 1 belongs to the last one (its distance from that one's centre is half of it exactly)
        int col = valid ? (int)__builtin_floorf(u * hu) + gu / 2 : 0;
        int row = valid ? (int)__builtin_floorf(v * hv) + gv / 2 : 0;
        col = col < 0 ? 0 : (col > gu - 1 ? gu - 1 : col);
        row = row < 0 ? 0 : (row > gv - 1 ? gv - 1 : row);
        const long g = cs.ray_base + i;
        const unsigned key = key_base + (unsigned)((row << (cs.gu_log2 + fl)) + col);
        keys[g] = key;
        ranks[g] = atomicAdd(&hist[key], 1);    // rays per fine cell; the old count is the ray's place among its cell's
        rec[g] = make_float2(valid ? u : qnan, v);
        len_sig[g] = make_float2((float)len, (float)sig);
        slot[g] = (int32_t)sl;
    }
}

// The records into cell order: a ray's place is its cell's first (table) plus the number of its
// cell's rays counted before it (the order of a cell's rays among themselves is whatever the
// atomics of k_cell_keys made it; no result depends on it).  No atomics here: a second one per ray
// for the place cost as much again as the first for the count.
__global__ void __launch_bounds__(RT_WG)
k_cell_place(const unsigned* __restrict__ keys, int n, const int* __restrict__ table, const int* __restrict__ ranks,
             const float2* __restrict__ rec_in, const float2* __restrict__ ls_in, const int32_t* __restrict__ slot_in,
             float2* __restrict__ rec, float2* __restrict__ ls, int32_t* __restrict__ slot)
{
    const int j = blockIdx.x * RT_WG + threadIdx.x;
    if (j >= n) return;
    const unsigned key = keys[j];
    const int at = table[key] + ranks[j];
    rec[at] = rec_in[j];
    ls[at] = ls_in[j];
    slot[at] = slot_in[j];
}

// blockIdx.y = set.  The set's squares (cu, cv, rho) on [-1, 1]^2 in units of its cells: cell c
// has its centre at coordinate c, i.e. x = (cu + 1) 2^(gu-1) - 1/2.  The half extents grow by half
// a cell (a ray lies within half a cell of its cell's centre) and by 2^-9 cell for the
// roundings of the sweep's test: the centre's (coordinates below 4096: <= 2^-13 cell), the first
// subtraction's and each addition's along the lattice (<= 2^-12 each on values below 8192).
// A triangle whose box would cover more than wide_cells cells of its set's grid (near the light,
// or with no bound at all) is left out of the lattice (its box never passes) and listed instead:
// k_cells_wide tests the listed triangles against every ray of the set, ray by ray.  In the
// lattice one wave alone would walk all the rays of its footprint for such a triangle.  A list
// that is full leaves the rest to the lattice.  wide: per set, a count and RT_CELLS_WIDE_CAP entries.
__global__ void __launch_bounds__(RT_WG)
k_cell_boxes(const float4* __restrict__ planes, long plane_stride, long light_plane_stride, int np, CellKeySets ks,
             float wide_cells, float4* __restrict__ boxes, int* __restrict__ wide)
{
    const int i = blockIdx.x * RT_WG + threadIdx.x;
    if (i >= np) return;
    const CellSet cs = ks.set[blockIdx.y];
    const float4 sq = planes[(long)cs.light * light_plane_stride + (long)(RT_APEX_PLANES + cs.axis) * plane_stride + i];
    const double su = (double)(1 << (cs.gu_log2 - 1)), sv = (double)(1 << (cs.gv_log2 - 1));
    float4 box;
    if (sq.z < 0.0f) {                          // no ray of this axis can hit
        box = make_float4(0, -1.0f, 0, -1.0f);
    } else if (!(sq.z < INF)) {                 // no bound: every cell passes
        box = make_float4(0, INF, 0, INF);
    } else {
        const double slack = 0.5 + 1.0 / 512;
        box = make_float4((float)(((double)sq.x + 1.0) * su - 0.5), (float)((double)sq.z * su + slack) * 1.0000002f,
                          (float)(((double)sq.y + 1.0) * sv - 0.5), (float)((double)sq.z * sv + slack) * 1.0000002f);
    }
    if (box.y >= 0.0f && !(4.0f * box.y * box.w <= wide_cells)) {
        int* list = wide + (long)blockIdx.y * (RT_CELLS_WIDE_CAP + 1);
        const int at = atomicAdd(list, 1);
        if (at < RT_CELLS_WIDE_CAP) {
            list[1 + at] = i;
            box = make_float4(0, -1.0f, 0, -1.0f);
        }
    }
    boxes[(long)blockIdx.y * np + i] = box;
}

// blockIdx.y = set.  Every ray of the set against the set's listed wide triangles: the square
// test, then the inside test, ray by ray -- what the lattice's rare path does for a cell's rays.
__global__ void __launch_bounds__(RT_WG)
k_cells_wide(const CellArgs A, const int* __restrict__ wide)
{
    const int s = blockIdx.y;
    const CellSet cs = A.set[s];
    const int* list = wide + (long)s * (RT_CELLS_WIDE_CAP + 1);
    int n = list[0];
    n = n > RT_CELLS_WIDE_CAP ? RT_CELLS_WIDE_CAP : n;
    if (n == 0) return;
    CellCtx X;
    X.planes = A.planes + (long)cs.light * A.light_plane_stride;
    X.plane_stride = A.plane_stride;
    X.square_plane = (long)(RT_APEX_PLANES + cs.axis) * A.plane_stride;
    X.axis = cs.axis;
    X.gu_log2 = cs.gu_log2;
    X.cell_base = cs.cell_base;
    X.fine_log2 = A.fine_log2;
    X.tri_order = A.tri_order ? A.tri_order + (long)cs.light * A.tri_order_stride : nullptr;
    const int ax = cs.axis;
    const float qnan = __builtin_nanf("");
    unsigned tested = 0;
    for (int k = blockIdx.x * RT_WG + threadIdx.x; k < cs.rays; k += gridDim.x * RT_WG) {
        const int i = cs.ray_base + k;
        float2 rec = A.rays[i];
        for (int w = 0; w < n && rec.x == rec.x; ++w) {
            const int tri = list[1 + w];
            const float4* __restrict__ g = X.planes + tri;
            const float4 sq = g[X.square_plane];
            ++tested;
            if (!(max_abs(rec.x - sq.x, rec.y - sq.y) <= sq.z)) continue;
            const float4 P = g[0], Hh = g[2 * X.plane_stride], N = g[3 * X.plane_stride];
            const float margin = g[X.plane_stride].w;
            const float Pp[3] = {ax == 0 ? P.y : P.x, ax == 2 ? P.y : P.z, ax == 0 ? P.x : (ax == 1 ? P.y : P.z)};
            const float Hp[3] = {ax == 0 ? Hh.y : Hh.x, ax == 2 ? Hh.y : Hh.z, ax == 0 ? Hh.x : (ax == 1 ? Hh.y : Hh.z)};
            const float Np[3] = {ax == 0 ? N.y : N.x, ax == 2 ? N.y : N.z, ax == 0 ? N.x : (ax == 1 ? N.y : N.z)};
            if (finish_cell(A, X, rec, i, tri, Pp, margin, Hp, Hh.w, Np, N.w)) {
                rec.x = qnan;
                A.rays[i].x = qnan;
            }
        }
    }
    if (A.ray_pairs) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) tested += __shfl_xor(tested, off);
        if ((threadIdx.x & 63) == 0 && tested) atomicAdd(A.ray_pairs, (unsigned long long)tested);
    }
}

int log2_of(int v)
{
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

}  // namespace

// The grids of one launch: every set with rays gets 2^gu x 2^gv cells, at least one footprint of
// the kernel shape, about cells_per_ray cells for each of its rays (the hot loop's cost follows
// the cells, the rare path's the rays and the area of a cell); the kernel shape (cells per lane)
// is the one whose hot loop costs least for these grids.
int rt_cells_prepare(rt_scene* sc, const SweepArgs& a, const int32_t* lists, const int* counts_host, int sets,
                     hipStream_t st, CellPlan* plan)
{
    if (!plan || !lists || !counts_host || sets < 1 || sets > RT_APEX_SETS || !a.planes || !a.S || !a.E || !a.cnt ||
        !a.list) {
        rt_set_error("cell sweep: bad arguments");
        return RT_E_ARG;
    }
    const int np = a.num_tris_padded;
    if (np <= 0 || np % RT_TILE != 0) {
        rt_set_error("cell sweep: triangle planes must be padded to a multiple of %d", RT_TILE);
        return RT_E_ARG;
    }
    // RT_AMD_CELLS_PER_RAY: cells per ray of a set's grid; RT_AMD_CELLS_FORCE="<cells per lane>[,<chunks>]"
    // (read at every call: tests run every kernel shape on small inputs)
    const char* cpr_env = getenv("RT_AMD_CELLS_PER_RAY");
    double cells_per_ray = cpr_env ? atof(cpr_env) : 0.008;
    if (!(cells_per_ray > 0)) cells_per_ray = 0.008;
    // RT_AMD_CELLS_FINE: fine cells per ray -- the grid the rays are sorted into and looked up by in
    // the rare path (every cell of the lattice is 4^fine fine cells; the rays the rare path tests
    // against a triangle are those of the fine cells its square touches)
    // Default: two per ray, four when there are eight rays and more for every triangle.  Finer
    // cells mean fewer rays tested one by one and more row look-ups, and the look-ups are per
    // triangle: measured at 3840 x 2160, a 100 k-triangle torus renders in 2.83 ms with one or two
    // per ray and 2.50 with four or more, a 69 k-triangle sphere in 4.0 / 3.7 / 3.7, the 1 M-triangle
    // soup (2.6 rays per triangle) in 5.6 / 5.6 / 5.85 -- the level steps by a factor of four in
    // cells, so neighbouring settings often give the same grid.
    const char* fine_env = getenv("RT_AMD_CELLS_FINE");
    double fine_per_ray = fine_env ? atof(fine_env) : 0.0;
    bool fine_auto = !(fine_per_ray > 0);
    // RT_AMD_CELLS_WIDE: cells a triangle's box may cover before it is swept ray by ray instead
    // (default: an eighth of a wave's footprint, at least 4096 -- the ray-by-ray kernel costs a
    // test per ray of the set and triangle, so it is only for the few boxes whose rays one wave
    // would otherwise walk through alone for milliseconds; measured on the 1 M-triangle frame:
    // 1024 fills the lists and costs 7 ms, anything from 4096 up to "unbounded boxes only" 5.1 ms)
    const char* wide_env = getenv("RT_AMD_CELLS_WIDE");
    float wide_cells = wide_env ? (float)atof(wide_env) : 0.0f;
    int forced_r = 0, forced_chunks = 0;
    if (const char* force = getenv("RT_AMD_CELLS_FORCE")) {
        int fr = 0, fc = 0;
        const int n = sscanf(force, "%d,%d", &fr, &fc);
        if (n >= 1 && (fr == 2048 || fr == 512 || fr == 128 || fr == 32 || fr == 8)) forced_r = fr;
        if (n >= 2 && fc >= 1) forced_chunks = fc;
    }
    static const int shapes[] = {2048, 512, 128, 32, 8};
    auto layout = [&](int r, CellPlan& p) {
        const int ww = rt_cells_wave_cols(r), wh = rt_cells_wave_rows(r);
        CellArgs& A = p.args;
        A.sets = 0;
        long cells = 0, fps = 0, rays = 0;
        for (int s = 0; s < sets; ++s) {
            if (counts_host[s] <= 0) continue;
            CellSet& cs = A.set[A.sets++];
            cs.light = s / 3;
            cs.axis = s % 3;
            cs.list = s;
            cs.gu_log2 = log2_of(ww);
            cs.gv_log2 = log2_of(wh);
            const double target = cells_per_ray * counts_host[s];
            while ((double)(1L << (cs.gu_log2 + cs.gv_log2)) < target && cs.gu_log2 + cs.gv_log2 < 24) {
                if (cs.gu_log2 <= cs.gv_log2) ++cs.gu_log2;
                else ++cs.gv_log2;
            }
            cs.cell_base = (int)cells;
            cs.fp_base = (int)fps;
            cs.fp_cols_log2 = cs.gu_log2 - log2_of(ww);
            cs.ray_base = (int)rays;
            cs.rays = counts_host[s];
            cells += 1L << (cs.gu_log2 + cs.gv_log2);
            fps += (1L << (cs.gu_log2 + cs.gv_log2)) / ((long)ww * wh);
            rays += counts_host[s];
        }
        p.cells = cells;
        p.footprints = fps;
        p.rays = rays;
        p.cells_per_lane = r;
    };
    // The launch has footprints x chunks workgroups, and a chunk is at least kGroup tiles long: a
    // mesh of few triangles on a coarse grid would leave most of the machine idle (69 000
    // triangles are 271 chunks; with 5 footprints 8.9 ms a frame at 3840x2160, with 40 6.2 ms).
    // The grid is refined until the launch has about three rounds of workgroups.
    const int tiles = np / RT_TILE;
    // (at most 4096 chunks: the 1 M-triangle frame at 3840 x 2160 renders in 5.63 ms with 1024 chunks
    // of 7-8 tiles, 5.60 with 2048, 5.46 with 3072, 5.39 with 4096 and 5.50 with one tile per chunk)
    const long max_chunks = tiles < 1 ? 1 : (tiles > 4096 ? 4096 : tiles);
    const long enough_wgs = 8L * rt_device_cus();
    CellPlan best{};
    for (int refine = 0; refine < 8; ++refine) {
        double best_cost = -1;
        for (int r : shapes) {
            if (forced_r && r != forced_r) continue;
            CellPlan p{};
            layout(r, p);
            const double cost = (double)p.cells * rt_cells_instr(r) / r;
            if (best_cost < 0 || cost < best_cost) {
                best = p;
                best_cost = cost;
            }
        }
        if (cpr_env || best.rays == 0 || best.footprints * max_chunks >= enough_wgs || cells_per_ray >= 1.0) break;
        cells_per_ray *= 4.0;
    }
    *plan = best;
    if (plan->rays == 0) return RT_OK;
    if (fine_auto) fine_per_ray = (double)plan->rays >= 8.0 * (double)np ? 4.0 : 2.0;
    {
        int fl = 0;
        while (fl < 8 && (double)(plan->cells << (2 * (fl + 1))) <= fine_per_ray * (double)plan->rays * 2.0 &&
               (plan->cells << (2 * (fl + 1))) < (1L << 28))
            ++fl;
        for (int k = 0; k < plan->args.sets; ++k)           // (coordinates in fine cells stay below 2^16)
            while (fl > 0 && (plan->args.set[k].gu_log2 + fl > 16 || plan->args.set[k].gv_log2 + fl > 16)) --fl;
        plan->fine_log2 = plan->args.fine_log2 = fl;
    }
    const long fine_cells = plan->cells << (2 * plan->fine_log2);
    if (!(wide_cells > 0.0f)) {
        wide_cells = (float)(rt_cells_wave_cols(plan->cells_per_lane) * rt_cells_wave_rows(plan->cells_per_lane) / 8);
        if (wide_cells < 4096.0f) wide_cells = 4096.0f;
    }
    if (fine_cells >= (1L << 30) || plan->rays >= (1L << 31)) {
        rt_set_error("cell sweep: too many cells or rays for one launch");
        return RT_E_LIMIT;
    }
    const int n = (int)plan->rays;
    // triangle chunks: about 24 rounds of workgroups, a chunk at least one group of tiles long
    {
        const long slots = 8L * rt_device_cus();
        long c = (24 * slots + plan->footprints - 1) / plan->footprints;
        if (c > max_chunks) c = max_chunks;
        if (forced_chunks) c = forced_chunks > tiles ? tiles : forced_chunks;
        plan->chunks = c < 1 ? 1 : (int)c;
        // one tile per chunk, still fewer than four rounds of workgroups, and many rays for every
        // triangle (10 and more): the tiles are cut as well.  RT_AMD_CELLS_PARTS = 1, 2 or 4 forces
        // the cut.  Frames at 3840 x 2160 with 1 / 2 / 4 parts: 69 k-triangle sphere (43 rays per
        // triangle in the main level) 3.70 / 3.57 / 3.42 ms, 100 k-triangle torus (13) 2.52 / 2.38 /
        // 2.41; the 100 k soup at 1920 x 1080 (8 rays per triangle) 1.66 / 1.70 / 1.77 -- no cut there.
        int parts = 1;
        if (plan->chunks == tiles && (double)plan->rays >= 10.0 * (double)np)
            while (parts < 4 && plan->footprints * plan->chunks * parts < 4 * slots) parts *= 2;
        if (const char* e = getenv("RT_AMD_CELLS_PARTS")) {
            const int p = atoi(e);
            if (p == 1 || p == 2 || p == 4 || p == 8) parts = p;
        }
        plan->args.tile_parts = parts;
    }

    // workspace: keys, records and slots (by ray, by cell); the table, the per-cell counts and
    // rocPRIM's own for the running sum
    size_t temp_bytes = 0;
    if (rocprim::exclusive_scan(nullptr, temp_bytes, (int*)nullptr, (int*)nullptr, 0, (size_t)fine_cells + 1,
                                rocprim::plus<int>(), st) != hipSuccess) {
        rt_set_error("cell sweep: scan size query failed");
        return RT_E_HIP;
    }
    temp_bytes = (temp_bytes + 255) & ~(size_t)255;
    const size_t a4 = ((size_t)n * 4 + 255) & ~(size_t)255, a8 = ((size_t)n * 8 + 255) & ~(size_t)255;
    const size_t t4 = (((size_t)fine_cells + 1) * 4 + 255) & ~(size_t)255;
    RT_TRY(sc->ws_cell_rays.reserve(4 * a4 + 4 * a8));
    RT_TRY(sc->ws_cell_table.reserve(2 * t4 + temp_bytes));
    RT_TRY(sc->ws_cell_boxes.reserve(sizeof(float4) * (size_t)np * plan->args.sets));
    RT_TRY(sc->ws_cell_wide.reserve(sizeof(int) * (RT_CELLS_WIDE_CAP + 1) * RT_APEX_SETS));
    RT_HIP(hipMemsetAsync(sc->ws_cell_wide.p, 0, sizeof(int) * (RT_CELLS_WIDE_CAP + 1) * RT_APEX_SETS, st));
    char* ws = sc->ws_cell_rays.as<char>();
    unsigned* keys = reinterpret_cast<unsigned*>(ws);
    int32_t* slot_in = reinterpret_cast<int32_t*>(ws + a4);
    int32_t* slot = reinterpret_cast<int32_t*>(ws + 2 * a4);
    float2* rec_in = reinterpret_cast<float2*>(ws + 3 * a4);
    float2* rec = reinterpret_cast<float2*>(ws + 3 * a4 + a8);
    float2* ls_in = reinterpret_cast<float2*>(ws + 3 * a4 + 2 * a8);
    float2* ls = reinterpret_cast<float2*>(ws + 3 * a4 + 3 * a8);
    int* ranks = reinterpret_cast<int*>(ws + 3 * a4 + 4 * a8);
    int* table = sc->ws_cell_table.as<int>();
    int* hist = reinterpret_cast<int*>(sc->ws_cell_table.as<char>() + t4);
    void* temp = sc->ws_cell_table.as<char>() + 2 * t4;
    RT_HIP(hipMemsetAsync(hist, 0, t4, st));

    CellKeySets ks{};
    ks.sets = plan->args.sets;
    ks.fine_log2 = plan->fine_log2;
    int largest = 0;
    for (int k = 0; k < ks.sets; ++k) {
        ks.set[k] = plan->args.set[k];
        if (ks.set[k].rays > largest) largest = ks.set[k].rays;
    }
    unsigned wgs = (unsigned)((largest + RT_WG - 1) / RT_WG);
    if (wgs > 4096) wgs = 4096;
    hipLaunchKernelGGL(k_cell_keys, dim3(wgs, ks.sets), dim3(RT_WG), 0, st, a.S, a.E, lists, a.light_ray_stride,
                       a.cnt, ks, keys, hist, ranks, rec_in, ls_in, slot_in);
    RT_HIP(hipGetLastError());
    // table[c] = rays in the fine cells before c, for c = 0 .. fine_cells (the last entry: all of them)
    RT_HIP(rocprim::exclusive_scan(temp, temp_bytes, hist, table, 0, (size_t)fine_cells + 1, rocprim::plus<int>(), st));
    hipLaunchKernelGGL(k_cell_place, dim3((unsigned)((n + RT_WG - 1) / RT_WG)), dim3(RT_WG), 0, st, keys, n, table, ranks,
                       rec_in, ls_in, slot_in, rec, ls, slot);
    RT_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_cell_boxes, dim3((unsigned)((np + RT_WG - 1) / RT_WG), ks.sets), dim3(RT_WG), 0, st, a.planes,
                       a.plane_stride, a.light_plane_stride, np, ks, wide_cells, sc->ws_cell_boxes.as<float4>(),
                       sc->ws_cell_wide.as<int>());
    RT_HIP(hipGetLastError());

    CellArgs& A = plan->args;
    A.boxes = sc->ws_cell_boxes.as<float4>();
    A.planes = a.planes;
    A.plane_stride = a.plane_stride;
    A.light_plane_stride = a.light_plane_stride;
    A.num_tris_padded = np;
    A.tri_begin = 0;
    A.tri_end = np;
    A.table = table;
    A.rays = rec;
    A.len_sig = ls;
    A.slot = slot;
    A.cnt = a.cnt;
    A.list = a.list;
    A.occ_id = a.occ_id;
    A.tri_order = a.tri_order;
    A.tri_order_stride = a.tri_order_stride;
    A.ray_pairs = nullptr;
    plan->wide = sc->ws_cell_wide.as<int>();
    plan->largest_set = largest;
    A.debug_skip = getenv("RT_AMD_CELLS_SKIP_RARE") ? atoi(getenv("RT_AMD_CELLS_SKIP_RARE")) : 0;
    return RT_OK;
}

int rt_cells_launch(const CellPlan& plan, hipStream_t st)
{
    if (plan.rays == 0) return RT_OK;
    // the wide triangles first: they are the ones most likely to occlude
    {
        unsigned wgs = (unsigned)((plan.largest_set + RT_WG - 1) / RT_WG);
        if (wgs > 8192) wgs = 8192;
        hipLaunchKernelGGL(k_cells_wide, dim3(wgs, plan.args.sets), dim3(RT_WG), 0, st, plan.args, plan.wide);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) {
            rt_set_error("cell sweep (wide triangles) launch failed: %s", hipGetErrorString(e));
            return RT_E_HIP;
        }
    }
    const dim3 grid((unsigned)plan.footprints, (unsigned)(plan.chunks * plan.args.tile_parts));
    switch (plan.cells_per_lane) {
    case 2048: hipLaunchKernelGGL(k_sweep_cells<2048>, grid, dim3(RT_WG), 0, st, plan.args); break;
    case 512: hipLaunchKernelGGL(k_sweep_cells<512>, grid, dim3(RT_WG), 0, st, plan.args); break;
    case 128: hipLaunchKernelGGL(k_sweep_cells<128>, grid, dim3(RT_WG), 0, st, plan.args); break;
    case 32: hipLaunchKernelGGL(k_sweep_cells<32>, grid, dim3(RT_WG), 0, st, plan.args); break;
    case 8: hipLaunchKernelGGL(k_sweep_cells<8>, grid, dim3(RT_WG), 0, st, plan.args); break;
    default:
        rt_set_error("cell sweep: unsupported cells per lane %d", plan.cells_per_lane);
        return RT_E_ARG;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        rt_set_error("cell sweep launch failed: %s", hipGetErrorString(e));
        return RT_E_HIP;
    }
    return RT_OK;
}
