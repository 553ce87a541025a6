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

#include <rocprim/device/device_radix_sort.hpp>

#ifndef RT_CELLS_WAVES
#define RT_CELLS_WAVES 8    // waves per SIMD the kernel must fit (64 registers)
#endif
#ifndef RT_CELLS_UNROLL
#define RT_CELLS_UNROLL 4
#endif
#ifndef RT_CELLS_GROUP
#define RT_CELLS_GROUP 4    // tiles of boxes staged at a time
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
    int axis, gu_log2, cell_base;
    const int32_t* tri_order;   // of the set's light, or nullptr
};

// The rest for one ray that lies in the triangle's square: finish_apex of rt_sweep_apex.hip with
// the planes' components already permuted for the set's axis (D' . p = u pu + v pv + pk).
__device__ __forceinline__ bool finish_cell(const CellArgs& A, const CellCtx& X, const float4& rec, int i, int tri,
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
        // rec.z = len = |light - S|, rec.w = sig = 1 / (unit direction)_k: t = len + sig * k / (D'.nn)
        const float T = rec.w * Hw;
        return candidate_shared(A.cnt + sl, A.list + (long)sl * RT_LIST_K, A.occ_id ? A.occ_id + sl : nullptr, real,
                                a, b, w, d, T, rec.z, margin, en, U22 * __builtin_fabsf(T));
    }
    const int at = atomicAdd(A.cnt + sl, 1) & RT_CNT_MASK;
    if (at < RT_LIST_K) A.list[(long)sl * RT_LIST_K + at] = real;
    return false;
}

// A lane with a column and a row of its lattice inside the grown box: the cells of the box, and
// the rays of those cells one by one.  d0, e0: distance of the lane's first column / row from
// the box's centre (cell units); col0, row0: that first cell.
template <int R>
__device__ __forceinline__ void cells_rare(const CellArgs& A, const CellCtx& X, const float4& box, int tri, float d0,
                                           float e0, int col0, int row0, unsigned& tested)
{
    using L = CellLane<R>;
    constexpr float KX = (float)L::SX, KY = (float)L::SY;
    // the columns and rows of the box, one more on either side for the rounding of these bounds
    // (each is tested again below with the hot loop's own expression)
    int c_lo = (int)__builtin_floorf(__builtin_fmaxf((-box.y - d0) * (1.0f / KX), 0.0f)) - 1;
    int c_hi = (int)__builtin_ceilf(__builtin_fminf((box.y - d0) * (1.0f / KX), (float)(L::RX - 1))) + 1;
    int j_lo = (int)__builtin_floorf(__builtin_fmaxf((-box.w - e0) * (1.0f / KY), 0.0f)) - 1;
    int j_hi = (int)__builtin_ceilf(__builtin_fminf((box.w - e0) * (1.0f / KY), (float)(L::RY - 1))) + 1;
    c_lo = c_lo < 0 ? 0 : c_lo;
    c_hi = c_hi > L::RX - 1 ? L::RX - 1 : c_hi;
    j_lo = j_lo < 0 ? 0 : j_lo;
    j_hi = j_hi > L::RY - 1 ? L::RY - 1 : j_hi;
    // the triangle's square in (u, v) and its planes (tri is wave-uniform)
    const float4* __restrict__ g = X.planes + tri;
    const float4 sq = g[X.square_plane];
    const float4 P = g[0], Hh = g[2 * X.plane_stride], N = g[3 * X.plane_stride];
    const float margin = g[X.plane_stride].w;
    // dot_ax of rt_sweep_apex.hip: D' = (1, u, v), (u, 1, v) or (u, v, 1)
    const int ax = X.axis;
    const float Pp[3] = {ax == 0 ? P.y : P.x, ax == 2 ? P.y : P.z, ax == 0 ? P.x : (ax == 1 ? P.y : P.z)};
    const float Hp[3] = {ax == 0 ? Hh.y : Hh.x, ax == 2 ? Hh.y : Hh.z, ax == 0 ? Hh.x : (ax == 1 ? Hh.y : Hh.z)};
    const float Np[3] = {ax == 0 ? N.y : N.x, ax == 2 ? N.y : N.z, ax == 0 ? N.x : (ax == 1 ? N.y : N.z)};
    const float qnan = __builtin_nanf("");
    for (int j = j_lo; j <= j_hi; ++j) {
        if (!(__builtin_fabsf(e0 + KY * (float)j) <= box.w)) continue;
        const int row_cell = X.cell_base + ((row0 + L::SY * j) << X.gu_log2) + col0;
        for (int c = c_lo; c <= c_hi; ++c) {
            if (!(__builtin_fabsf(d0 + KX * (float)c) <= box.y)) continue;
            const int2 se = A.table[row_cell + L::SX * c];
            for (int i = se.x; i < se.y; ++i) {
                const float4 rec = A.rays[i];
                ++tested;
                // (a retired ray's u is NaN and never passes)
                if (max_abs(rec.x - sq.x, rec.y - sq.y) <= sq.z &&
                    finish_cell(A, X, rec, i, tri, Pp, margin, Hp, Hh.w, Np, N.w))
                    A.rays[i].x = qnan;
            }
        }
    }
}

// One triangle against the lane's R cells: RX columns SX cells apart, RY rows SY apart.
// box = (cx, rx, cy, ry) in cell units.
template <int R>
__device__ __forceinline__ void cells_triangle(const CellArgs& A, const CellCtx& X, const float4& box, int tri,
                                               float X0, float Y0, int col0, int row0, unsigned& tested)
{
    using L = CellLane<R>;
    constexpr float KX = (float)L::SX, KY = (float)L::SY;
    const float d0 = X0 - box.x;
    float accx = INF;
#pragma unroll
    for (int c = 0; c + 1 < L::RX; c += 2)      // (d0 + 0.0f is not d0 to the compiler: it would keep the addition)
        accx = min3_abs(accx, c == 0 ? d0 : d0 + KX * (float)c, d0 + KX * (float)(c + 1));
    const float e0 = Y0 - box.z;
    float accy = INF;
#pragma unroll
    for (int j = 0; j + 1 < L::RY; j += 2)
        accy = min3_abs(accy, j == 0 ? e0 : e0 + KY * (float)j, e0 + KY * (float)(j + 1));
    if (__builtin_expect(accx <= box.y && accy <= box.w, 0))
        cells_rare<R>(A, X, box, tri, d0, e0, col0, row0, tested);
}

// Stages the boxes of `count` (<= kGroup) consecutive tiles from tile t on.
__device__ __forceinline__ void stage_cell_boxes(const float4* __restrict__ boxes, int t, int count, float4* buf,
                                                 int wave, int lane)
{
    const float4* src = boxes + (long)t * RT_TILE;
#pragma unroll
    for (int c = 0; c < (kGroup * (RT_TILE / 64) + 3) / 4; ++c) {
        const int part = c * 4 + wave;
        if (part < count * (RT_TILE / 64))
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(src + part * 64 + lane),
                (__attribute__((address_space(3))) void*)&buf[part * 64], 16, 0, 0);
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
    X.tri_order = A.tri_order ? A.tri_order + (long)cs.light * A.num_tris_padded : nullptr;

    const int f = (int)blockIdx.x - cs.fp_base;
    const int fx = f & ((1 << cs.fp_cols_log2) - 1), fy = f >> cs.fp_cols_log2;
    const int col0 = fx * L::WW + lane % L::SX, row0 = fy * L::WH + lane / L::SX;
    const float X0 = (float)col0, Y0 = (float)row0;
    const float4* __restrict__ boxes = A.boxes + (long)s * A.num_tris_padded;

    // this workgroup's share of the triangle range: chunk blockIdx.y of gridDim.y
    const int t_first = A.tri_begin / RT_TILE, t_count = A.tri_end / RT_TILE - t_first;
    const int t_begin = t_first + (int)((long)t_count * blockIdx.y / gridDim.y);
    const int t_end = t_first + (int)((long)t_count * (blockIdx.y + 1) / gridDim.y);
    auto group = [&](int t) { return t_end - t < kGroup ? t_end - t : kGroup; };
    unsigned tested = 0;
    if (t_begin < t_end) {
        stage_cell_boxes(boxes, t_begin, group(t_begin), tile[0], wave, lane);
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
    }
    int round = 0;
    for (int t = t_begin; t < t_end; t += kGroup, ++round) {
        const int buf = round & 1, count = group(t);
        if (t + kGroup < t_end) stage_cell_boxes(boxes, t + kGroup, group(t + kGroup), tile[buf ^ 1], wave, lane);
        const float4* __restrict__ base = tile[buf];
        const int tri0 = t * RT_TILE + wave * (RT_TILE / 4);
        for (int sub = 0; sub < count; ++sub) {
            // (the address in a vector register the compiler cannot see through, advanced once per
            // kUnroll triangles, the reads at immediate offsets, one triangle ahead; the last read
            // of a quarter touches the entry behind it and is not used)
            unsigned box_at = lds_address(base + sub * RT_TILE + wave * (RT_TILE / 4));
            asm volatile("" : "+v"(box_at));
            float4 next = lds_read_float4(box_at);
            for (int j4 = 0; j4 < RT_TILE / 4; j4 += kUnroll) {
#pragma unroll
                for (int k = 0; k < kUnroll; ++k) {
                    const float4 box = next;
                    next = lds_read_float4(box_at + 16u * (unsigned)(k + 1));
                    cells_triangle<R>(A, X, box, tri0 + sub * RT_TILE + j4 + k, X0, Y0, col0, row0, tested);
                }
                box_at += 16u * kUnroll;
            }
        }
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
    }
    if (A.ray_pairs) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) tested += __shfl_xor(tested, off);
        if (lane == 0 && tested) atomicAdd(A.ray_pairs, (unsigned long long)tested);
    }
}

// ------------------------------------------------------------------------------------------
// preparation: rays into cells, boxes into cell units
// ------------------------------------------------------------------------------------------
struct CellKeySets {
    CellSet set[RT_APEX_SETS];
    int sets;
};

// blockIdx.y = set.  Ray i of the set: its record (u, v, len, sig) exactly as k_sweep_apex forms
// it, its slot, and the number of its cell as the sort key.
__global__ void __launch_bounds__(RT_WG)
k_cell_keys(const double* __restrict__ S, const double* __restrict__ E, const int32_t* __restrict__ lists,
            long stride, const int32_t* __restrict__ cnt, CellKeySets ks, unsigned* __restrict__ keys,
            int32_t* __restrict__ vals, float4* __restrict__ rec, int32_t* __restrict__ slot)
{
    const CellSet cs = ks.set[blockIdx.y];
    const int axis = cs.axis;
    const long ray0 = (long)cs.light * stride;
    const int32_t* ids = lists + (long)cs.list * stride;
    const float qnan = __builtin_nanf("");
    const float hu = (float)(1 << (cs.gu_log2 - 1)), hv = (float)(1 << (cs.gv_log2 - 1));    // cells per unit
    const int gu = 1 << cs.gu_log2, gv = 1 << cs.gv_log2;
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
        // cell: u * 2^(gu-1) is exact in fp32, and so are the floor and the integer offset; u = 1
        // belongs to the last cell (its distance from that cell's centre is half a cell exactly)
        int col = valid ? (int)__builtin_floorf(u * hu) + gu / 2 : 0;
        int row = valid ? (int)__builtin_floorf(v * hv) + gv / 2 : 0;
        col = col < 0 ? 0 : (col > gu - 1 ? gu - 1 : col);
        row = row < 0 ? 0 : (row > gv - 1 ? gv - 1 : row);
        const long g = cs.ray_base + i;
        keys[g] = (unsigned)(cs.cell_base + (row << cs.gu_log2) + col);
        vals[g] = (int32_t)g;
        rec[g] = make_float4(valid ? u : qnan, v, (float)len, (float)sig);
        slot[g] = (int32_t)sl;
    }
}

// The records into cell order, and the table: cell -> [begin, end) of its rays.
__global__ void __launch_bounds__(RT_WG)
k_cell_table(const unsigned* __restrict__ keys, const int32_t* __restrict__ vals, int n,
             const float4* __restrict__ rec_in, const int32_t* __restrict__ slot_in, float4* __restrict__ rec,
             int32_t* __restrict__ slot, int2* __restrict__ table)
{
    const int j = blockIdx.x * RT_WG + threadIdx.x;
    if (j >= n) return;
    const int src = vals[j];
    rec[j] = rec_in[src];
    slot[j] = slot_in[src];
    const unsigned key = keys[j];
    if (j == 0 || keys[j - 1] != key) table[key].x = j;
    if (j == n - 1 || keys[j + 1] != key) table[key].y = j + 1;
}

// blockIdx.y = set.  The set's squares (cu, cv, rho) on [-1, 1]^2 in units of its cells: cell c
// has its centre at coordinate c, i.e. x = (cu + 1) 2^(gu-1) - 1/2.  The half extents grow by half
// a cell (a ray lies within half a cell of its cell's centre) and by 2^-9 cell for the
// roundings of the sweep's test: the centre's (coordinates below 4096: <= 2^-13 cell), the first
// subtraction's and each addition's along the lattice (<= 2^-12 each on values below 8192).
__global__ void __launch_bounds__(RT_WG)
k_cell_boxes(const float4* __restrict__ planes, long plane_stride, long light_plane_stride, int np, CellKeySets ks,
             float4* __restrict__ boxes)
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
    boxes[(long)blockIdx.y * np + i] = box;
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
    double cells_per_ray = cpr_env ? atof(cpr_env) : 0.5;
    if (!(cells_per_ray > 0)) cells_per_ray = 0.5;
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
    CellPlan best{};
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
    *plan = best;
    if (plan->rays == 0) return RT_OK;
    if (plan->cells >= (1L << 30) || plan->rays >= (1L << 31)) {
        rt_set_error("cell sweep: too many cells or rays for one launch");
        return RT_E_LIMIT;
    }
    const int n = (int)plan->rays;
    const int tiles = np / RT_TILE;
    // triangle chunks: about 24 rounds of workgroups, a chunk at least one group of tiles long
    {
        const long slots = 8L * rt_device_cus();
        long c = (24 * slots + plan->footprints - 1) / plan->footprints;
        const long max_chunks = tiles / kGroup < 1 ? 1 : tiles / kGroup;
        if (c > max_chunks) c = max_chunks;
        if (c > 1024) c = 1024;
        if (forced_chunks) c = forced_chunks > tiles ? tiles : forced_chunks;
        plan->chunks = c < 1 ? 1 : (int)c;
    }

    // workspace: sort keys and values (in, out), records and slots (by ray, by cell), rocPRIM's own
    unsigned key_bits = 1;
    while ((1L << key_bits) < plan->cells) ++key_bits;
    size_t temp_bytes = 0;
    if (rocprim::radix_sort_pairs(nullptr, temp_bytes, (unsigned*)nullptr, (unsigned*)nullptr, (int32_t*)nullptr,
                                  (int32_t*)nullptr, (unsigned)n, 0u, key_bits, st) != hipSuccess) {
        rt_set_error("cell sweep: sort size query failed");
        return RT_E_HIP;
    }
    const size_t a4 = ((size_t)n * 4 + 255) & ~(size_t)255, a16 = ((size_t)n * 16 + 255) & ~(size_t)255;
    RT_TRY(sc->ws_cell_rays.reserve(6 * a4 + 2 * a16 + temp_bytes));
    RT_TRY(sc->ws_cell_table.reserve(sizeof(int2) * (size_t)plan->cells));
    RT_TRY(sc->ws_cell_boxes.reserve(sizeof(float4) * (size_t)np * plan->args.sets));
    char* ws = sc->ws_cell_rays.as<char>();
    unsigned* keys_in = reinterpret_cast<unsigned*>(ws);
    unsigned* keys_out = reinterpret_cast<unsigned*>(ws + a4);
    int32_t* vals_in = reinterpret_cast<int32_t*>(ws + 2 * a4);
    int32_t* vals_out = reinterpret_cast<int32_t*>(ws + 3 * a4);
    int32_t* slot_in = reinterpret_cast<int32_t*>(ws + 4 * a4);
    int32_t* slot = reinterpret_cast<int32_t*>(ws + 5 * a4);
    float4* rec_in = reinterpret_cast<float4*>(ws + 6 * a4);
    float4* rec = reinterpret_cast<float4*>(ws + 6 * a4 + a16);
    void* temp = ws + 6 * a4 + 2 * a16;

    CellKeySets ks{};
    ks.sets = plan->args.sets;
    int largest = 0;
    for (int k = 0; k < ks.sets; ++k) {
        ks.set[k] = plan->args.set[k];
        if (ks.set[k].rays > largest) largest = ks.set[k].rays;
    }
    unsigned wgs = (unsigned)((largest + RT_WG - 1) / RT_WG);
    if (wgs > 4096) wgs = 4096;
    hipLaunchKernelGGL(k_cell_keys, dim3(wgs, ks.sets), dim3(RT_WG), 0, st, a.S, a.E, lists, a.light_ray_stride,
                       a.cnt, ks, keys_in, vals_in, rec_in, slot_in);
    RT_HIP(hipGetLastError());
    RT_HIP(rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, (unsigned)n, 0u,
                                     key_bits, st));
    RT_HIP(hipMemsetAsync(sc->ws_cell_table.p, 0, sizeof(int2) * (size_t)plan->cells, st));
    hipLaunchKernelGGL(k_cell_table, dim3((unsigned)((n + RT_WG - 1) / RT_WG)), dim3(RT_WG), 0, st, keys_out, vals_out,
                       n, rec_in, slot_in, rec, slot, sc->ws_cell_table.as<int2>());
    RT_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_cell_boxes, dim3((unsigned)((np + RT_WG - 1) / RT_WG), ks.sets), dim3(RT_WG), 0, st, a.planes,
                       a.plane_stride, a.light_plane_stride, np, ks, sc->ws_cell_boxes.as<float4>());
    RT_HIP(hipGetLastError());

    CellArgs& A = plan->args;
    A.boxes = sc->ws_cell_boxes.as<float4>();
    A.planes = a.planes;
    A.plane_stride = a.plane_stride;
    A.light_plane_stride = a.light_plane_stride;
    A.num_tris_padded = np;
    A.tri_begin = 0;
    A.tri_end = np;
    A.table = sc->ws_cell_table.as<int2>();
    A.rays = rec;
    A.slot = slot;
    A.cnt = a.cnt;
    A.list = a.list;
    A.occ_id = a.occ_id;
    A.tri_order = a.tri_order;
    A.ray_pairs = nullptr;
    return RT_OK;
}

int rt_cells_launch(const CellPlan& plan, hipStream_t st)
{
    if (plan.rays == 0) return RT_OK;
    const dim3 grid((unsigned)plan.footprints, (unsigned)plan.chunks);
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
