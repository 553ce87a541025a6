// rt_exact.hip -- fp64 evaluation with the reference's exact arithmetic and comparison rules.
//
// This translation unit is compiled with -ffp-contract=off: every product and sum below is
// rounded separately, in the statement order of triangle_intersection.cu:13-29,59-95, so
// that t, u, v and every accept/reject decision are those of the reference's formulas in
// strict IEEE binary64 (fp64 division and square root are correctly rounded on gfx950).
//
// Three users:
//   * the confirmation pass behind the fp32 sweep (rt_sweep.hip): a handful of candidate
//     triangles per ray, in ascending index order, reduced with the host loop's rule
//     `dist > 0 && dist < t` (triangle_intersection.cu:167-176: strict '<', lowest index
//     wins a tie);
//   * the fallback for rays whose candidate list overflowed: a full sweep, one workgroup
//     per ray, threads striding over triangles -- the reference's own parallel axis;
//   * intersect_all_triangles_device, the reference's one-ray API.
#include "rt_internal.h"

namespace {


// triangle_intersection.cu:6-11
__device__ __forceinline__ double det_2x2(double a0, double a1, double b0, double b1)
{
    return a0 * b1 - a1 * b0;
}

// triangle_intersection.cu:13-29: 0 + A0*|B1 C1;B2 C2| + (-B0)*|A1 C1;A2 C2| + C0*|A1 B1;A2 B2|
__device__ __forceinline__ double det_3x3(const double A[3], const double B[3], const double C[3])
{
    double det = 0;
    det += A[0] * det_2x2(B[1], B[2], C[1], C[2]);
    det += -B[0] * det_2x2(A[1], A[2], C[1], C[2]);
    det += C[0] * det_2x2(A[1], A[2], B[1], B[2]);
    return det;
}

// triangle_intersection.cu:59-95 for one triangle.  Returns dist (t or -1).
__device__ __forceinline__ double exact_single(const double S[3], const double E[3],
                                               const double A[3], const double B[3],
                                               const double C[3], double& u, double& v)
{
    double AB[3], AC[3], ES[3], AS[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        AB[k] = B[k] - A[k];
        AC[k] = C[k] - A[k];
        ES[k] = S[k] - E[k];
        AS[k] = S[k] - A[k];
    }
    const double den = det_3x3(AB, AC, ES);
    if (den == 0) return -1;
    const double t = det_3x3(AB, AC, AS) / den;
    if (t < 0) return -1;
    u = det_3x3(AS, AC, ES) / den;
    if ((u < 0) || (u > 1)) return -1;
    v = det_3x3(AB, AS, ES) / den;
    if ((v < 0) || (v > 1)) return -1;
    if (u + v > 1) return -1;
    return t;
}

__device__ __forceinline__ double exact_tri(const ExactScene& es, int i, const double S[3],
                                            const double E[3], double& u, double& v)
{
    if (es.tv) {
        const double* t = es.tv + 9 * (long)i;
        const double A[3] = {t[0], t[1], t[2]};
        const double B[3] = {t[3], t[4], t[5]};
        const double C[3] = {t[6], t[7], t[8]};
        return exact_single(S, E, A, B, C, u, v);
    }
    const int a = es.A[i], b = es.B[i], c = es.C[i];
    const double A[3] = {es.x[a], es.y[a], es.z[a]};
    const double B[3] = {es.x[b], es.y[b], es.z[b]};
    const double C[3] = {es.x[c], es.y[c], es.z[c]};
    return exact_single(S, E, A, B, C, u, v);
}

// driver.c:121-133
__device__ __forceinline__ void primary_ray(int W, int H, double tan_half, int px, int py,
                                            double S[3], double E[3])
{
    S[0] = S[1] = S[2] = 0;
    E[0] = (double)(px - W / 2);
    E[1] = (double)(py - H / 2);
    E[2] = (double)(W / 2) / tan_half;
}

struct Best {
    double t, u, v;
    int idx;
};

// Confirmation of one ray's candidate list (K slots, in any order).  The reference's loop runs
// over ascending triangle numbers and replaces its winner only on `dist > 0 && dist < t`
// (triangle_intersection.cu:167-176), i.e. it keeps the smallest t and, among equal t, the lowest
// triangle number: that is the rule applied here, whatever order the candidates were listed in.
// Returns false when the list overflowed; ncand = candidates evaluated.  Entry k of ray p is
// list[p * K + k], or list[k * slot_major + p] when slot_major is not 0.
template <bool ANYHIT, int K>
__device__ __forceinline__ bool confirm_list(const ExactScene& es, const double S[3],
                                             const double E[3], const int32_t* __restrict__ cnt,
                                             const int32_t* __restrict__ list, long p, Best& best,
                                             bool& occluded, int& ncand, long slot_major = 0)
{
    best.t = 1e50;                  // triangle_intersection.cu:108
    best.idx = -1;
    best.u = best.v = 0;
    occluded = false;
    ncand = 0;
    int n = cnt[p];
    if (ANYHIT && (n & RT_CNT_OCCLUDED)) {
        occluded = true;
        return true;
    }
    n &= RT_CNT_MASK;
    if (n > K) return false;
    ncand = n;
    const int32_t* l = slot_major ? list + p : list + p * K;
    const long step = slot_major ? slot_major : 1;
    // four entries at a time, fetched together whether the list has them or not (all K slots
    // exist): fetched one by one as they were tested, every candidate cost a memory round trip for
    // its number before the one for its vertices
    static_assert(K % 4 == 0, "the list is read in groups of four slots");
    for (int k0 = 0; k0 < n; k0 += 4) {
        int ids[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) ids[j] = l[(k0 + j) * step];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (k0 + j >= n) break;
            const int i = ids[j];
            double u, v;
            const double dist = exact_tri(es, i, S, E, u, v);
            if ((dist > 0) && (dist < best.t || (dist == best.t && i < best.idx))) {       // :169
                best.t = dist; best.u = u; best.v = v; best.idx = i;
                if (ANYHIT) { occluded = true; return true; }
            }
        }
    }
    return true;
}

// Statistics counters all point at one address; a workgroup adds to them once, at its end.
__device__ __forceinline__ void block_add(int* counter, int v)
{
    __shared__ int s_sum;
    if (threadIdx.x == 0) s_sum = 0;
    __syncthreads();
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(&s_sum, v);
    __syncthreads();
    if (threadIdx.x == 0 && s_sum) atomicAdd(counter, s_sum);
    __syncthreads();
}

constexpr int RESOLVE_GRID = 4096;      // grid-stride: workgroups of the resolve kernels

__global__ void __launch_bounds__(RT_WG)
k_resolve_primary(ExactScene es, int W, int H, double tan_half, RowMap rows,
                  const int32_t* __restrict__ cnt, const int32_t* __restrict__ list, long list_stride,
                  int32_t* __restrict__ hit, double* __restrict__ tuv,
                  int32_t* __restrict__ ovf, int* __restrict__ counters)
{
    const long npix = (long)W * rows.rows;
    int total = 0;
    for (long p = (long)blockIdx.x * RT_WG + threadIdx.x; p < npix; p += (long)gridDim.x * RT_WG) {
        const int py = rows.image_row((int)(p / W)), px = (int)(p % W);
        double S[3], E[3];
        primary_ray(W, H, tan_half, px, py, S, E);
        Best b;
        bool occ;
        int ncand;
        if (!confirm_list<false, RT_RASTER_LIST_K>(es, S, E, cnt, list, p, b, occ, ncand, list_stride)) {
            ovf[atomicAdd(&counters[0], 1)] = (int32_t)p;
        } else {
            total += ncand;
            hit[p] = b.idx;
            if (b.idx != -1 && tuv) { tuv[3 * p] = b.t; tuv[3 * p + 1] = b.u; tuv[3 * p + 2] = b.v; }
        }
    }
    block_add(&counters[1], total);
}

template <bool ANYHIT>
__global__ void __launch_bounds__(RT_WG)
k_resolve_rays(ExactScene es, int n, const double* __restrict__ Sa, const double* __restrict__ Ea,
               const int32_t* __restrict__ cnt, const int32_t* __restrict__ list,
               int32_t* __restrict__ hit, double* __restrict__ tuv, double* __restrict__ point,
               uint8_t* __restrict__ occ_out, int32_t* __restrict__ ovf, int* __restrict__ counters,
               const int32_t* __restrict__ gate, long gate_n)
{
    int total = 0, live_rays = 0;
    for (long p = (long)blockIdx.x * RT_WG + threadIdx.x; p < n; p += (long)gridDim.x * RT_WG) {
        if (gate && gate[p % gate_n] < 0) continue;     // no such ray: its slots were never written
        const double Sx = Sa[3 * p];
        if (!(Sx == Sx)) {                      // NaN origin = unused ray slot
            if (ANYHIT) occ_out[p] = 0; else hit[p] = -1;
            continue;
        }
        ++live_rays;
        const double S[3] = {Sx, Sa[3 * p + 1], Sa[3 * p + 2]};
        const double E[3] = {Ea[3 * p], Ea[3 * p + 1], Ea[3 * p + 2]};
        Best b;
        bool occ;
        int ncand;
        if (!confirm_list<ANYHIT, RT_LIST_K>(es, S, E, cnt, list, p, b, occ, ncand)) {
            ovf[atomicAdd(&counters[0], 1)] = (int32_t)p;
            continue;
        }
        total += ncand;
        if (ANYHIT) {
            occ_out[p] = occ ? 1 : 0;
        } else {
            hit[p] = b.idx;
            if (b.idx != -1) {
                if (tuv) { tuv[3 * p] = b.t; tuv[3 * p + 1] = b.u; tuv[3 * p + 2] = b.v; }
                if (point) {              // triangle_intersection.cu:185-187
                    point[3 * p] = S[0] + b.t * (E[0] - S[0]);
                    point[3 * p + 1] = S[1] + b.t * (E[1] - S[1]);
                    point[3 * p + 2] = S[2] + b.t * (E[2] - S[2]);
                }
            }
        }
    }
    block_add(&counters[1], total);
    block_add(&counters[4], live_rays);
}

// Full fp64 sweep of triangles [begin,end) for one ray by one workgroup.  Thread k visits
// triangles begin+k, begin+k+256, ... in ascending order with the strict '<' rule, then the 256
// partial winners are merged with (smaller t, then smaller index), which is what the sequential
// host loop would have kept.
__device__ __forceinline__ bool better(double t, int i, double bt, int bi)
{
    return i != -1 && (bi == -1 || t < bt || (t == bt && i < bi));
}

__device__ __forceinline__ Best full_sweep(const ExactScene& es, const double S[3], const double E[3],
                                           int begin, int end)
{
    __shared__ double s_t[RT_WG];
    __shared__ int s_i[RT_WG];
    __shared__ double s_u[RT_WG], s_v[RT_WG];
    Best b;
    b.t = 1e50; b.idx = -1; b.u = b.v = 0;
    for (int i = begin + threadIdx.x; i < end; i += RT_WG) {
        double u, v;
        const double dist = exact_tri(es, i, S, E, u, v);
        if ((dist > 0) && (dist < b.t)) { b.t = dist; b.u = u; b.v = v; b.idx = i; }
    }
    s_t[threadIdx.x] = b.t; s_i[threadIdx.x] = b.idx; s_u[threadIdx.x] = b.u; s_v[threadIdx.x] = b.v;
    __syncthreads();
    for (int off = RT_WG / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            const int o = threadIdx.x + off;
            if (better(s_t[o], s_i[o], s_t[threadIdx.x], s_i[threadIdx.x])) {
                s_t[threadIdx.x] = s_t[o]; s_i[threadIdx.x] = s_i[o];
                s_u[threadIdx.x] = s_u[o]; s_v[threadIdx.x] = s_v[o];
            }
        }
        __syncthreads();
    }
    b.t = s_t[0]; b.idx = s_i[0]; b.u = s_u[0]; b.v = s_v[0];
    __syncthreads();
    return b;
}

struct Partial { double t, u, v; int idx; int pad; };

// Overflowed rays: n of them (a device-side count) share a fixed grid.  With more rays than
// workgroups each workgroup sweeps the whole mesh for its rays; with fewer, every ray is given
// `parts` workgroups that sweep a slice of the mesh each and leave a partial winner, and
// k_overflow_merge picks the winner.  parts is a function of (n, grid) both kernels agree on.
constexpr int OVERFLOW_GRID = 2048;
__device__ __forceinline__ int overflow_parts(int n)
{
    int parts = 1;
    while (parts < 64 && 2 * parts * (long)n <= OVERFLOW_GRID) parts *= 2;
    return parts;
}

struct PrimaryRays {
    int W, H;
    double tan_half;
    RowMap rows;
    __device__ void get(long p, double S[3], double E[3]) const
    {
        primary_ray(W, H, tan_half, (int)(p % W), rows.image_row((int)(p / W)), S, E);
    }
};
struct ListedRays {
    const double *Sa, *Ea;
    __device__ void get(long p, double S[3], double E[3]) const
    {
        S[0] = Sa[3 * p]; S[1] = Sa[3 * p + 1]; S[2] = Sa[3 * p + 2];
        E[0] = Ea[3 * p]; E[1] = Ea[3 * p + 1]; E[2] = Ea[3 * p + 2];
    }
};

template <bool ANYHIT>
__device__ __forceinline__ void emit(long p, const Best& b, const double S[3], const double E[3],
                                     int32_t* __restrict__ hit, double* __restrict__ tuv,
                                     double* __restrict__ point, uint8_t* __restrict__ occ_out)
{
    if (ANYHIT) {
        occ_out[p] = b.idx != -1 ? 1 : 0;
        return;
    }
    hit[p] = b.idx;
    if (b.idx == -1) return;
    if (tuv) { tuv[3 * p] = b.t; tuv[3 * p + 1] = b.u; tuv[3 * p + 2] = b.v; }
    if (point) {              // triangle_intersection.cu:185-187
        point[3 * p] = S[0] + b.t * (E[0] - S[0]);
        point[3 * p + 1] = S[1] + b.t * (E[1] - S[1]);
        point[3 * p + 2] = S[2] + b.t * (E[2] - S[2]);
    }
}

template <class RAYS, bool ANYHIT>
__global__ void __launch_bounds__(RT_WG)
k_overflow(ExactScene es, RAYS rays, const int32_t* __restrict__ ovf, const int* __restrict__ counters,
           Partial* __restrict__ part, int32_t* __restrict__ hit, double* __restrict__ tuv,
           double* __restrict__ point, uint8_t* __restrict__ occ_out)
{
    const int n = counters[0];
    const int parts = overflow_parts(n);
    double S[3], E[3];
    if (parts == 1) {
        for (int k = blockIdx.x; k < n; k += gridDim.x) {
            const long p = ovf[k];
            rays.get(p, S, E);
            const Best b = full_sweep(es, S, E, 0, es.num_tris);
            if (threadIdx.x == 0) emit<ANYHIT>(p, b, S, E, hit, tuv, point, occ_out);
        }
        return;
    }
    const int k = blockIdx.x / parts, g = blockIdx.x % parts;
    if (k >= n) return;
    rays.get(ovf[k], S, E);
    const int begin = (int)((long)es.num_tris * g / parts), end = (int)((long)es.num_tris * (g + 1) / parts);
    const Best b = full_sweep(es, S, E, begin, end);
    if (threadIdx.x == 0) {
        part[blockIdx.x].t = b.t; part[blockIdx.x].u = b.u; part[blockIdx.x].v = b.v;
        part[blockIdx.x].idx = b.idx;
    }
}

template <class RAYS, bool ANYHIT>
__global__ void __launch_bounds__(RT_WG)
k_overflow_merge(RAYS rays, const int32_t* __restrict__ ovf, const int* __restrict__ counters,
                 const Partial* __restrict__ part, int32_t* __restrict__ hit, double* __restrict__ tuv,
                 double* __restrict__ point, uint8_t* __restrict__ occ_out)
{
    const int n = counters[0];
    const int parts = overflow_parts(n);
    const int k = blockIdx.x * RT_WG + threadIdx.x;
    if (parts == 1 || k >= n) return;
    Best b;
    b.t = 1e50; b.idx = -1; b.u = b.v = 0;
    for (int g = 0; g < parts; ++g) {
        const Partial q = part[k * parts + g];
        if (better(q.t, q.idx, b.t, b.idx)) { b.t = q.t; b.u = q.u; b.v = q.v; b.idx = q.idx; }
    }
    const long p = ovf[k];
    double S[3], E[3];
    rays.get(p, S, E);
    emit<ANYHIT>(p, b, S, E, hit, tuv, point, occ_out);
}

// Occluder cache of the shadow pass.  Shadow rays of neighbouring pixels towards the same light
// are nearly parallel and start close together, so whatever occludes one very often occludes
// the next.  The rays of a coarse grid of pixels (every `step`-th in x and y) are settled first
// and record the triangle that occludes them; every ray of the next finer level then tries the
// (up to RT_SHADOW_NEIGHBOURS squared) triangles recorded around it with the reference's exact test, records the one
// that occludes it (`record`, for the level after it), and only the rays that are still open
// go through the sweep.  An any-hit query (light_model.c:70 asks whether ANY triangle is hit at
// dist > 0) does not depend on the order in which triangles are tried, so the result is
// unchanged.
__global__ void __launch_bounds__(RT_WG)
k_try_cached(ExactScene es, const double* __restrict__ Sa, const double* __restrict__ Ea,
             const int32_t* __restrict__ lists, const int* __restrict__ counts, long stride, int W,
             int rows, int row_block, int step, int record, int32_t* __restrict__ cnt,
             int32_t* __restrict__ occ_id)
{
    const long set = blockIdx.y, light = set / 3;
    const int n = counts[set];
    lists += set * stride;
    Sa += 3 * light * stride;
    Ea += 3 * light * stride;
    cnt += light * stride;
    occ_id += light * stride;
    for (long j = (long)blockIdx.x * RT_WG + threadIdx.x; j < n; j += (long)gridDim.x * RT_WG) {
        const int id = lists[j];
        const int px = id % W, lr = id / W;
        const int sx = px - px % step, sy = lr - lr % step;
        const double S[3] = {Sa[3 * (long)id], Sa[3 * (long)id + 1], Sa[3 * (long)id + 2]};
        const double E[3] = {Ea[3 * (long)id], Ea[3 * (long)id + 1], Ea[3 * (long)id + 2]};
        constexpr int NB = RT_SHADOW_NEIGHBOURS, NB2 = NB * NB, LO = (NB - 2) / 2;
        // The sample rows come from the ray's own block of rows: with interleaved row parts
        // the next block of local rows lies elsewhere in the image.  The window of NB sample
        // rows around the ray is shifted to stay inside the block where the block is tall enough.
        const int b0 = lr - lr % row_block;
        const int b1 = b0 + row_block < rows ? b0 + row_block : rows;
        const int first = (b0 + step - 1) / step * step;                 // first sample row of the block
        const int last = (b1 - 1) / step * step;                         // last one
        int wy = sy - LO * step;
        if (wy + (NB - 1) * step > last) wy = last - (NB - 1) * step;
        if (wy < first) wy = first;
        // the recorded triangles around the ray first, all loads in flight together (tried one after
        // the other, each look-up waited for the previous test: up to NB2 memory round trips in a row)
        // ... nearest samples first (the corners of the ray's own cell of the sample grid, then the
        // ring around them): what occludes a ray is most often what occludes its nearest neighbours
        int tried[NB2];
#pragma unroll
        for (int k = 0; k < NB2; ++k) {
            // position k of the try order -> (kx, ky) in the window: the inner (NB - 2 LO)^2 block first
            int kx, ky;
            if (NB == 4) {
                constexpr int order[16] = {5, 6, 9, 10, 1, 2, 4, 7, 8, 11, 13, 14, 0, 3, 12, 15};
                kx = order[k % 16] % NB;
                ky = order[k % 16] / NB;
            } else {
                kx = k % NB;
                ky = k / NB;
            }
            const int qx = sx + (kx - LO) * step, qy = wy + ky * step;
            tried[k] = (qx < 0 || qx >= W || qy > last) ? -1 : occ_id[(long)qy * W + qx];
        }
        // The distinct ones, nearest first.  A lane works through its own list densely -- bit k of
        // `open` says try k is still to be made, and a triangle that failed is struck out wherever
        // else it was recorded -- so a wave goes round as often as its unluckiest ray has distinct
        // candidates, not as often as the positions its lanes' candidates happen to sit at; and
        // nothing is compared before a try has failed (striking out duplicates up front was 120
        // comparisons for every ray, most of which are settled by their first try).
        unsigned open = 0;
#pragma unroll
        for (int k = 0; k < NB2; ++k) open |= (tried[k] >= 0 && tried[k] < es.num_tris ? 1u : 0u) << k;
        while (open) {
            const int k = __builtin_ctz(open);
            int t = tried[0];
#pragma unroll
            for (int m = 1; m < NB2; ++m) t = k == m ? tried[m] : t;
            double u, v;
            if (exact_tri(es, t, S, E, u, v) > 0) {        // triangle_intersection.cu:169
                atomicOr(&cnt[id], RT_CNT_OCCLUDED);
                if (record) occ_id[id] = t;
                break;
            }
#pragma unroll
            for (int m = 0; m < NB2; ++m) open &= ~((tried[m] == t ? 1u : 0u) << m);
        }
    }
}

// intersect_all_triangles_device: one ray, triangles spread over the whole grid, partial
// winners merged by a last single-workgroup pass.  SE = S[3],E[3]; out = {idx, t, u, v,
// point[3]} as 7 doubles (idx stored as a double, exact for any int).
__global__ void __launch_bounds__(RT_WG)
k_single_ray_partial(ExactScene es, const double* __restrict__ SE, Partial* __restrict__ part)
{
    const double S[3] = {SE[0], SE[1], SE[2]};
    const double E[3] = {SE[3], SE[4], SE[5]};
    __shared__ double s_t[RT_WG];
    __shared__ int s_i[RT_WG];
    __shared__ double s_u[RT_WG], s_v[RT_WG];
    Best b;
    b.t = 1e50; b.idx = -1; b.u = b.v = 0;
    for (long i = (long)blockIdx.x * RT_WG + threadIdx.x; i < es.num_tris; i += (long)gridDim.x * RT_WG) {
        double u, v;
        const double dist = exact_tri(es, (int)i, S, E, u, v);
        if ((dist > 0) && (dist < b.t)) { b.t = dist; b.u = u; b.v = v; b.idx = (int)i; }
    }
    s_t[threadIdx.x] = b.t; s_i[threadIdx.x] = b.idx; s_u[threadIdx.x] = b.u; s_v[threadIdx.x] = b.v;
    __syncthreads();
    for (int off = RT_WG / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            const int o = threadIdx.x + off;
            const bool other_hit = s_i[o] != -1, mine_hit = s_i[threadIdx.x] != -1;
            const bool take = other_hit && (!mine_hit || s_t[o] < s_t[threadIdx.x] ||
                                            (s_t[o] == s_t[threadIdx.x] && s_i[o] < s_i[threadIdx.x]));
            if (take) {
                s_t[threadIdx.x] = s_t[o]; s_i[threadIdx.x] = s_i[o];
                s_u[threadIdx.x] = s_u[o]; s_v[threadIdx.x] = s_v[o];
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        part[blockIdx.x].t = s_t[0]; part[blockIdx.x].idx = s_i[0];
        part[blockIdx.x].u = s_u[0]; part[blockIdx.x].v = s_v[0];
    }
}

__global__ void k_single_ray_final(const Partial* __restrict__ part, int nparts,
                                   const double* __restrict__ SE, double* __restrict__ out)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    Best b;
    b.t = 1e50; b.idx = -1; b.u = b.v = 0;
    for (int k = 0; k < nparts; ++k) {
        if (part[k].idx == -1) continue;
        if (b.idx == -1 || part[k].t < b.t || (part[k].t == b.t && part[k].idx < b.idx)) {
            b.t = part[k].t; b.idx = part[k].idx; b.u = part[k].u; b.v = part[k].v;
        }
    }
    out[0] = (double)b.idx;
    out[1] = b.t; out[2] = b.u; out[3] = b.v;
    if (b.idx != -1) {
        out[4] = SE[0] + b.t * (SE[3] - SE[0]);
        out[5] = SE[1] + b.t * (SE[4] - SE[1]);
        out[6] = SE[2] + b.t * (SE[5] - SE[2]);
    }
}

int launched(const char* what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        rt_set_error("%s launch failed: %s", what, hipGetErrorString(e));
        return RT_E_HIP;
    }
    return RT_OK;
}

constexpr int SINGLE_RAY_GRID = 256;

}  // namespace

int rt_launch_resolve_primary(const ExactScene& es, int W, int H, double tan_half, const RowMap& rows,
                              const int32_t* cnt, const int32_t* list, long list_stride,
                              int32_t* hit, double* tuv, int32_t* ovf, int* counters, hipStream_t st)
{
    const long npix = (long)W * rows.rows;
    if (npix <= 0) return RT_OK;
    long wgs = (npix + RT_WG - 1) / RT_WG;
    if (wgs > RESOLVE_GRID) wgs = RESOLVE_GRID;
    hipLaunchKernelGGL(k_resolve_primary, dim3((unsigned)wgs), dim3(RT_WG),
                       0, st, es, W, H, tan_half, rows, cnt, list, list_stride, hit, tuv, ovf, counters);
    return launched("resolve_primary");
}

int rt_launch_resolve_rays(const ExactScene& es, int n, const double* S, const double* E,
                           const int32_t* cnt, const int32_t* list, bool anyhit, int32_t* hit,
                           double* tuv, double* point, uint8_t* occ, int32_t* ovf, int* counters,
                           const int32_t* gate, long gate_n, hipStream_t st)
{
    if (n <= 0) return RT_OK;
    if (gate && gate_n <= 0) gate = nullptr;
    dim3 grid((n + RT_WG - 1) / RT_WG < RESOLVE_GRID ? (n + RT_WG - 1) / RT_WG : RESOLVE_GRID);
    if (anyhit)
        hipLaunchKernelGGL(k_resolve_rays<true>, grid, dim3(RT_WG), 0, st, es, n, S, E, cnt, list,
                           hit, tuv, point, occ, ovf, counters, gate, gate_n);
    else
        hipLaunchKernelGGL(k_resolve_rays<false>, grid, dim3(RT_WG), 0, st, es, n, S, E, cnt, list,
                           hit, tuv, point, occ, ovf, counters, gate, gate_n);
    return launched("resolve_rays");
}

static_assert(sizeof(Partial) * OVERFLOW_GRID == RT_OVERFLOW_PARTIAL_BYTES, "partial-winner buffer size");

template <class RAYS, bool ANYHIT>
static int launch_overflow(const ExactScene& es, const RAYS& rays, const int32_t* ovf, const int* counters,
                           int32_t* hit, double* tuv, double* point, uint8_t* occ, void* partials,
                           hipStream_t st)
{
    Partial* part = static_cast<Partial*>(partials);
    if (!part) {
        rt_set_error("overflow: no buffer for the partial winners");
        return RT_E_ARG;
    }
    int rc;
    hipLaunchKernelGGL((k_overflow<RAYS, ANYHIT>), dim3(OVERFLOW_GRID), dim3(RT_WG), 0, st, es, rays, ovf,
                       counters, part, hit, tuv, point, occ);
    rc = launched("overflow");
    if (rc) return rc;
    // merges at most OVERFLOW_GRID / 2 rays (more rays than that take the direct path)
    hipLaunchKernelGGL((k_overflow_merge<RAYS, ANYHIT>), dim3(OVERFLOW_GRID / 2 / RT_WG), dim3(RT_WG), 0, st,
                       rays, ovf, counters, part, hit, tuv, point, occ);
    return launched("overflow_merge");
}

int rt_launch_overflow_primary(const ExactScene& es, int W, int H, double tan_half, const RowMap& rows,
                               const int32_t* ovf, const int* counters, int32_t* hit, double* tuv,
                               void* partials, hipStream_t st)
{
    const PrimaryRays rays{W, H, tan_half, rows};
    return launch_overflow<PrimaryRays, false>(es, rays, ovf, counters, hit, tuv, nullptr, nullptr, partials, st);
}

int rt_launch_overflow_rays(const ExactScene& es, const double* S, const double* E,
                            const int32_t* ovf, const int* counters, bool anyhit, int32_t* hit,
                            double* tuv, double* point, uint8_t* occ, void* partials, hipStream_t st)
{
    const ListedRays rays{S, E};
    return anyhit ? launch_overflow<ListedRays, true>(es, rays, ovf, counters, hit, tuv, point, occ, partials, st)
                  : launch_overflow<ListedRays, false>(es, rays, ovf, counters, hit, tuv, point, occ, partials, st);
}

int rt_launch_try_cached(const ExactScene& es, const double* S, const double* E, const int32_t* lists,
                         const int* counts, int sets, int n_max, long stride, int W, int rows,
                         int row_block, int step, bool record, int32_t* cnt, int32_t* occ_id, hipStream_t st)
{
    if (n_max <= 0 || sets <= 0) return RT_OK;
    long wgs = ((long)n_max + RT_WG - 1) / RT_WG;
    if (wgs > 4096) wgs = 4096;
    hipLaunchKernelGGL(k_try_cached, dim3((unsigned)wgs, sets), dim3(RT_WG), 0, st, es, S, E, lists, counts,
                       stride, W, rows, row_block > 0 ? row_block : rows, step, record ? 1 : 0, cnt, occ_id);
    return launched("try_cached");
}

// out_dev: 7 doubles followed by SINGLE_RAY_GRID Partial records of workspace.
int rt_launch_single_ray(const ExactScene& es, const double* SE_dev, double* out_dev, hipStream_t st)
{
    Partial* part = reinterpret_cast<Partial*>(out_dev + 8);
    int grid = (es.num_tris + RT_WG - 1) / RT_WG;
    if (grid > SINGLE_RAY_GRID) grid = SINGLE_RAY_GRID;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(k_single_ray_partial, dim3(grid), dim3(RT_WG), 0, st, es, SE_dev, part);
    int rc = launched("single_ray_partial");
    if (rc) return rc;
    hipLaunchKernelGGL(k_single_ray_final, dim3(1), dim3(64), 0, st, part, grid, SE_dev, out_dev);
    return launched("single_ray_final");
}
