// rt_sweep.hip -- brute-force ray-triangle sweep for rays in general position (the batched
// counterpart of intersect_all_triangles_device for arbitrary S, E; the two ray families of
// the pixel loop have cheaper kernels of their own in rt_sweep_apex.hip).
//
// Replaces the per-ray launch of intersect_single_triangle_device
// (triangle_intersection.cu:31-96,145-150: one thread per triangle, one launch per ray) by
// the inverse mapping: every lane owns RT_RPL rays and sweeps the triangles, which arrive
// as RT_PLANES float4 planes (built by k_pack, rt_transform.hip) staged through LDS in tiles
// of RT_TILE triangles that all four waves of the workgroup read at wave-uniform addresses
// (LDS broadcast, no bank conflicts).
//
// Per (ray, triangle) pair the kernel evaluates, in fp32 FMA, the determinants of the
// reference's Cramer solve (:59-94).  With D = E - S, tvec = S - A, e1 = B - A, e2 = C - A:
//     det = e1.(D x e2)     U = tvec.(D x e2)     V = D.(tvec x e1)     T = e2.(tvec x e1)
// and u = U/det, v = V/det, t = T/det.  By linearity in tvec = O - v0, with the ray moment
// M = O x D kept per ray and nn = e2 x e1, g = e2 x v0, h = e1 x v0, k = v0.nn stored per
// triangle (the "moment form"),
//     U = e2.M - D.g      V = D.h - e1.M      T = k - O.nn      det = D.nn
// The divisions are replaced by sign-consistent products: with a = U*det, b = V*det,
// w = det*det - a - b a pair is inside when min(a, b, w) >= 0.  Evaluation is lazy, the SIMD
// counterpart of the early returns of the reference kernel (:67-94): det, U and the u-slab
// test come first (14 VALU instructions per pair); V, the v and u+v tests and T are only
// evaluated where some lane of the wave is inside the u slab.
//
// The moment depends only on the ray's line, so O is taken at the point of the line closest
// to the coordinate origin: |M| is then the distance of the line from the origin, a line
// farther away than the farthest vertex coordinate allows cannot hit anything (the ray is
// retired at set-up), and every other ray has |M| within the range the per-triangle rounding
// bounds assume, wherever its S lies.
//
// The fp32 pass is a conservative FILTER, not the final word: a pair passes when it lies
// inside the triangle widened by a per-triangle rounding bound (eu, et; see rt_transform.hip).
// Passing pairs are appended to a short per-ray candidate list in ascending triangle order;
// rt_exact.hip then re-evaluates only those candidates with the reference's exact fp64
// arithmetic and comparison rules, so hit indices equal the fp64 result wherever the filter
// bound holds.  A per-lane running minimum of t (updated only by pairs that are inside by
// the same bound, i.e. certain hits) prunes candidates that cannot win.
//
// Any-hit sweeps retire a ray at its first certain hit.  Because the workgroup only stops
// when all of its rays have retired, long sweeps are cut into triangle ranges and the
// surviving rays are compacted between ranges (k_compact), so the work per ray is
// proportional to the index of its first occluder instead of the triangle count.
#include "rt_sweep_common.h"

#ifndef RT_SWEEP_WAVES
#define RT_SWEEP_WAVES 5    // waves per SIMD the kernel must fit (96 registers: the ray set-up and the rare path
                            // spill, the hot loop does not; measured on 8.3 M rays x 100 k triangles: 217 ms
                            // at three waves (150 registers), 201 at four, 198 at five)
#endif

namespace {

using namespace rt_sweep;

struct Ray {
    float mx, my, mz;   // moment O x D
    float dx, dy, dz;   // unit direction
};

// planes: 0 = (nn, 2eu)   1 = (-g, eu), g = e2 x v0   2 = (e2, et)   3 = (e1, k = v0.nn)
//         4 = (h = e1 x v0, 0)                                          [3, 4: u slab only]
struct Tri { float4 NN, G, E2; };
__device__ __forceinline__ Tri load_tri(const float4* p) { return {p[0], p[RT_TILE], p[2 * RT_TILE]}; }

// u-slab test of one pair: 14 VALU instructions.
__device__ __forceinline__ bool in_u_slab(const Ray& r, const Tri& q, float& d, float& a, float& w1)
{
    d = dot3(r.dx, r.dy, r.dz, q.NN.x, q.NN.y, q.NN.z);
    // U = e2.M - D.g, one six-term FMA chain (the plane holds -g: a negated operand would cost every
    // one of these three the longer encoding, which issues more slowly)
    float U = r.mx * q.E2.x;
    U = __builtin_fmaf(r.my, q.E2.y, U);
    U = __builtin_fmaf(r.mz, q.E2.z, U);
    U = __builtin_fmaf(r.dx, q.G.x, U);
    U = __builtin_fmaf(r.dy, q.G.y, U);
    U = __builtin_fmaf(r.dz, q.G.z, U);
    a = U * d;
    w1 = __builtin_fmaf(d, d, -a);
    return __builtin_fmaf(q.NN.w, __builtin_fabsf(d), __builtin_fminf(a, w1)) > 0.0f;
}

// The rest of the test for a pair inside the u slab (a few percent of the wave-pairs).
template <bool ANYHIT>
__device__ __forceinline__ void finish_pair(Book& bk, const Ray& r, int32_t* __restrict__ list_slot,
                                            int tri, const Tri& q, const float4* __restrict__ p,
                                            const double* __restrict__ S, float d, float a, float w1)
{
    const float4 E1 = p[3 * RT_TILE], H = p[4 * RT_TILE];
    float V = r.dx * H.x;                                       // V = D.h - e1.M
    V = __builtin_fmaf(r.dy, H.y, V);
    V = __builtin_fmaf(r.dz, H.z, V);
    V = __builtin_fmaf(-r.mx, E1.x, V);
    V = __builtin_fmaf(-r.my, E1.y, V);
    V = __builtin_fmaf(-r.mz, E1.z, V);
    const float b = V * d;
    const float w = w1 - b;
    const float m = __builtin_fminf(__builtin_fminf(a, b), w);
    if (__builtin_fmaf(q.G.w, __builtin_fabsf(d), m) > 0.0f) {
        // T' = k - O'.nn with O' = D x M, the point of the line closest to the origin; the
        // parameter from S is T'/det - S.D.  O' and S.D are not kept in registers.
        const float ox = __builtin_fmaf(r.dy, r.mz, -(r.dz * r.my));
        const float oy = __builtin_fmaf(r.dz, r.mx, -(r.dx * r.mz));
        const float oz = __builtin_fmaf(r.dx, r.my, -(r.dy * r.mx));
        const float T = E1.w - dot3(ox, oy, oz, q.NN.x, q.NN.y, q.NN.z);
        const float tau = (float)(S[0] * (double)r.dx + S[1] * (double)r.dy + S[2] * (double)r.dz);
        const float en = U20 * __builtin_sqrtf(dot3(q.NN.x, q.NN.y, q.NN.z, q.NN.x, q.NN.y, q.NN.z));
        candidate<ANYHIT>(bk, list_slot, tri, a, b, w, d, T, -tau, q.G.w, en, q.E2.w);
    }
}

enum { MODE_RAYS_LINEAR = 1, MODE_RAYS_INDEXED = 3 };

template <int MODE, bool ANYHIT>
__global__ void __launch_bounds__(RT_WG, RT_SWEEP_WAVES) k_sweep(const SweepArgs A)
{
    // double-buffered planes of one tile
    __shared__ float4 tile[2][RT_PLANES][RT_TILE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;

    Ray ray[RT_RPL];
    Book bk[RT_RPL];
    int slot[RT_RPL];         // index into cnt[] / list[] (ray number, < 2^31)
    bool in_range[RT_RPL];

    // ---- ray set-up ------------------------------------------------------------------
#pragma unroll
    for (int r = 0; r < RT_RPL; ++r) {
        const long j = (long)blockIdx.x * RT_RAYS_PER_WG + r * RT_WG + tid;
        bool valid = j < A.n_rays;
        if (MODE == MODE_RAYS_INDEXED) slot[r] = valid ? A.active[j] : 0;
        else slot[r] = valid ? (int)j : 0;
        in_range[r] = valid;
        double Sx = 0, Sy = 0, Sz = 0, Dx = 0, Dy = 0, Dz = 0;
        if (valid) {
            const double* s = A.S + 3 * (long)slot[r];
            const double* e = A.E + 3 * (long)slot[r];
            Sx = s[0]; Sy = s[1]; Sz = s[2];
            Dx = e[0] - Sx; Dy = e[1] - Sy; Dz = e[2] - Sz;
            if (!(Sx == Sx)) valid = false;       // NaN origin marks an unused ray slot
        }
        // D is normalised: t scales by |D| for every triangle alike, so the per-ray order of
        // candidates is unchanged and the per-triangle error bounds need no |D| factor.
        const double inv = 1.0 / __builtin_sqrt(Dx * Dx + Dy * Dy + Dz * Dz);
        if (!(inv == inv) || inv > 1e300) valid = false;      // zero-length ray hits nothing
        if (!valid) { Sx = Sy = Sz = 0; Dx = Dy = Dz = 0; } else { Dx *= inv; Dy *= inv; Dz *= inv; }
        // moment in fp64 from the fp64 ray, rounded once
        const double Mx = Sy * Dz - Sz * Dy, My = Sz * Dx - Sx * Dz, Mz = Sx * Dy - Sy * Dx;
        // |M| = distance of the line from the origin; every vertex lies within sqrt(3)*max|coord|
        if (Mx * Mx + My * My + Mz * Mz > 3.01 * A.scene_radius * A.scene_radius) valid = false;
        ray[r].dx = (float)Dx;
        ray[r].dy = (float)Dy;
        ray[r].dz = (float)Dz;
        ray[r].mx = valid ? (float)Mx : 0.0f;
        ray[r].my = valid ? (float)My : 0.0f;
        ray[r].mz = valid ? (float)Mz : 0.0f;
        bk[r].best = INF;
        bk[r].best_hi = valid ? INF : -INF;
        bk[r].cnt = valid ? 0 : RT_CNT_RETIRED;
        if (A.resume && in_range[r]) {
            // a later triangle range of the same sweep: pick up the candidate count
            bk[r].cnt = A.cnt[slot[r]];
            if (bk[r].cnt & (RT_CNT_OCCLUDED | RT_CNT_RETIRED)) bk[r].best_hi = -INF;
        }
    }

    // ---- tile loop -------------------------------------------------------------------
    const int t_begin = A.tri_begin / RT_TILE, t_end = A.tri_end / RT_TILE;
    unsigned long long tiles_done = 0;
    if (t_begin < t_end) {
        stage_tile<RT_PLANES>(A.planes, A.plane_stride, t_begin, tile[0], wave, lane);
        __builtin_amdgcn_s_waitcnt(0);      // vmcnt(0) & lgkmcnt(0)
        __syncthreads();
    }
    for (int t = t_begin; t < t_end; ++t) {
        const int buf = (t - t_begin) & 1;
        if (t + 1 < t_end) stage_tile<RT_PLANES>(A.planes, A.plane_stride, t + 1, tile[buf ^ 1], wave, lane);
        const float4* __restrict__ base = tile[buf][0];
        const int tri0 = t * RT_TILE;
        // The rare path splits the loop body into basic blocks, which stops the compiler from
        // hoisting the next triangle's LDS reads above the current triangle's arithmetic; the
        // reads are therefore issued one triangle ahead by hand.
        Tri next = load_tri(base);
#pragma unroll 2
        for (int j = 0; j < RT_TILE; ++j) {
            const Tri cur = next;
            next = load_tri(base + ((j + 1) & (RT_TILE - 1)));      // last iteration re-reads entry 0 (unused)
            // one exec-mask branch per triangle for all rays of the lane; the slab quantities
            // are recomputed inside (rare) so that only the predicates stay live across it
            bool in[RT_RPL], any = false;
#pragma unroll
            for (int r = 0; r < RT_RPL; ++r) {
                float d, a, w1;
                in[r] = in_u_slab(ray[r], cur, d, a, w1);
                any = any || in[r];
            }
            if (__builtin_expect(any, 0)) {
#pragma unroll
                for (int r = 0; r < RT_RPL; ++r) {
                    float d, a, w1;
                    if (in[r] && in_u_slab(ray[r], cur, d, a, w1))
                        finish_pair<ANYHIT>(bk[r], ray[r], A.list + (long)slot[r] * RT_LIST_K, tri0 + j,
                                            cur, base + j, A.S + 3 * (long)slot[r], d, a, w1);
                }
            }
        }
        ++tiles_done;
        __builtin_amdgcn_s_waitcnt(0);
        if (ANYHIT) {
            // retire the workgroup once every ray in it is certainly occluded or unused
            bool done = true;
#pragma unroll
            for (int r = 0; r < RT_RPL; ++r) done = done && (bk[r].best_hi == -INF);
            if (__syncthreads_and(done)) break;
        } else {
            __syncthreads();
        }
    }

    // ---- results ---------------------------------------------------------------------
#pragma unroll
    for (int r = 0; r < RT_RPL; ++r)
        if (in_range[r]) A.cnt[slot[r]] = bk[r].cnt;
    if (A.tests && tid == 0)
        atomicAdd(A.tests, tiles_done * (unsigned long long)RT_TILE * RT_RAYS_PER_WG);
}

// Survivors of an any-hit range: rays that are neither certainly occluded nor unused.
// `in` == nullptr means "all rays 0..n-1".  blockIdx.y = ray set (see SweepArgs).  Each
// workgroup owns one contiguous slice of the input.  WRITE == false: it counts its survivors
// into slice_counts[set][workgroup].  WRITE == true (a second launch of the same shape): its
// place in the output is the sum of the counts of the slices before it, and it writes its
// survivors there in input order.  The output keeps the order of the input throughout: rays
// sorted by direction stay sorted, and a wave's rays stay one narrow bundle.
constexpr int kCompactSlices = RT_COMPACT_SLICES;
template <bool WRITE>
__global__ void __launch_bounds__(RT_WG)
k_compact(const int32_t* __restrict__ cnt, const int32_t* __restrict__ in,
          const int* __restrict__ in_counts, int n, long stride, int cnt_div,
          int32_t* __restrict__ out, int* __restrict__ out_counts, int* __restrict__ slice_counts)
{
    __shared__ int s_wave[RT_WG / 64];
    __shared__ int s_before[RT_WG / 64], s_total[RT_WG / 64];
    const long set = blockIdx.y;
    cnt += (set / cnt_div) * stride;
    out += set * stride;
    slice_counts += set * kCompactSlices;
    if (in) {
        in += set * stride;
        if (in_counts) n = in_counts[set];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // slice of this workgroup, a multiple of the workgroup size
    const long per = ((long)(n + gridDim.x - 1) / gridDim.x + RT_WG - 1) / RT_WG * RT_WG;
    const long lo = per * blockIdx.x, hi = lo + per < n ? lo + per : n;
    // kCompactRounds entries per lane at a time, their ray numbers fetched together and then their
    // states (entry `lo` stands in beyond the slice's end: a fetch under a condition is waited for
    // inside its branch, and a round of them was two memory round trips in a row for every 256 rays)
    constexpr int kCompactRounds = 8, WAVES = RT_WG / 64;
    auto survivors = [&](long j0, int (&id)[kCompactRounds]) {
        int state[kCompactRounds];
#pragma unroll
        for (int r = 0; r < kCompactRounds; ++r) {
            const long j = j0 + (long)r * RT_WG + threadIdx.x;
            const long jc = j < hi ? j : lo;
            id[r] = in ? in[jc] : (int)jc;
        }
#pragma unroll
        for (int r = 0; r < kCompactRounds; ++r) state[r] = cnt[id[r]];
#pragma unroll
        for (int r = 0; r < kCompactRounds; ++r) {
            const bool live = j0 + (long)r * RT_WG + threadIdx.x < hi && !(state[r] & (RT_CNT_OCCLUDED | RT_CNT_RETIRED));
            id[r] = live ? id[r] : -1;
        }
    };
    if (!WRITE) {
        int mine = 0;
        for (long j0 = lo; j0 < hi; j0 += (long)kCompactRounds * RT_WG) {
            int id[kCompactRounds];
            survivors(j0, id);
#pragma unroll
            for (int r = 0; r < kCompactRounds; ++r) mine += id[r] >= 0;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off);
        if (lane == 0) s_wave[wave] = mine;
        __syncthreads();
        if (threadIdx.x == 0) {
            int total = 0;
#pragma unroll
            for (int k = 0; k < RT_WG / 64; ++k) total += s_wave[k];
            slice_counts[blockIdx.x] = total;       // 0 for a slice beyond the end
        }
        return;
    }
    // survivors in the slices before this one, and (first workgroup) in all of them
    int before = 0, total = 0;
    for (int k = threadIdx.x; k < (int)gridDim.x; k += RT_WG) {
        const int c = slice_counts[k];
        before += k < (int)blockIdx.x ? c : 0;
        total += c;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        before += __shfl_xor(before, off);
        total += __shfl_xor(total, off);
    }
    if (lane == 0) {
        s_before[wave] = before;
        s_total[wave] = total;
    }
    __syncthreads();
    int base = 0;
    total = 0;
#pragma unroll
    for (int k = 0; k < RT_WG / 64; ++k) {
        base += s_before[k];
        total += s_total[k];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) out_counts[set] = total;
    if (lo >= n) return;
    // a survivor's place: those of the rounds and waves before it (s_round[round][wave], summed
    // up by the first wave), then those of the lanes before it
    __shared__ int s_round[kCompactRounds * WAVES + 1];
    for (long j0 = lo; j0 < hi; j0 += (long)kCompactRounds * RT_WG) {
        int id[kCompactRounds], ahead[kCompactRounds];
        survivors(j0, id);
#pragma unroll
        for (int r = 0; r < kCompactRounds; ++r) {
            const unsigned long long m = __ballot(id[r] >= 0);
            if (lane == 0) s_round[r * WAVES + wave] = __popcll(m);
            ahead[r] = __popcll(m & ((1ull << lane) - 1));
        }
        __syncthreads();
        if (wave == 0) {
            const int c = lane < kCompactRounds * WAVES ? s_round[lane] : 0;
            int incl = c;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int v = __shfl_up(incl, off);
                incl += lane >= off ? v : 0;
            }
            if (lane < kCompactRounds * WAVES) s_round[lane] = incl - c;
            if (lane == 63) s_round[kCompactRounds * WAVES] = incl;      // the batch's survivors
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < kCompactRounds; ++r)
            if (id[r] >= 0) out[base + s_round[r * WAVES + wave] + ahead[r]] = id[r];
        base += s_round[kCompactRounds * WAVES];
        __syncthreads();                 // s_round is reused by the next batch
    }
}

template <int MODE, bool ANYHIT>
int launch(const SweepArgs& a, dim3 grid, hipStream_t st)
{
    hipLaunchKernelGGL((k_sweep<MODE, ANYHIT>), grid, dim3(RT_WG), 0, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        rt_set_error("sweep launch failed: %s", hipGetErrorString(e));
        return RT_E_HIP;
    }
    return RT_OK;
}

}  // namespace

int rt_launch_sweep_rays(const SweepArgs& a, bool anyhit, hipStream_t st)
{
    if (!a.planes || !a.cnt || !a.list) {
        rt_set_error("sweep: null buffer");
        return RT_E_ARG;
    }
    if (a.num_tris_padded <= 0 || a.num_tris_padded % RT_TILE != 0 ||
        a.num_tris > a.num_tris_padded || a.plane_stride < a.num_tris_padded) {
        rt_set_error("sweep: triangle planes must be padded to a multiple of %d", RT_TILE);
        return RT_E_ARG;
    }
    if (a.tri_begin < 0 || a.tri_end > a.num_tris_padded || a.tri_begin > a.tri_end ||
        a.tri_begin % RT_TILE != 0 || a.tri_end % RT_TILE != 0) {
        rt_set_error("sweep: triangle range [%d,%d) must be tile-aligned inside [0,%d]",
                     a.tri_begin, a.tri_end, a.num_tris_padded);
        return RT_E_ARG;
    }
    if (!a.S || !a.E) {
        rt_set_error("sweep: null ray arrays");
        return RT_E_ARG;
    }
    if (!(a.scene_radius >= 0.0)) {
        rt_set_error("sweep: scene radius not set");
        return RT_E_ARG;
    }
    if (a.n_rays <= 0) return RT_OK;
    dim3 grid((a.n_rays + RT_RAYS_PER_WG - 1) / RT_RAYS_PER_WG, 1);
    if (a.active)       // indexed: n_rays entries of `active` name the rays
        return anyhit ? launch<MODE_RAYS_INDEXED, true>(a, grid, st)
                      : launch<MODE_RAYS_INDEXED, false>(a, grid, st);
    return anyhit ? launch<MODE_RAYS_LINEAR, true>(a, grid, st)
                  : launch<MODE_RAYS_LINEAR, false>(a, grid, st);
}

int rt_launch_compact(const int32_t* cnt, const int32_t* in, const int* in_counts, int n, int sets,
                      long stride, int cnt_div, int32_t* out, int* out_counts, int* slice_counts,
                      hipStream_t st)
{
    if (n <= 0 || sets <= 0 || cnt_div <= 0) return RT_OK;
    if (!slice_counts || sets > RT_APEX_SETS) {
        rt_set_error("compact: no slice buffer");
        return RT_E_ARG;
    }
    long wgs = (n + RT_WG - 1) / RT_WG;
    if (wgs > kCompactSlices) wgs = kCompactSlices;     // grid-stride: a workgroup per 256 rays costs more to dispatch than to run
    hipLaunchKernelGGL(k_compact<false>, dim3((unsigned)wgs, sets), dim3(RT_WG), 0, st, cnt, in, in_counts, n,
                       stride, cnt_div, out, out_counts, slice_counts);
    hipLaunchKernelGGL(k_compact<true>, dim3((unsigned)wgs, sets), dim3(RT_WG), 0, st, cnt, in, in_counts, n,
                       stride, cnt_div, out, out_counts, slice_counts);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        rt_set_error("compact launch failed: %s", hipGetErrorString(e));
        return RT_E_HIP;
    }
    return RT_OK;
}
