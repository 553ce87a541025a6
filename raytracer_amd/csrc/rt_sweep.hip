// rt_sweep.hip -- the hot kernel: brute-force ray-triangle sweep for gfx950.
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
// and u = U/det, v = V/det, t = T/det.  Three evaluation orders are kept (RT_SWEEP_FORM):
//   0  textbook Moller-Trumbore: pvec = D x e2, qvec = tvec x e1            (31 instr / pair)
//   1  shared cross product c = tvec x D: U = e2.c, V = -e1.c, det = D.nn with the stored
//      normal nn = e2 x e1; lazy: u test first                              (20 instr / pair)
//   2  moment form: by linearity in tvec = O - v0, with the ray moment M = O x D kept per ray
//      and g = e2 x v0, h = e1 x v0, k = v0.nn stored per triangle,
//          U = e2.M - D.g      V = D.h - e1.M      T = k - O.nn      det = D.nn
//      lazy: u test first                                                   (14 instr / pair)
// The divisions are replaced by sign-consistent products: with a = U*det, b = V*det,
// w = det*det - a - b a pair is inside when min(a, b, w) >= 0.  "Lazy" is the SIMD counterpart
// of the early returns of the reference kernel (:67-94): V, the v and u+v tests and T are only
// evaluated where some lane of the wave is inside the u slab.
//
// The fp32 pass is a conservative FILTER, not the final word: a pair passes when it lies
// inside the triangle widened by a per-triangle rounding bound (eu, et; see rt_transform.hip).
// Passing pairs are appended to a short per-ray candidate list in ascending triangle order;
// rt_exact.hip then re-evaluates only those candidates with the reference's exact fp64
// arithmetic and comparison rules, so hit indices equal the fp64 result wherever the filter
// bound holds.  A per-lane running minimum of t (updated only by pairs that are inside by
// the same bound, i.e. certain hits) prunes candidates that cannot win.
//
// Any-hit (shadow) sweeps retire a ray at its first certain hit.  Because the workgroup only
// stops when all of its rays have retired, long sweeps are cut into triangle ranges and the
// surviving rays are compacted between ranges (k_compact), so the work per ray is
// proportional to the index of its first occluder instead of the triangle count.
#include "rt_internal.h"

namespace {

constexpr float RT_INF = __builtin_huge_valf();

struct Ray {
#if RT_SWEEP_FORM == 2
    float mx, my, mz;   // moment O x D (the origin itself is re-read in the rare path)
#else
    float ox, oy, oz;
#endif
    float dx, dy, dz;   // unit direction
    float best;         // smallest certain t so far (closest-hit mode)
    float best_hi;      // pruning threshold: best + its error bound; -inf = ray retired/invalid
    int cnt;            // candidates pushed so far, plus the RT_CNT_* flag bits
};

// One candidate that passed the widened barycentric test.  Rare: a few per ray per sweep.
template <bool ANYHIT>
__device__ __forceinline__ void rare_candidate(Ray& r, int32_t* __restrict__ list_slot, int tri,
                                               float a, float b, float w, float d, float T,
                                               float eu, float et)
{
    const float rd = 1.0f / d;
    const float t = T * rd;
    const float terr = et * __builtin_fabsf(rd) + 2e-6f * __builtin_fabsf(t);
    if (!(t >= -terr) || !(t - terr <= r.best_hi)) return;
    const int slot = r.cnt & RT_CNT_MASK;
    if (slot < RT_LIST_K) list_slot[slot] = tri;
    r.cnt++;
    const float margin = eu * __builtin_fabsf(d);
    const bool certain = (__builtin_fminf(__builtin_fminf(a, b), w) >= margin) && (t >= terr);
    if (certain) {
        if (ANYHIT) {
            r.cnt |= RT_CNT_OCCLUDED;
            r.best_hi = -RT_INF;
        } else if (t < r.best) {
            r.best = t;
            r.best_hi = t + terr;
        }
    }
}

__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz)
{
    return __builtin_fmaf(az, bz, __builtin_fmaf(ay, by, ax * bx));
}

// ---- one pair, per formulation.  `tile` points at entry j of plane 0 of the current LDS
// buffer; plane p of the same triangle is RT_TILE entries further per plane.
#if RT_SWEEP_FORM == 0
// planes: 0 = (A, eu)   1 = (e1, et)   2 = (e2, 0)
struct Tri { float4 V0, E1, E2; };
__device__ __forceinline__ Tri load_tri(const float4* p) { return {p[0], p[RT_TILE], p[2 * RT_TILE]}; }

template <bool ANYHIT>
__device__ __forceinline__ void pair(Ray& r, int32_t* __restrict__ list_slot, int tri, const Tri& q,
                                     const float4*, const double*)
{
    const float tx = r.ox - q.V0.x, ty = r.oy - q.V0.y, tz = r.oz - q.V0.z;
    const float px = __builtin_fmaf(r.dy, q.E2.z, -(r.dz * q.E2.y));
    const float py = __builtin_fmaf(r.dz, q.E2.x, -(r.dx * q.E2.z));
    const float pz = __builtin_fmaf(r.dx, q.E2.y, -(r.dy * q.E2.x));
    const float d = dot3(q.E1.x, q.E1.y, q.E1.z, px, py, pz);
    const float U = dot3(tx, ty, tz, px, py, pz);
    const float qx = __builtin_fmaf(ty, q.E1.z, -(tz * q.E1.y));
    const float qy = __builtin_fmaf(tz, q.E1.x, -(tx * q.E1.z));
    const float qz = __builtin_fmaf(tx, q.E1.y, -(ty * q.E1.x));
    const float V = dot3(r.dx, r.dy, r.dz, qx, qy, qz);
    const float a = U * d, b = V * d;
    const float w = __builtin_fmaf(d, d, -a) - b;
    const float m = __builtin_fminf(__builtin_fminf(a, b), w);      // v_min3_f32
    if (__builtin_expect(__builtin_fmaf(q.V0.w, __builtin_fabsf(d), m) > 0.0f, 0)) {
        const float T = dot3(q.E2.x, q.E2.y, q.E2.z, qx, qy, qz);
        rare_candidate<ANYHIT>(r, list_slot, tri, a, b, w, d, T, q.V0.w, q.E1.w);
    }
}
#elif RT_SWEEP_FORM == 1
// planes: 0 = (A, eu)   1 = (e2, 2eu)   2 = (nn, 0)   3 = (e1, et)      [3: u slab only]
struct Tri { float4 V0, E2, NN; };
__device__ __forceinline__ Tri load_tri(const float4* p) { return {p[0], p[RT_TILE], p[2 * RT_TILE]}; }

template <bool ANYHIT>
__device__ __forceinline__ void pair(Ray& r, int32_t* __restrict__ list_slot, int tri, const Tri& q,
                                     const float4* __restrict__ p, const double*)
{
    const float tx = r.ox - q.V0.x, ty = r.oy - q.V0.y, tz = r.oz - q.V0.z;
    const float cx = __builtin_fmaf(ty, r.dz, -(tz * r.dy));       // c = tvec x D
    const float cy = __builtin_fmaf(tz, r.dx, -(tx * r.dz));
    const float cz = __builtin_fmaf(tx, r.dy, -(ty * r.dx));
    const float d = dot3(r.dx, r.dy, r.dz, q.NN.x, q.NN.y, q.NN.z);
    const float U = dot3(q.E2.x, q.E2.y, q.E2.z, cx, cy, cz);
    const float a = U * d;
    const float w1 = __builtin_fmaf(d, d, -a);
    // u slab, widened by 2*eu: u <= 1 follows from w >= -eu|d| and b >= -eu|d| only up to 2*eu|d|
    if (__builtin_expect(__builtin_fmaf(q.E2.w, __builtin_fabsf(d), __builtin_fminf(a, w1)) > 0.0f, 0)) {
        const float4 E1 = p[3 * RT_TILE];
        const float b = dot3(E1.x, E1.y, E1.z, cx, cy, cz) * -d;   // V = -e1.c
        const float w = w1 - b;
        const float m = __builtin_fminf(__builtin_fminf(a, b), w);
        if (__builtin_fmaf(q.V0.w, __builtin_fabsf(d), m) > 0.0f) {
            const float T = -dot3(tx, ty, tz, q.NN.x, q.NN.y, q.NN.z);
            rare_candidate<ANYHIT>(r, list_slot, tri, a, b, w, d, T, q.V0.w, E1.w);
        }
    }
}
#else
// planes: 0 = (nn, 2eu)   1 = (g = e2 x v0, eu)   2 = (e2, et)   3 = (e1, k = v0.nn)
//         4 = (h = e1 x v0, 0)                                          [3, 4: u slab only]
struct Tri { float4 NN, G, E2; };
__device__ __forceinline__ Tri load_tri(const float4* p) { return {p[0], p[RT_TILE], p[2 * RT_TILE]}; }

// u-slab test of one pair: 14 VALU instructions.
__device__ __forceinline__ bool in_u_slab(const Ray& r, const Tri& q, float& d, float& a, float& w1)
{
    d = dot3(r.dx, r.dy, r.dz, q.NN.x, q.NN.y, q.NN.z);
    // U = e2.M - D.g, one six-term FMA chain
    float U = r.mx * q.E2.x;
    U = __builtin_fmaf(r.my, q.E2.y, U);
    U = __builtin_fmaf(r.mz, q.E2.z, U);
    U = __builtin_fmaf(-r.dx, q.G.x, U);
    U = __builtin_fmaf(-r.dy, q.G.y, U);
    U = __builtin_fmaf(-r.dz, q.G.z, U);
    a = U * d;
    w1 = __builtin_fmaf(d, d, -a);
    return __builtin_fmaf(q.NN.w, __builtin_fabsf(d), __builtin_fminf(a, w1)) > 0.0f;
}

// The rest of the test for a pair inside the u slab (a few percent of the wave-pairs).
template <bool ANYHIT>
__device__ __forceinline__ void finish_pair(Ray& r, int32_t* __restrict__ list_slot, int tri,
                                            const Tri& q, const float4* __restrict__ p,
                                            const double* __restrict__ origin, float d, float a,
                                            float w1)
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
        // T = k - O.nn; the origin is not kept in registers (primary rays: O = 0)
        float ox = 0.0f, oy = 0.0f, oz = 0.0f;
        if (origin) { ox = (float)origin[0]; oy = (float)origin[1]; oz = (float)origin[2]; }
        const float T = E1.w - dot3(ox, oy, oz, q.NN.x, q.NN.y, q.NN.z);
        rare_candidate<ANYHIT>(r, list_slot, tri, a, b, w, d, T, q.G.w, q.E2.w);
    }
}

template <bool ANYHIT>
__device__ __forceinline__ void pair(Ray& r, int32_t* __restrict__ list_slot, int tri, const Tri& q,
                                     const float4* __restrict__ p, const double* __restrict__ origin)
{
    float d, a, w1;
    if (__builtin_expect(in_u_slab(r, q, d, a, w1), 0))
        finish_pair<ANYHIT>(r, list_slot, tri, q, p, origin, d, a, w1);
}
#endif

// Makes a value opaque to the optimiser so that per-ray quantities that happen to be equal
// in every lane (the shared origin of primary rays) are still carried and used per ray.
__device__ __forceinline__ float opaque(float x)
{
    asm volatile("" : "+v"(x));
    return x;
}

enum { MODE_PRIMARY = 0, MODE_RAYS_LINEAR = 1, MODE_RAYS_TILED = 2, MODE_RAYS_INDEXED = 3 };

template <int MODE, bool ANYHIT>
__global__ void __launch_bounds__(RT_WG, RT_MIN_WAVES) k_sweep(const SweepArgs A)
{
    // double-buffered planes of one tile
    __shared__ float4 tile[2][RT_PLANES][RT_TILE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;

    Ray ray[RT_RPL];
    int slot[RT_RPL];         // index into cnt[] / list[] (band-local pixel or ray number, < 2^31)
    bool in_range[RT_RPL];

    // ---- ray set-up ------------------------------------------------------------------
#pragma unroll
    for (int r = 0; r < RT_RPL; ++r) {
        bool valid;
        double Sx = 0, Sy = 0, Sz = 0, Ex = 0, Ey = 0, Ez = 0;
        if (MODE == MODE_RAYS_LINEAR) {
            const long j = (long)blockIdx.x * RT_RAYS_PER_WG + r * RT_WG + tid;
            valid = j < A.n_rays;
            slot[r] = valid ? (int)j : 0;
        } else if (MODE == MODE_RAYS_INDEXED) {
            const long j = (long)blockIdx.x * RT_RAYS_PER_WG + r * RT_WG + tid;
            valid = j < A.n_rays;
            slot[r] = valid ? A.active[j] : 0;
        } else {
            // workgroup = RT_TILE_W x RT_TILE_H pixels; wave = (8*RPL) x 8; lane = 8 x 8
            const int tiles_x = (A.W + RT_TILE_W - 1) / RT_TILE_W;
            const int tyi = blockIdx.x / tiles_x, txi = blockIdx.x - tyi * tiles_x;
            const int px = txi * RT_TILE_W + (wave & 1) * (8 * RT_RPL) + r * 8 + (lane & 7);
            const int py = A.y0 + tyi * RT_TILE_H + (wave >> 1) * 8 + (lane >> 3);
            valid = (px < A.W) && (py < A.y1);
            const long npix = (long)A.W * (A.y1 - A.y0);
            slot[r] = valid ? (int)((long)blockIdx.y * npix + (long)(py - A.y0) * A.W + px) : 0;
            if (MODE == MODE_PRIMARY) {
                // driver.c:131-133 with S = origin = 0 (driver.c:121-123)
                Ex = (double)(px - A.W / 2);
                Ey = (double)(py - A.H / 2);
                Ez = A.ez;
            }
        }
        in_range[r] = valid;
        if (MODE != MODE_PRIMARY && valid) {
            const double* s = A.S + 3 * (long)slot[r];
            const double* e = A.E + 3 * (long)slot[r];
            Sx = s[0]; Sy = s[1]; Sz = s[2];
            Ex = e[0]; Ey = e[1]; Ez = e[2];
            if (!(Sx == Sx)) valid = false;       // NaN origin marks an unused ray slot
        }
        // D is normalised: t scales by |D| for every triangle alike, so the per-ray order of
        // candidates is unchanged and the per-triangle error bounds need no |D| factor.
        double Dx = Ex - Sx, Dy = Ey - Sy, Dz = Ez - Sz;
        const double inv = 1.0 / __builtin_sqrt(Dx * Dx + Dy * Dy + Dz * Dz);
        if (!(inv == inv) || inv > 1e300) valid = false;      // zero-length ray hits nothing
        if (!valid) { Sx = Sy = Sz = 0; Dx = Dy = Dz = 0; } else { Dx *= inv; Dy *= inv; Dz *= inv; }
        ray[r].dx = (float)Dx;
        ray[r].dy = (float)Dy;
        ray[r].dz = (float)Dz;
#if RT_SWEEP_FORM == 2
        // moment in fp64 from the fp64 ray, rounded once; opaque so that primary rays (M = 0 in
        // every lane) still execute the general test
        ray[r].mx = opaque((float)(Sy * Dz - Sz * Dy));
        ray[r].my = opaque((float)(Sz * Dx - Sx * Dz));
        ray[r].mz = opaque((float)(Sx * Dy - Sy * Dx));
#else
        ray[r].ox = opaque((float)Sx);
        ray[r].oy = opaque((float)Sy);
        ray[r].oz = opaque((float)Sz);
#endif
        ray[r].best = RT_INF;
        ray[r].best_hi = valid ? RT_INF : -RT_INF;
        ray[r].cnt = valid ? 0 : RT_CNT_RETIRED;
        if (A.resume && in_range[r]) {
            // a later triangle range of the same sweep: pick up the candidate count
            ray[r].cnt = A.cnt[slot[r]];
            if (ray[r].cnt & (RT_CNT_OCCLUDED | RT_CNT_RETIRED)) ray[r].best_hi = -RT_INF;
        }
    }

    // ---- tile loop -------------------------------------------------------------------
    const int t_begin = A.tri_begin / RT_TILE, t_end = A.tri_end / RT_TILE;
    auto stage = [&](int t, int buf) {
        // RT_PLANES x RT_TILE float4; one LDS-DMA instruction moves 64 float4 (1 KiB) per wave:
        // global source per lane, LDS destination = wave-uniform base + 16 * lane.
#pragma unroll
        for (int c = 0; c < (RT_PLANES * (RT_TILE / 64) + 3) / 4; ++c) {
            const int chunk = c * 4 + wave;                 // chunk = plane * (RT_TILE/64) + part
            if (chunk < RT_PLANES * (RT_TILE / 64)) {
                const int plane = chunk / (RT_TILE / 64), part = chunk % (RT_TILE / 64);
                const long g = (long)plane * A.plane_stride + (long)t * RT_TILE + part * 64 + lane;
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(A.planes + g),
                    (__attribute__((address_space(3))) void*)&tile[buf][plane][part * 64], 16, 0, 0);
            }
        }
    };

    unsigned long long tiles_done = 0;
    if (t_begin < t_end) {
        stage(t_begin, 0);
        __builtin_amdgcn_s_waitcnt(0);      // vmcnt(0) & lgkmcnt(0)
        __syncthreads();
    }
    for (int t = t_begin; t < t_end; ++t) {
        const int buf = (t - t_begin) & 1;
        if (t + 1 < t_end) stage(t + 1, buf ^ 1);
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
#if RT_SWEEP_FORM == 2 && RT_MERGED_BRANCH
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
                        finish_pair<ANYHIT>(ray[r], A.list + (long)slot[r] * RT_LIST_K, tri0 + j, cur, base + j,
                                            MODE == MODE_PRIMARY ? nullptr : A.S + 3 * (long)slot[r], d, a, w1);
                }
            }
#else
#pragma unroll
            for (int r = 0; r < RT_RPL; ++r)
                pair<ANYHIT>(ray[r], A.list + (long)slot[r] * RT_LIST_K, tri0 + j, cur, base + j,
                             MODE == MODE_PRIMARY ? nullptr : A.S + 3 * (long)slot[r]);
#endif
        }
        ++tiles_done;
        __builtin_amdgcn_s_waitcnt(0);
        if (ANYHIT) {
            // retire the workgroup once every ray in it is certainly occluded or unused
            bool done = true;
#pragma unroll
            for (int r = 0; r < RT_RPL; ++r) done = done && (ray[r].best_hi == -RT_INF);
            if (__syncthreads_and(done)) break;
        } else {
            __syncthreads();
        }
    }

    // ---- results ---------------------------------------------------------------------
#pragma unroll
    for (int r = 0; r < RT_RPL; ++r)
        if (in_range[r]) A.cnt[slot[r]] = ray[r].cnt;
    if (A.tests && tid == 0)
        atomicAdd(A.tests, tiles_done * (unsigned long long)RT_TILE * RT_RAYS_PER_WG);
}

// Survivors of an any-hit range: rays that are neither certainly occluded nor unused.
// `in` == nullptr means "all rays 0..n-1".  Each wave reserves one contiguous block of the
// output, so rays that shared a wave stay together.
__global__ void __launch_bounds__(RT_WG)
k_compact(const int32_t* __restrict__ cnt, const int32_t* __restrict__ in, int n,
          int32_t* __restrict__ out, int* __restrict__ out_count)
{
    const long j = (long)blockIdx.x * RT_WG + threadIdx.x;
    int id = -1;
    if (j < n) {
        id = in ? in[j] : (int)j;
        if (cnt[id] & (RT_CNT_OCCLUDED | RT_CNT_RETIRED)) id = -1;
    }
    const unsigned long long m = __ballot(id >= 0);
    const int lane = threadIdx.x & 63;
    int base = 0;
    if (lane == 0 && m) base = atomicAdd(out_count, __popcll(m));
    base = __shfl(base, 0);
    if (id >= 0) out[base + __popcll(m & ((1ull << lane) - 1))] = id;
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

static int check_common(const SweepArgs& a)
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
    return RT_OK;
}

int rt_launch_sweep_primary(const SweepArgs& a, hipStream_t st)
{
    int rc = check_common(a);
    if (rc) return rc;
    if (a.W <= 0 || a.H <= 0 || a.y1 <= a.y0) {
        rt_set_error("sweep: empty band");
        return RT_E_ARG;
    }
    const int tiles_x = (a.W + RT_TILE_W - 1) / RT_TILE_W;
    const int tiles_y = (a.y1 - a.y0 + RT_TILE_H - 1) / RT_TILE_H;
    return launch<MODE_PRIMARY, false>(a, dim3(tiles_x * tiles_y, 1), st);
}

int rt_launch_sweep_rays(const SweepArgs& a, bool anyhit, hipStream_t st)
{
    int rc = check_common(a);
    if (rc) return rc;
    if (!a.S || !a.E) {
        rt_set_error("sweep: null ray arrays");
        return RT_E_ARG;
    }
    if (a.n_rays <= 0) return RT_OK;
    if (a.active) {     // indexed: n_rays entries of `active` name the rays
        dim3 grid((a.n_rays + RT_RAYS_PER_WG - 1) / RT_RAYS_PER_WG, 1);
        return anyhit ? launch<MODE_RAYS_INDEXED, true>(a, grid, st)
                      : launch<MODE_RAYS_INDEXED, false>(a, grid, st);
    }
    if (a.W > 0) {      // tiled: n_rays = layers * W * (y1-y0)
        const long npix = (long)a.W * (a.y1 - a.y0);
        if (npix <= 0 || a.n_rays % npix != 0) {
            rt_set_error("sweep: tiled ray count must be a multiple of the band size");
            return RT_E_ARG;
        }
        const int layers = (int)(a.n_rays / npix);
        const int tiles_x = (a.W + RT_TILE_W - 1) / RT_TILE_W;
        const int tiles_y = (a.y1 - a.y0 + RT_TILE_H - 1) / RT_TILE_H;
        dim3 grid(tiles_x * tiles_y, layers);
        return anyhit ? launch<MODE_RAYS_TILED, true>(a, grid, st)
                      : launch<MODE_RAYS_TILED, false>(a, grid, st);
    }
    dim3 grid((a.n_rays + RT_RAYS_PER_WG - 1) / RT_RAYS_PER_WG, 1);
    return anyhit ? launch<MODE_RAYS_LINEAR, true>(a, grid, st)
                  : launch<MODE_RAYS_LINEAR, false>(a, grid, st);
}

int rt_launch_compact(const int32_t* cnt, const int32_t* in, int n, int32_t* out, int* out_count,
                      hipStream_t st)
{
    if (n <= 0) return RT_OK;
    hipLaunchKernelGGL(k_compact, dim3((n + RT_WG - 1) / RT_WG), dim3(RT_WG), 0, st, cnt, in, n, out,
                       out_count);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        rt_set_error("compact launch failed: %s", hipGetErrorString(e));
        return RT_E_HIP;
    }
    return RT_OK;
}
