// rt_sweep.hip -- the hot kernel: brute-force ray-triangle sweep for gfx950.
//
// Replaces the per-ray launch of intersect_single_triangle_device
// (triangle_intersection.cu:31-96,145-150: one thread per triangle, one launch per ray) by
// the inverse mapping: every lane owns RT_RPL rays and sweeps the triangles, which arrive
// as float4 planes (v0|eu, e1|et, e2, nn) staged through LDS in tiles of RT_TILE
// triangles that all four waves of the workgroup read at wave-uniform addresses
// (LDS broadcast, no bank conflicts).
//
// Arithmetic per (ray, triangle) pair: the general-origin Moller-Trumbore evaluation in fp32
// FMA -- tvec = O - v0, pvec = D x e2, det = e1.pvec, U = tvec.pvec, qvec = tvec x e1,
// V = D.qvec, T = e2.qvec -- which is the reference's Cramer solve (:59-94) with D = E - S,
// u = U/det, v = V/det, t = T/det.  The three divisions are replaced by sign-consistent
// products (a = U*det, b = V*det, c = det*det), so the hot loop is 32 VALU instructions per
// pair; T and the division are evaluated only for the rare pair that passes.
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
    float ox, oy, oz;
    float dx, dy, dz;
    float best;      // smallest certain t so far (closest-hit mode)
    float best_hi;   // pruning threshold: best + its error bound; -inf = ray retired/invalid
    int cnt;         // candidates pushed so far, plus the RT_CNT_* flag bits
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

// The pair evaluation.  Returns true when the pair needs the rare path.
//
// Both forms compute the same three scalar triple products
//     det = e1.(D x e2),  U = tvec.(D x e2),  V = D.(tvec x e1)        (tvec = O - v0)
// and defer the fourth, T = e2.(tvec x e1), to the rare path.
//   RT_SWEEP_FORM 0: the textbook Moller-Trumbore order (pvec = D x e2, qvec = tvec x e1):
//                    24 instructions for det, U, V.
//   RT_SWEEP_FORM 1: one shared cross product c = tvec x D, U = e2.c, V = -e1.c, and
//                    det = D.nn with the per-triangle nn = e2 x e1 precomputed by k_pack:
//                    18 instructions.  T = -tvec.nn.
// The acceptance test is division-free: with a = U*det, b = V*det, w = det*det - a - b the
// pair passes when min(a, b, w) + eu*|det| > 0.
[[maybe_unused]] __device__ __forceinline__ bool pair_test(const Ray& r, const float4 V0, const float4 E1,
                                          const float4 E2, const float4 NN, float& a, float& b,
                                          float& w, float& d, float& T)
{
    const float tx = r.ox - V0.x, ty = r.oy - V0.y, tz = r.oz - V0.z;
#if RT_SWEEP_FORM == 0
    const float px = __builtin_fmaf(r.dy, E2.z, -(r.dz * E2.y));
    const float py = __builtin_fmaf(r.dz, E2.x, -(r.dx * E2.z));
    const float pz = __builtin_fmaf(r.dx, E2.y, -(r.dy * E2.x));
    d = __builtin_fmaf(E1.z, pz, __builtin_fmaf(E1.y, py, E1.x * px));
    const float U = __builtin_fmaf(tz, pz, __builtin_fmaf(ty, py, tx * px));
    const float qx = __builtin_fmaf(ty, E1.z, -(tz * E1.y));
    const float qy = __builtin_fmaf(tz, E1.x, -(tx * E1.z));
    const float qz = __builtin_fmaf(tx, E1.y, -(ty * E1.x));
    const float V = __builtin_fmaf(r.dz, qz, __builtin_fmaf(r.dy, qy, r.dx * qx));
    T = __builtin_fmaf(E2.z, qz, __builtin_fmaf(E2.y, qy, E2.x * qx));
    (void)NN;
    a = U * d;
    b = V * d;
#else
    // c = tvec x D
    const float cx = __builtin_fmaf(ty, r.dz, -(tz * r.dy));
    const float cy = __builtin_fmaf(tz, r.dx, -(tx * r.dz));
    const float cz = __builtin_fmaf(tx, r.dy, -(ty * r.dx));
    d = __builtin_fmaf(r.dz, NN.z, __builtin_fmaf(r.dy, NN.y, r.dx * NN.x));
    const float U = __builtin_fmaf(E2.z, cz, __builtin_fmaf(E2.y, cy, E2.x * cx));
    const float Vn = __builtin_fmaf(E1.z, cz, __builtin_fmaf(E1.y, cy, E1.x * cx));   // = -V
    T = -__builtin_fmaf(tz, NN.z, __builtin_fmaf(ty, NN.y, tx * NN.x));
    a = U * d;
    b = -(Vn * d);
#endif
    w = __builtin_fmaf(d, d, -a) - b;
    const float m = __builtin_fminf(__builtin_fminf(a, b), w);      // v_min3_f32
    return __builtin_fmaf(V0.w, __builtin_fabsf(d), m) > 0.0f;
}

// RT_SWEEP_FORM 1 with lazy evaluation, the SIMD counterpart of the early returns in
// intersect_single_triangle_device (triangle_intersection.cu:67-94): the u test needs only
// tvec, c = tvec x D, det = D.nn and U = e2.c (20 instructions with its compare).  V = -e1.c,
// the v and u+v tests and finally T are evaluated only where some lane of the wave is inside the
// u slab, which for coherent rays and small triangles is a few percent of the wave-pairs.
// Tolerances: the u slab is widened by e2.w = 2*eu because u <= 1 is implied by the final
// w >= -eu*|det| and b >= -eu*|det| only up to 2*eu*|det|.
template <bool ANYHIT>
__device__ __forceinline__ void staged_pair(Ray& r, int32_t* __restrict__ list_slot, int tri,
                                            const float4 V0, const float4 E2, const float4 NN,
                                            const float4* __restrict__ pe1_j)
{
    const float tx = r.ox - V0.x, ty = r.oy - V0.y, tz = r.oz - V0.z;
    const float cx = __builtin_fmaf(ty, r.dz, -(tz * r.dy));
    const float cy = __builtin_fmaf(tz, r.dx, -(tx * r.dz));
    const float cz = __builtin_fmaf(tx, r.dy, -(ty * r.dx));
    const float d = __builtin_fmaf(r.dz, NN.z, __builtin_fmaf(r.dy, NN.y, r.dx * NN.x));
    const float U = __builtin_fmaf(E2.z, cz, __builtin_fmaf(E2.y, cy, E2.x * cx));
    const float a = U * d;
    const float w1 = __builtin_fmaf(d, d, -a);
    const float m1 = __builtin_fminf(a, w1);
    if (__builtin_expect(__builtin_fmaf(E2.w, __builtin_fabsf(d), m1) > 0.0f, 0)) {
        const float4 E1 = *pe1_j;
        const float Vn = __builtin_fmaf(E1.z, cz, __builtin_fmaf(E1.y, cy, E1.x * cx));   // = -V
        const float b = -(Vn * d);
        const float w = w1 - b;
        const float m = __builtin_fminf(__builtin_fminf(a, b), w);
        if (__builtin_fmaf(V0.w, __builtin_fabsf(d), m) > 0.0f) {
            const float T = -__builtin_fmaf(tz, NN.z, __builtin_fmaf(ty, NN.y, tx * NN.x));
            rare_candidate<ANYHIT>(r, list_slot, tri, a, b, w, d, T, V0.w, E1.w);
        }
    }
}

// Makes a value opaque to the optimiser so that per-ray quantities that happen to be equal
// in every lane (the shared origin of primary rays) are still carried and used per ray.
__device__ __forceinline__ float opaque(float x)
{
    asm volatile("" : "+v"(x));
    return x;
}

enum { MODE_PRIMARY = 0, MODE_RAYS_LINEAR = 1, MODE_RAYS_TILED = 2, MODE_RAYS_INDEXED = 3 };

template <int MODE, bool ANYHIT>
__global__ void __launch_bounds__(RT_WG) k_sweep(const SweepArgs A)
{
    // double-buffered v0/e1/e2 planes of one tile
    __shared__ float4 tile[2][RT_PLANES][RT_TILE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;

    Ray ray[RT_RPL];
    long slot[RT_RPL];        // index into cnt[] / list[] (band-local pixel or ray number)
    bool in_range[RT_RPL];

    // ---- ray set-up ------------------------------------------------------------------
#pragma unroll
    for (int r = 0; r < RT_RPL; ++r) {
        bool valid;
        double Sx = 0, Sy = 0, Sz = 0, Ex = 0, Ey = 0, Ez = 0;
        if (MODE == MODE_RAYS_LINEAR) {
            const long j = (long)blockIdx.x * RT_RAYS_PER_WG + r * RT_WG + tid;
            valid = j < A.n_rays;
            slot[r] = j;
        } else if (MODE == MODE_RAYS_INDEXED) {
            const long j = (long)blockIdx.x * RT_RAYS_PER_WG + r * RT_WG + tid;
            valid = j < A.n_rays;
            slot[r] = valid ? (long)A.active[j] : 0;
        } else {
            // workgroup = RT_TILE_W x RT_TILE_H pixels; wave = (8*RPL) x 8; lane = 8 x 8
            const int tiles_x = (A.W + RT_TILE_W - 1) / RT_TILE_W;
            const int tyi = blockIdx.x / tiles_x, txi = blockIdx.x - tyi * tiles_x;
            const int px = txi * RT_TILE_W + (wave & 1) * (8 * RT_RPL) + r * 8 + (lane & 7);
            const int py = A.y0 + tyi * RT_TILE_H + (wave >> 1) * 8 + (lane >> 3);
            valid = (px < A.W) && (py < A.y1);
            const long npix = (long)A.W * (A.y1 - A.y0);
            slot[r] = (long)blockIdx.y * npix + (long)(py - A.y0) * A.W + px;
            if (MODE == MODE_PRIMARY) {
                // driver.c:131-133 with S = origin = 0 (driver.c:121-123)
                Ex = (double)(px - A.W / 2);
                Ey = (double)(py - A.H / 2);
                Ez = A.ez;
            }
        }
        in_range[r] = valid;
        if (MODE != MODE_PRIMARY && valid) {
            const double* s = A.S + 3 * slot[r];
            const double* e = A.E + 3 * slot[r];
            Sx = s[0]; Sy = s[1]; Sz = s[2];
            Ex = e[0]; Ey = e[1]; Ez = e[2];
            if (!(Sx == Sx)) valid = false;       // NaN origin marks an unused ray slot
        }
        // D is normalised: t scales by |D| for every triangle alike, so the per-ray order of
        // candidates is unchanged and the per-triangle error bounds need no |D| factor.
        const double Dx = Ex - Sx, Dy = Ey - Sy, Dz = Ez - Sz;
        const double inv = 1.0 / __builtin_sqrt(Dx * Dx + Dy * Dy + Dz * Dz);
        if (!(inv == inv) || inv > 1e300) valid = false;      // zero-length ray hits nothing
        ray[r].ox = opaque(valid ? (float)Sx : 0.0f);
        ray[r].oy = opaque(valid ? (float)Sy : 0.0f);
        ray[r].oz = opaque(valid ? (float)Sz : 0.0f);
        ray[r].dx = valid ? (float)(Dx * inv) : 0.0f;
        ray[r].dy = valid ? (float)(Dy * inv) : 0.0f;
        ray[r].dz = valid ? (float)(Dz * inv) : 0.0f;
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
        // 3 planes x RT_TILE float4; one LDS-DMA instruction moves 64 float4 (1 KiB) per wave:
        // global source per lane, LDS destination = wave-uniform base + 16 * lane.
#pragma unroll
        for (int c = 0; c < (RT_TILE / 64 + 3) / 4; ++c) {
            const int chunk = c * 4 + wave;
            if (chunk < RT_TILE / 64) {
                const long g = (long)t * RT_TILE + chunk * 64 + lane;
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(A.v0 + g),
                    (__attribute__((address_space(3))) void*)&tile[buf][0][chunk * 64], 16, 0, 0);
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(A.e1 + g),
                    (__attribute__((address_space(3))) void*)&tile[buf][1][chunk * 64], 16, 0, 0);
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(A.e2 + g),
                    (__attribute__((address_space(3))) void*)&tile[buf][2][chunk * 64], 16, 0, 0);
#if RT_SWEEP_FORM == 1
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(A.nn + g),
                    (__attribute__((address_space(3))) void*)&tile[buf][3][chunk * 64], 16, 0, 0);
#endif
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
        const float4* __restrict__ pv0 = tile[buf][0];
        const float4* __restrict__ pe1 = tile[buf][1];
        const float4* __restrict__ pe2 = tile[buf][2];
        const float4* __restrict__ pnn = tile[buf][RT_PLANES - 1];
        const int tri0 = t * RT_TILE;
        // The rare path splits the loop body into basic blocks, which stops the compiler from
        // hoisting the next triangle's LDS reads above the current triangle's arithmetic; the
        // reads are therefore issued one triangle ahead by hand (RT_PREFETCH).
#if RT_STAGED
        // hot loop reads three planes; the e1 plane is read only inside the u slab
        float4 nV0 = pv0[0], nE2 = pe2[0], nNN = pnn[0];
#pragma unroll 2
        for (int j = 0; j < RT_TILE; ++j) {
            const float4 V0 = nV0, E2 = nE2, NN = nNN;
            const int jn = (j + 1) & (RT_TILE - 1);     // last iteration re-reads entry 0 (unused)
            nV0 = pv0[jn]; nE2 = pe2[jn]; nNN = pnn[jn];
#pragma unroll
            for (int r = 0; r < RT_RPL; ++r)
                staged_pair<ANYHIT>(ray[r], A.list + slot[r] * RT_LIST_K, tri0 + j, V0, E2, NN, pe1 + j);
        }
#else
#if RT_PREFETCH
        float4 nV0 = pv0[0], nE1 = pe1[0], nE2 = pe2[0], nNN = pnn[0];
#endif
#pragma unroll 2
        for (int j = 0; j < RT_TILE; ++j) {
#if RT_PREFETCH
            const float4 V0 = nV0, E1 = nE1, E2 = nE2, NN = nNN;
            const int jn = (j + 1) & (RT_TILE - 1);     // last iteration re-reads entry 0 (unused)
            nV0 = pv0[jn]; nE1 = pe1[jn]; nE2 = pe2[jn]; nNN = pnn[jn];
#else
            const float4 V0 = pv0[j];
            const float4 E1 = pe1[j];
            const float4 E2 = pe2[j];
            const float4 NN = pnn[j];       // form 0: aliases the e2 plane and is unused
#endif
#pragma unroll
            for (int r = 0; r < RT_RPL; ++r) {
                float a, b, w, d, T;
                if (__builtin_expect(pair_test(ray[r], V0, E1, E2, NN, a, b, w, d, T), 0))
                    rare_candidate<ANYHIT>(ray[r], A.list + slot[r] * RT_LIST_K, tri0 + j,
                                           a, b, w, d, T, V0.w, E1.w);
            }
        }
#endif
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
    if (!a.v0 || !a.e1 || !a.e2 || !a.nn || !a.cnt || !a.list) {
        rt_set_error("sweep: null buffer");
        return RT_E_ARG;
    }
    if (a.num_tris_padded <= 0 || a.num_tris_padded % RT_TILE != 0 ||
        a.num_tris > a.num_tris_padded) {
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
