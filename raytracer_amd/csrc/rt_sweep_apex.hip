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
// A pair lies in the u slab (0 <= u <= 1) exactly when U and W' have the sign of det, and
// since U + W' = det that is when U * W' >= 0.  The hot loop therefore evaluates, per pair,
//
//     s = fma(U, W', c)        and enters the rare path when s >= 0,
//
// where c is the per-triangle rounding allowance that makes the fp32 test a conservative
// filter (derivation at k_pack_apex).  The rare path finishes the inside test (V, det, the
// v and u+v edges, the ray parameter) and appends survivors to the ray's candidate list; the
// fp64 pass of rt_exact.hip then decides with the reference's exact arithmetic.
//
//   k_sweep_apex    any-hit, unit-length directions: two 3-term dot products + the product
//                   test = 8 VALU instructions per pair.
//   k_sweep_raster  closest hit for the pixel grid.  D = (X, Y, ez) with integer X, Y and a
//                   constant ez that k_pack_apex folds into the z components, and each lane
//                   owns RT_RASTER_RPL pixels of one row, so the row part Y*p.y + ez*p.z is
//                   formed once per lane and triangle and each pixel adds one FMA:
//                   (2 + 3*RPL) FMA + the sign tests per RPL pairs, about 4 VALU instructions
//                   per pair.  Every (pixel, triangle) pair is still tested individually.
//
// Tiles of RT_TILE triangles (4 planes) are double-buffered in LDS by LDS-DMA loads and read
// at wave-uniform addresses (broadcast, no bank conflicts).
#include "rt_sweep_common.h"

#ifndef RT_PREFETCH2
#define RT_PREFETCH2 0
#endif
#ifndef RT_APEX_UNROLL
#define RT_APEX_UNROLL 4
#endif
#ifndef RT_RASTER_UNROLL
#define RT_RASTER_UNROLL 4
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
__device__ __forceinline__ void finish_raster(Book& bk, int32_t* __restrict__ list_slot, int tri,
                                              const Tri2& q, const float4* __restrict__ p,
                                              float X, float Y, float yu)
{
    const float4 Hh = p[2 * RT_TILE], N = p[3 * RT_TILE];
    const float margin = q.Q.w;
    if (margin == INF) {            // nearly degenerate triangle: no fp32 verdict
        if (bk.best_hi != -INF) push(bk, list_slot, tri);
        return;
    }
    const float U = __builtin_fmaf(X, q.P.x, yu);
    const float V = __builtin_fmaf(X, Hh.x, __builtin_fmaf(Y, Hh.y, Hh.z));
    const float d = __builtin_fmaf(X, N.x, __builtin_fmaf(Y, N.y, N.z));
    const float a = U * d, b = V * d;
    const float w = __builtin_fmaf(d, d, -a) - b;
    const float m3 = __builtin_fminf(__builtin_fminf(a, b), w);
    if (__builtin_fmaf(margin, __builtin_fabsf(d), m3) > 0.0f)
        candidate<false>(bk, list_slot, tri, a, b, w, d, Hh.w, 0.0f, margin, N.w,
                         U22 * __builtin_fabsf(Hh.w));
}

template <int R>
__global__ void __launch_bounds__(RT_WG, RT_MIN_WAVES) k_sweep_raster(const SweepArgs A)
{
    __shared__ float4 tile[2][RT_APEX_PLANES][RT_TILE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;

    // workgroup = (16 R) x 16 pixels; wave = (8 R) x 8; lane = one row, R pixels 8 apart
    const int tiles_x = (A.W + 16 * R - 1) / (16 * R);
    const int tyi = blockIdx.x / tiles_x, txi = blockIdx.x - tyi * tiles_x;
    const int px0 = txi * (16 * R) + (wave & 1) * (8 * R) + (lane & 7);
    const int lr = tyi * 16 + (wave >> 1) * 8 + (lane >> 3);       // local row of the call's row set
    const float Y = (float)(A.rows.image_row(lr) - A.H / 2);        // driver.c:132
    // ray number: local pixel, in the segment of this workgroup's triangle chunk
    const long slot0 = (long)blockIdx.y * A.chunk_stride + (long)lr * A.W + px0;

    float X[R];
    Book bk[R];
    bool valid[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int px = px0 + 8 * r;
        valid[r] = (px < A.W) && (lr < A.rows.rows);
        X[r] = (float)(px - A.W / 2);                   // driver.c:131
        bk[r].best = INF;
        bk[r].best_hi = valid[r] ? INF : -INF;
        bk[r].cnt = valid[r] ? 0 : RT_CNT_RETIRED;
    }

    // this workgroup's share of the triangle range: chunk blockIdx.y of gridDim.y
    const int t_first = A.tri_begin / RT_TILE, t_count = A.tri_end / RT_TILE - t_first;
    const int t_begin = t_first + (int)((long)t_count * blockIdx.y / gridDim.y);
    const int t_end = t_first + (int)((long)t_count * (blockIdx.y + 1) / gridDim.y);
    unsigned long long tiles_done = 0;
    if (t_begin < t_end) {
        stage_tile<RT_APEX_PLANES>(A.planes, A.plane_stride, t_begin, tile[0], wave, lane);
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
    }
    for (int t = t_begin; t < t_end; ++t) {
        const int buf = (t - t_begin) & 1;
        if (t + 1 < t_end)
            stage_tile<RT_APEX_PLANES>(A.planes, A.plane_stride, t + 1, tile[buf ^ 1], wave, lane);
        const float4* __restrict__ base = tile[buf][0];
        const int tri0 = t * RT_TILE;
        // LDS reads are issued one triangle ahead by hand (the rare path splits the body into
        // basic blocks, which keeps the compiler from hoisting them)
#if RT_PREFETCH2
        Tri2 next = load_tri2(base), next2 = load_tri2(base + 1);
#else
        Tri2 next = load_tri2(base);
#endif
#pragma unroll kRasterUnroll
        for (int j = 0; j < RT_TILE; ++j) {
            const Tri2 cur = next;
#if RT_PREFETCH2
            next = next2;
            next2 = load_tri2(base + ((j + 2) & (RT_TILE - 1)));    // the last two iterations re-read entries 0, 1 (unused)
#else
            next = load_tri2(base + ((j + 1) & (RT_TILE - 1)));     // last iteration re-reads entry 0 (unused)
#endif
            const float yu = __builtin_fmaf(Y, cur.P.y, cur.P.z);
            const float yw = __builtin_fmaf(Y, cur.Q.y, cur.Q.z);
            float s[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float U = __builtin_fmaf(X[r], cur.P.x, yu);
                const float Wp = __builtin_fmaf(X[r], cur.Q.x, yw);
                s[r] = __builtin_fmaf(U, Wp, cur.P.w);
            }
            float mx = s[0];
#pragma unroll
            for (int r = 1; r < R; ++r) mx = __builtin_fmaxf(mx, s[r]);     // v_max3_f32 pairs
            if (__builtin_expect(mx >= 0.0f, 0)) {
#pragma unroll
                for (int r = 0; r < R; ++r)
                    if (s[r] >= 0.0f)
                        finish_raster(bk[r], A.list + (slot0 + 8 * r) * RT_LIST_K, tri0 + j, cur,
                                      base + j, X[r], Y, yu);
            }
        }
        ++tiles_done;
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
    }

#pragma unroll
    for (int r = 0; r < R; ++r)
        if (valid[r]) A.cnt[slot0 + 8 * r] = bk[r].cnt;
    if (A.tests && tid == 0)
        atomicAdd(A.tests, tiles_done * (unsigned long long)RT_TILE * (RT_WG * R));
}

// ------------------------------------------------------------------------------------------
// any-hit sweep for rays that end at the apex (shadow rays of one light)
// ------------------------------------------------------------------------------------------
struct ApexRay {
    float dx, dy, dz;   // unit direction S -> apex; NaN once the ray has retired
    float len;          // |apex - S|: ray parameter of the apex
};

// candidate() for an any-hit sweep whose triangle range is shared by several workgroups per
// ray: the candidate count and the occluded flag of a ray live in global memory only and are
// updated atomically.  The order of a ray's list entries is then arbitrary, which an any-hit
// query does not care about (triangle_intersection.cu:169 asks for any dist > 0).
__device__ __forceinline__ void candidate_shared(Book& r, int32_t* __restrict__ cnt_slot,
                                                 int32_t* __restrict__ list_slot, int tri, float a,
                                                 float b, float w, float d, float T, float t0,
                                                 float margin, float en, float eT)
{
    if (r.best_hi == -INF) return;
    const float ad = __builtin_fabsf(d);
    const float den = ad - en;
    bool certain = false;
    if (den > 0.0f) {
        const float q = T / d;
        const float t = t0 + q;
        const float terr = (eT + __builtin_fabsf(q) * en) / den * 1.000001f +
                           U22 * (__builtin_fabsf(t0) + __builtin_fabsf(q));
        if (!(t >= -terr)) return;
        certain = (__builtin_fminf(__builtin_fminf(a, b), w) >= margin * ad) && (t > terr);
    }
    if (certain) {
        atomicOr(cnt_slot, RT_CNT_OCCLUDED);    // the list no longer matters
        r.best_hi = -INF;
        return;
    }
    const int at = atomicAdd(cnt_slot, 1) & RT_CNT_MASK;
    if (at < RT_LIST_K) list_slot[at] = tri;
}

__device__ __forceinline__ void finish_apex(Book& bk, const ApexRay& r, int32_t* __restrict__ cnt_slot,
                                            int32_t* __restrict__ list_slot,
                                            int tri, const Tri2& q, const float4* __restrict__ p)
{
    const float4 Hh = p[2 * RT_TILE], N = p[3 * RT_TILE];
    const float margin = q.Q.w;
    if (margin == INF) {
        if (bk.best_hi != -INF) {
            const int at = atomicAdd(cnt_slot, 1) & RT_CNT_MASK;
            if (at < RT_LIST_K) list_slot[at] = tri;
        }
        return;
    }
    const float U = dot3(r.dx, r.dy, r.dz, q.P.x, q.P.y, q.P.z);
    const float V = dot3(r.dx, r.dy, r.dz, Hh.x, Hh.y, Hh.z);
    const float d = dot3(r.dx, r.dy, r.dz, N.x, N.y, N.z);
    const float a = U * d, b = V * d;
    const float w = __builtin_fmaf(d, d, -a) - b;
    const float m3 = __builtin_fminf(__builtin_fminf(a, b), w);
    if (__builtin_fmaf(margin, __builtin_fabsf(d), m3) > 0.0f)
        // parameter from S along the unit direction: len + k/det (the reference's t times len;
        // only its sign matters for an any-hit query, triangle_intersection.cu:169)
        candidate_shared(bk, cnt_slot, list_slot, tri, a, b, w, d, Hh.w, r.len, margin, N.w,
                         U22 * __builtin_fabsf(Hh.w));
}

enum { APEX_TILED = 0, APEX_INDEXED = 1 };

template <int MODE, int R>
__global__ void __launch_bounds__(RT_WG, RT_MIN_WAVES) k_sweep_apex(SweepArgs A)
{
    __shared__ float4 tile[2][RT_APEX_PLANES][RT_TILE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;

    // ray set (light) of this workgroup
    const long set = blockIdx.y;
    A.planes += set * A.light_plane_stride;
    A.S += 3 * set * A.light_ray_stride;
    A.E += 3 * set * A.light_ray_stride;
    A.cnt += set * A.light_ray_stride;
    A.list += set * A.light_ray_stride * RT_LIST_K;
    if (MODE == APEX_INDEXED) {
        A.active += set * A.light_ray_stride;
        if (A.live_counts) A.n_rays = A.live_counts[set];
        if ((long)blockIdx.x * (RT_WG * R) >= A.n_rays) return;     // grid sized for the largest set
    }

    ApexRay ray[R];
    Book bk[R];
    int slot[R];
    bool in_range[R];
    const float qnan = __builtin_nanf("");

#pragma unroll
    for (int r = 0; r < R; ++r) {
        bool valid;
        if (MODE == APEX_INDEXED) {
            const long j = (long)blockIdx.x * (RT_WG * R) + r * RT_WG + tid;
            valid = j < A.n_rays;
            slot[r] = valid ? A.active[j] : 0;
        } else {
            const int tiles_x = (A.W + 16 * R - 1) / (16 * R);
            const int tyi = blockIdx.x / tiles_x, txi = blockIdx.x - tyi * tiles_x;
            const int px = txi * (16 * R) + (wave & 1) * (8 * R) + r * 8 + (lane & 7);
            const int lr = tyi * 16 + (wave >> 1) * 8 + (lane >> 3);
            valid = (px < A.W) && (lr < A.rows.rows);
            slot[r] = valid ? (int)((long)lr * A.W + px) : 0;
        }
        in_range[r] = valid;
        double Dx = 0, Dy = 0, Dz = 0, len = 0;
        if (valid) {
            const double* s = A.S + 3 * (long)slot[r];
            const double* e = A.E + 3 * (long)slot[r];
            const double Sx = s[0];
            Dx = e[0] - Sx; Dy = e[1] - s[1]; Dz = e[2] - s[2];
            len = __builtin_sqrt(Dx * Dx + Dy * Dy + Dz * Dz);
            const double inv = 1.0 / len;
            // NaN origin marks an unused ray slot; a zero-length ray hits nothing
            if (!(Sx == Sx) || !(inv == inv) || inv > 1e300) valid = false;
            Dx *= inv; Dy *= inv; Dz *= inv;
        }
        // the ray's count and flags live in cnt[] (zeroed by the host before the first range)
        if (in_range[r]) {
            const int c0 = A.cnt[slot[r]];
            if (!valid && !(c0 & RT_CNT_RETIRED)) atomicOr(&A.cnt[slot[r]], RT_CNT_RETIRED);
            if (c0 & (RT_CNT_OCCLUDED | RT_CNT_RETIRED)) valid = false;
        }
        bk[r].best = INF;
        bk[r].best_hi = valid ? INF : -INF;
        bk[r].cnt = 0;
        // a retired ray carries a NaN direction: s is NaN and never enters the rare path
        ray[r].dx = valid ? (float)Dx : qnan;
        ray[r].dy = valid ? (float)Dy : qnan;
        ray[r].dz = valid ? (float)Dz : qnan;
        ray[r].len = (float)len;
    }

    // this workgroup's share of the triangle range: chunk blockIdx.z of gridDim.z
    const int t_first = A.tri_begin / RT_TILE, t_count = A.tri_end / RT_TILE - t_first;
    const int t_begin = t_first + (int)((long)t_count * blockIdx.z / gridDim.z);
    const int t_end = t_first + (int)((long)t_count * (blockIdx.z + 1) / gridDim.z);
    unsigned long long tiles_done = 0;
    if (t_begin < t_end) {
        stage_tile<RT_APEX_PLANES>(A.planes, A.plane_stride, t_begin, tile[0], wave, lane);
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
    }
    for (int t = t_begin; t < t_end; ++t) {
        const int buf = (t - t_begin) & 1;
        if (t + 1 < t_end)
            stage_tile<RT_APEX_PLANES>(A.planes, A.plane_stride, t + 1, tile[buf ^ 1], wave, lane);
        const float4* __restrict__ base = tile[buf][0];
        const int tri0 = t * RT_TILE;
        Tri2 next = load_tri2(base);
#pragma unroll kApexUnroll
        for (int j = 0; j < RT_TILE; ++j) {
            const Tri2 cur = next;
            next = load_tri2(base + ((j + 1) & (RT_TILE - 1)));
            float s[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float U = dot3(ray[r].dx, ray[r].dy, ray[r].dz, cur.P.x, cur.P.y, cur.P.z);
                const float Wp = dot3(ray[r].dx, ray[r].dy, ray[r].dz, cur.Q.x, cur.Q.y, cur.Q.z);
                s[r] = __builtin_fmaf(U, Wp, cur.P.w);
            }
            float mx = s[0];
#pragma unroll
            for (int r = 1; r < R; ++r) mx = __builtin_fmaxf(mx, s[r]);
            if (__builtin_expect(mx >= 0.0f, 0)) {
#pragma unroll
                for (int r = 0; r < R; ++r)
                    if (s[r] >= 0.0f) {
                        finish_apex(bk[r], ray[r], A.cnt + slot[r], A.list + (long)slot[r] * RT_LIST_K,
                                    tri0 + j, cur, base + j);
                        if (bk[r].best_hi == -INF) ray[r].dx = qnan;
                    }
            }
        }
        ++tiles_done;
        __builtin_amdgcn_s_waitcnt(0);
        // retire the workgroup once every ray in it is certainly occluded or unused; with
        // several chunks per ray the flag another chunk's workgroup may have set is polled
        bool done = true;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (gridDim.z > 1 && bk[r].best_hi != -INF &&
                (__hip_atomic_load(&A.cnt[slot[r]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) &
                 RT_CNT_OCCLUDED)) {
                bk[r].best_hi = -INF;
                ray[r].dx = qnan;
            }
            done = done && (bk[r].best_hi == -INF);
        }
        if (__syncthreads_and(done)) break;
    }

    if (A.tests && tid == 0)
        atomicAdd(A.tests, tiles_done * (unsigned long long)RT_TILE * (RT_WG * R));
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

static int device_cus()
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
// triangle range into chunks that run as separate workgroups (with candidate lists of their
// own) makes the rounds shorter and fills small bands.  Measured on the 1 M-triangle soup:
// 8 chunks instead of 1 take 6 % off a 3840x2160 sweep and 33 % off a 3840x270 one, at the
// price of three times as many candidates for the exact pass (less pruning across chunks).
// A chunk keeps at least 8 tiles; a mesh too small to fill the machine that way gives up
// pixels per lane instead.
void rt_raster_shape(int W, int rows, int tiles, int* pixels_per_lane, int* chunks)
{
    const long slots = 5L * device_cus();           // resident workgroups
    const long tiles_y = (rows + RT_TILE_H - 1) / RT_TILE_H;
    int max_chunks = tiles / 8;
    if (max_chunks < 1) max_chunks = 1;
    if (max_chunks > RT_RASTER_MAX_CHUNKS) max_chunks = RT_RASTER_MAX_CHUNKS;
    int r = RT_RASTER_RPL;
    while (r > 1 && 16 * r > W + 8 * r) r /= 2;         // no lane set mostly off the image
    for (;;) {
        const long wgs = (long)((W + 16 * r - 1) / (16 * r)) * tiles_y;
        if (wgs * max_chunks >= slots || r == 1) {
            long c = (32 * slots + wgs - 1) / wgs;      // about 32 rounds are fine-grained enough
            if (c > max_chunks) c = max_chunks;
            if (c < 1) c = 1;
            *pixels_per_lane = r;
            *chunks = (int)c;
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
    int r = 1, chunks = 1;
    rt_raster_shape(a.W, a.rows.rows, (a.tri_end - a.tri_begin) / RT_TILE, &r, &chunks);
    if (a.chunks != chunks || (chunks > 1 && a.chunk_stride < (long)a.W * a.rows.rows)) {
        rt_set_error("raster sweep: caller and launcher disagree on the chunk count (%d vs %d)",
                     a.chunks, chunks);
        return RT_E_ARG;
    }
    const int tiles_x = (a.W + 16 * r - 1) / (16 * r);
    const int tiles_y = (a.rows.rows + RT_TILE_H - 1) / RT_TILE_H;
    const dim3 grid(tiles_x * tiles_y, chunks);
    switch (r) {
    case 1: return launch(k_sweep_raster<1>, a, grid, st, "raster sweep");
    case 2: return launch(k_sweep_raster<2>, a, grid, st, "raster sweep");
    case 4: return launch(k_sweep_raster<4>, a, grid, st, "raster sweep");
    case 8: return launch(k_sweep_raster<8>, a, grid, st, "raster sweep");
#if RT_RASTER_RPL > 8
    case RT_RASTER_RPL: return launch(k_sweep_raster<RT_RASTER_RPL>, a, grid, st, "raster sweep");
#endif
    }
    rt_set_error("raster sweep: unsupported pixels per lane %d", r);
    return RT_E_ARG;
}

// Rays per lane of an any-hit launch over n rays: RT_APEX_RPL while that still gives every CU
// two to three workgroups, fewer for small batches.  The work per pair is the same, but the two
// LDS reads per triangle are shared by fewer pairs: below 4 rays per lane the loop is bound by
// LDS bandwidth, so this is only worth it when the machine would otherwise sit idle.
static int apex_rays_per_lane(long n)
{
    int r = RT_APEX_RPL;
    while (r > 1 && 2 * n / (RT_WG * r) < 5L * device_cus()) r /= 2;      // 2.5 workgroups per CU
    return r;
}

// Triangle chunks of an any-hit launch of `wgs` workgroups in total over `tiles` tiles: enough
// to give the machine about 8 rounds of work, each chunk at least 8 tiles long.
static int apex_chunks(long wgs, int tiles)
{
    const long slots = 5L * device_cus();
    long c = (8 * slots + wgs - 1) / wgs;
    if (c > tiles / 8) c = tiles / 8;
    if (c > RT_RASTER_MAX_CHUNKS) c = RT_RASTER_MAX_CHUNKS;
    if (const char* e = getenv("RT_AMD_APEX_CHUNKS")) c = atoi(e);
    return c < 1 ? 1 : (int)c;
}

template <int MODE>
static int launch_apex(const SweepArgs& a, int r, unsigned wgs, hipStream_t st)
{
    const int sets = a.lights > 0 ? a.lights : 1;
    const dim3 grid(wgs, sets, apex_chunks((long)wgs * sets, (a.tri_end - a.tri_begin) / RT_TILE));
    switch (r) {
    case 1: return launch(k_sweep_apex<MODE, 1>, a, grid, st, "apex sweep");
    case 2: return launch(k_sweep_apex<MODE, 2>, a, grid, st, "apex sweep");
    case 4: return launch(k_sweep_apex<MODE, 4>, a, grid, st, "apex sweep");
#if RT_APEX_RPL > 4
    case RT_APEX_RPL: return launch(k_sweep_apex<MODE, RT_APEX_RPL>, a, grid, st, "apex sweep");
#endif
    }
    rt_set_error("apex sweep: unsupported rays per lane %d", r);
    return RT_E_ARG;
}

int rt_launch_sweep_apex(const SweepArgs& a, hipStream_t st)
{
    int rc = check_common(a);
    if (rc) return rc;
    if (!a.S || !a.E) {
        rt_set_error("apex sweep: null ray arrays");
        return RT_E_ARG;
    }
    if (a.active) {     // indexed: n_rays entries of `active` name the rays
        if (a.n_rays <= 0) return RT_OK;
        const int r = apex_rays_per_lane((long)a.n_rays * (a.lights > 0 ? a.lights : 1));
        const int per_wg = RT_WG * r;
        return launch_apex<APEX_INDEXED>(a, r, (unsigned)((a.n_rays + per_wg - 1) / per_wg), st);
    }
    if (a.W <= 0 || a.rows.rows <= 0) {
        rt_set_error("apex sweep: empty band");
        return RT_E_ARG;
    }
    const int r = apex_rays_per_lane((long)a.W * a.rows.rows * (a.lights > 0 ? a.lights : 1));
    const int tiles_x = (a.W + 16 * r - 1) / (16 * r);
    const int tiles_y = (a.rows.rows + RT_TILE_H - 1) / RT_TILE_H;
    return launch_apex<APEX_TILED>(a, r, (unsigned)(tiles_x * tiles_y), st);
}
