// rt_sort.hip -- orders the shadow rays of a set by direction as seen from their light.
//
// The any-hit sweep (rt_sweep_apex.hip) pays for a triangle's second-level test and rare path
// whenever ANY of a wave's 64 R rays lies in the u extent of the triangle's bounding square.
// Rays of neighbouring pixels are neighbours on a smooth surface, but where neighbouring pixels
// see different depths (a triangle soup, silhouettes) their shadow rays fan out along the
// light's epipolar lines.  Ordering the rays of a set by the two minor components of
// D' = D / D_k (u-major, or along a Hilbert curve: see rt_launch_sort_rays) makes every wave's
// rays a narrow bundle through the light, which fewer triangles' extents meet, and whose rays
// tend to share occluders and retire together.  An any-hit query does not depend on the order
// in which rays are swept, so the results are unchanged.
#include "rt_internal.h"

#include <rocprim/device/device_radix_sort.hpp>

namespace {

struct KeyArgs {
    int key_off[RT_APEX_SETS];      // where the keys of set s start in the packed key array
};

// Position of the cell (x, y) of a 4096 x 4096 grid along the Hilbert curve.  Unlike the Z
// order, consecutive positions are always neighbouring cells, so any run of consecutive rays
// covers a compact patch of directions (a Z-order run that straddles a quadrant boundary spans
// half the grid).
__device__ __forceinline__ unsigned hilbert_index(unsigned x, unsigned y)
{
    unsigned d = 0;
#pragma unroll
    for (unsigned s = 2048u; s > 0u; s >>= 1) {
        const unsigned rx = (x & s) ? 1u : 0u, ry = (y & s) ? 1u : 0u;
        d += s * s * ((3u * rx) ^ ry);
        if (ry == 0u) {
            if (rx == 1u) {
                x = 4095u - x;
                y = 4095u - y;
            }
            const unsigned t = x;
            x = y;
            y = t;
        }
    }
    return d;
}

// blockIdx.y = set = light * 3 + dominant axis; ids of set s at ids[s * stride ...], n = counts[s].
// The sets are sorted in one pass over all of them: a ray's key carries its set above the 25 bits
// of its direction, and keys and ray numbers are written back to back (set s from key_off[s] on).
constexpr unsigned kRayKeyBits = 25;

__global__ void __launch_bounds__(RT_WG)
k_ray_keys(const double* __restrict__ S, const double* __restrict__ E, const int32_t* __restrict__ ids,
           const int* __restrict__ counts, long stride, KeyArgs ka, int u_major, unsigned* __restrict__ keys,
           int32_t* __restrict__ vals)
{
    const int set = blockIdx.y, light = set / 3, axis = set - 3 * light;
    const int n = counts[set];
    S += 3 * light * stride;
    E += 3 * light * stride;
    ids += set * stride;
    keys += ka.key_off[set];
    vals += ka.key_off[set];
    for (int i = blockIdx.x * RT_WG + threadIdx.x; i < n; i += gridDim.x * RT_WG) {
        const long id = ids[i];
        const double dx = E[3 * id] - S[3 * id], dy = E[3 * id + 1] - S[3 * id + 1],
                     dz = E[3 * id + 2] - S[3 * id + 2];
        const double dk = axis == 0 ? dx : (axis == 1 ? dy : dz);
        const double du = axis == 0 ? dy : dx, dv = axis == 2 ? dy : dz;
        // minor components in [-1, 1] -> 12 bits each; opposite directions are different bundles
        const float u = (float)(du / dk), v = (float)(dv / dk);
        const unsigned qu = (unsigned)fminf(fmaxf((u + 1.0f) * 2047.5f, 0.0f), 4095.0f);
        const unsigned qv = (unsigned)fminf(fmaxf((v + 1.0f) * 2047.5f, 0.0f), 4095.0f);
        keys[i] = (u_major ? (qu << 12) | qv : hilbert_index(qu, qv)) | (dk < 0.0 ? 1u << 24 : 0u) |
                  ((unsigned)set << kRayKeyBits);
        vals[i] = (int32_t)id;
    }
}

// The sorted ray numbers back into the sets' own lists: set s to out[s * stride ...].
__global__ void __launch_bounds__(RT_WG)
k_unpack_sets(const int32_t* __restrict__ vals, const int* __restrict__ counts, long stride, KeyArgs ka,
              int32_t* __restrict__ out)
{
    const int set = blockIdx.y, n = counts[set];
    vals += ka.key_off[set];
    out += set * stride;
    for (int i = blockIdx.x * RT_WG + threadIdx.x; i < n; i += gridDim.x * RT_WG) out[i] = vals[i];
}

// Solid angle a triangle subtends at the apex (Van Oosterom and Strackee), as a sort key: the
// larger, the more of the rays through the apex the triangle can stop.
__global__ void __launch_bounds__(RT_WG)
k_tri_keys(const double* __restrict__ x, const double* __restrict__ y, const double* __restrict__ z,
           const int* __restrict__ iA, const int* __restrict__ iB, const int* __restrict__ iC,
           int num_tris, int num_tris_padded, double px, double py, double pz,
           unsigned* __restrict__ keys, int32_t* __restrict__ ids)
{
    const int i = blockIdx.x * RT_WG + threadIdx.x;
    if (i >= num_tris_padded) return;
    float omega = -1.0f;                        // padding sorts last
    if (i < num_tris) {
        const int a = iA[i], b = iB[i], c = iC[i];
        const double ax = x[a] - px, ay = y[a] - py, az = z[a] - pz;
        const double bx = x[b] - px, by = y[b] - py, bz = z[b] - pz;
        const double cx = x[c] - px, cy = y[c] - py, cz = z[c] - pz;
        const double ra = sqrt(ax * ax + ay * ay + az * az), rb = sqrt(bx * bx + by * by + bz * bz),
                     rc = sqrt(cx * cx + cy * cy + cz * cz);
        const double num = fabs(ax * (by * cz - bz * cy) + ay * (bz * cx - bx * cz) + az * (bx * cy - by * cx));
        const double den = ra * rb * rc + (ax * bx + ay * by + az * bz) * rc +
                           (ax * cx + ay * cy + az * cz) * rb + (bx * cx + by * cy + bz * cz) * ra;
        const double w = 2.0 * atan2(num, den);
        omega = (w == w) ? (float)w : 0.0f;
    }
    // descending order of omega through an ascending sort: non-negative floats order like their bits.
    // (The full-width key matters: the triangles of an evenly tessellated mesh subtend nearly the
    // same angle, and the low bits of the key are what scatters neighbours in the mesh over the
    // tiles -- with a 16-bit key, ties keep the file's order, the triangles of a tile lie side by
    // side, and a 100 000-triangle torus at 3840x2160 takes 5.5 ms a frame instead of 4.2.)
    keys[i] = omega < 0.0f ? 0xFFFFFFFFu : 0x7FFFFFFFu - __float_as_uint(omega);
    ids[i] = i < num_tris ? i : -1;
}

// Keys of a fixed pseudo-random order of the triangles (a hash of the triangle number; the
// padding sorts last).
__global__ void __launch_bounds__(RT_WG)
k_scatter_keys(int num_tris, int num_tris_padded, unsigned* __restrict__ keys, int32_t* __restrict__ ids)
{
    const int i = blockIdx.x * RT_WG + threadIdx.x;
    if (i >= num_tris_padded) return;
    unsigned h = (unsigned)i * 2654435761u;
    h ^= h >> 15;
    h *= 2246822519u;
    h ^= h >> 13;
    keys[i] = i < num_tris ? h >> 1 : 0xFFFFFFFFu;
    ids[i] = i < num_tris ? i : -1;
}

}  // namespace

// Order in which the any-hit sweep of the rays through `apex` visits the triangles: by the solid
// angle they subtend there, largest first, so that occluded rays meet an occluder early.  An
// any-hit query does not depend on the order in which triangles are tried.  order[k] = triangle
// at position k, -1 for the padding at the end.
int rt_launch_sort_triangles(rt_scene* sc, const double apex[3], int32_t* order, hipStream_t st)
{
    const unsigned np = (unsigned)sc->num_tris_padded;
    if (np == 0) return RT_OK;
    size_t temp_bytes = 0;
    if (rocprim::radix_sort_pairs(nullptr, temp_bytes, (unsigned*)nullptr, (unsigned*)nullptr,
                                  (int32_t*)nullptr, (int32_t*)nullptr, np, 0u, 32u, st) != hipSuccess) {
        rt_set_error("triangle sort: size query failed");
        return RT_E_HIP;
    }
    const size_t arr = ((size_t)np * 4 + 255) & ~(size_t)255;
    if (int rc = sc->ws_sort_tri.reserve(3 * arr + temp_bytes)) return rc;
    unsigned* keys_in = sc->ws_sort_tri.as<unsigned>();
    unsigned* keys_out = reinterpret_cast<unsigned*>(sc->ws_sort_tri.as<char>() + arr);
    int32_t* ids_in = reinterpret_cast<int32_t*>(sc->ws_sort_tri.as<char>() + 2 * arr);
    void* temp = sc->ws_sort_tri.as<char>() + 3 * arr;
    hipLaunchKernelGGL(k_tri_keys, dim3((np + RT_WG - 1) / RT_WG), dim3(RT_WG), 0, st, sc->ex.as<double>(),
                       sc->ey.as<double>(), sc->ez.as<double>(), sc->iA.as<int>(), sc->iB.as<int>(),
                       sc->iC.as<int>(), sc->num_tris, (int)np, apex[0], apex[1], apex[2], keys_in, ids_in);
    RT_HIP(hipGetLastError());
    RT_HIP(rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, ids_in, order, np, 0u, 32u, st));
    return RT_OK;
}

// A fixed order that scatters the triangles: neighbours in the mesh file (neighbours in space,
// for a tessellated surface) land in different tiles.  The cell-lattice sweep deals the tiles to
// its workgroups, and a tile whose triangles lie side by side sends its workgroup to the same few
// rays again and again while others idle; scattered, every tile is a sample of the whole mesh.
// Computed once per mesh (it does not depend on the apex).  order[k] = triangle at position k,
// -1 for the padding at the end.
int rt_launch_scatter_triangles(rt_scene* sc, int32_t* order, hipStream_t st)
{
    const unsigned np = (unsigned)sc->num_tris_padded;
    if (np == 0) return RT_OK;
    size_t temp_bytes = 0;
    if (rocprim::radix_sort_pairs(nullptr, temp_bytes, (unsigned*)nullptr, (unsigned*)nullptr,
                                  (int32_t*)nullptr, (int32_t*)nullptr, np, 0u, 32u, st) != hipSuccess) {
        rt_set_error("triangle scatter: size query failed");
        return RT_E_HIP;
    }
    const size_t arr = ((size_t)np * 4 + 255) & ~(size_t)255;
    if (int rc = sc->ws_sort_tri.reserve(3 * arr + temp_bytes)) return rc;
    unsigned* keys_in = sc->ws_sort_tri.as<unsigned>();
    unsigned* keys_out = reinterpret_cast<unsigned*>(sc->ws_sort_tri.as<char>() + arr);
    int32_t* ids_in = reinterpret_cast<int32_t*>(sc->ws_sort_tri.as<char>() + 2 * arr);
    void* temp = sc->ws_sort_tri.as<char>() + 3 * arr;
    hipLaunchKernelGGL(k_scatter_keys, dim3((np + RT_WG - 1) / RT_WG), dim3(RT_WG), 0, st, sc->num_tris, (int)np,
                       keys_in, ids_in);
    RT_HIP(hipGetLastError());
    RT_HIP(rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, ids_in, order, np, 0u, 32u, st));
    return RT_OK;
}

// Sorts the ray numbers of every set (ids_in[s * stride ...], counts[s] of them; counts_dev the
// same numbers on the device) by direction into ids_out; empty sets are skipped.
int rt_launch_sort_rays(rt_scene* sc, const double* S, const double* E, const int32_t* ids_in,
                        int32_t* ids_out, const int* counts, const int* counts_dev, int sets,
                        long stride, hipStream_t st)
{
    KeyArgs ka{};
    long total = 0;
    unsigned largest = 0;
    for (int s = 0; s < sets; ++s) {
        ka.key_off[s] = (int)total;
        total += counts[s];
        if ((unsigned)counts[s] > largest) largest = (unsigned)counts[s];
    }
    if (total == 0) return RT_OK;
    // one sort over all sets (the set number above the direction bits of the key): a radix sort
    // of a few hundred thousand keys is a chain of small launches, and a chain per set made the
    // ordering of a row part's rays cost 0.7 ms of launch latency per frame
    unsigned set_bits = 0;
    while ((1 << set_bits) < sets) ++set_bits;
    const unsigned end_bit = kRayKeyBits + set_bits;
    size_t temp_bytes = 0;
    if (rocprim::radix_sort_pairs(nullptr, temp_bytes, (unsigned*)nullptr, (unsigned*)nullptr,
                                  (int32_t*)nullptr, (int32_t*)nullptr, (unsigned)total, 0u, end_bit, st) != hipSuccess) {
        rt_set_error("ray sort: size query failed");
        return RT_E_HIP;
    }
    const size_t arr = ((size_t)total * sizeof(unsigned) + 255) & ~(size_t)255;
    if (int rc = sc->ws_sort.reserve(4 * arr + temp_bytes)) return rc;
    char* ws = sc->ws_sort.as<char>();
    unsigned* keys_in = reinterpret_cast<unsigned*>(ws);
    unsigned* keys_out = reinterpret_cast<unsigned*>(ws + arr);
    int32_t* vals_in = reinterpret_cast<int32_t*>(ws + 2 * arr);
    int32_t* vals_out = reinterpret_cast<int32_t*>(ws + 3 * arr);
    void* temp = ws + 4 * arr;
    unsigned wgs = (largest + RT_WG - 1) / RT_WG;
    if (wgs > 2048) wgs = 2048;
    // Hilbert order over (u, v).  RT_AMD_RAY_ORDER=u: u-major instead, which suits the all-pairs
    // sweep's first test (it looks at u alone; a wave's 512 rays then span a few 1e-4 of the u
    // range): 1.5 ms less on the 1 M-triangle soup's 65 ms of shadow sweeps, 0.4 ms more on the
    // 8-11 ms of closed 70-100 k-triangle meshes, and worse for the culled sweep, which compares a
    // wave's (u, v) box with the squares -- not adopted.
    const char* order_env = getenv("RT_AMD_RAY_ORDER");
    const int u_major = order_env && order_env[0] == 'u';
    hipLaunchKernelGGL(k_ray_keys, dim3(wgs, sets), dim3(RT_WG), 0, st, S, E, ids_in, counts_dev, stride, ka,
                       u_major, keys_in, vals_in);
    RT_HIP(hipGetLastError());
    RT_HIP(rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, (unsigned)total, 0u,
                                     end_bit, st));
    hipLaunchKernelGGL(k_unpack_sets, dim3(wgs, sets), dim3(RT_WG), 0, st, vals_out, counts_dev, stride, ka,
                       ids_out);
    RT_HIP(hipGetLastError());
    return RT_OK;
}
