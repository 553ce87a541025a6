// rt_sweep_common.h -- pieces shared by the sweep kernels (rt_sweep.hip: rays in general
// position; rt_sweep_apex.hip: rays through a common point).
#pragma once

#include "rt_internal.h"

namespace rt_sweep {

constexpr float INF = __builtin_huge_valf();
constexpr float U20 = 9.5367431640625e-07f;     // 2^-20
constexpr float U22 = 2.384185791015625e-07f;   // 2^-22

__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz)
{
    return __builtin_fmaf(az, bz, __builtin_fmaf(ay, by, ax * bx));
}

// Makes a value opaque to the optimiser so that per-ray quantities that happen to be equal
// in every lane are still carried and used per ray.
__device__ __forceinline__ float opaque(float x)
{
    asm volatile("" : "+v"(x));
    return x;
}

// Per-ray bookkeeping of a sweep.
struct Book {
    float best;      // smallest t of a certain hit so far (closest-hit mode)
    float best_hi;   // pruning threshold: upper bound of that hit's true t; -inf = ray retired/invalid
    int cnt;         // candidates pushed so far, plus the RT_CNT_* flag bits
};

__device__ __forceinline__ void push(Book& r, int32_t* __restrict__ list_slot, int tri)
{
    const int slot = r.cnt & RT_CNT_MASK;
    if (slot < RT_LIST_K) list_slot[slot] = tri;
    r.cnt++;
}

// One pair that passed the widened inside test; rare (a few per ray per sweep).
//   a = U*d, b = V*d, w = d*d - a - b with d the determinant; the pair is certainly inside
//   when min(a,b,w) >= margin*|d| and |d| exceeds its own error bound en.
//   The ray parameter is t = t0 + T/d with |error of T| <= eT.  Its error bound
//       terr = (eT + |T/d| en) / (|d| - en) + 2^-22 (|t0| + |T/d|)
//   is rigorous, so "t - terr > best_hi" proves the pair lies behind a certain hit.
// A pair whose determinant is not resolved (|d| <= en) is handed to the exact pass as is.
template <bool ANYHIT>
__device__ __forceinline__ void candidate(Book& r, int32_t* __restrict__ list_slot, int tri,
                                          float a, float b, float w, float d, float T, float t0,
                                          float margin, float en, float eT)
{
    if (r.best_hi == -INF) return;
    const float ad = __builtin_fabsf(d);
    const float den = ad - en;
    if (!(den > 0.0f)) {
        push(r, list_slot, tri);
        return;
    }
    const float q = T / d;
    const float t = t0 + q;
    const float terr = (eT + __builtin_fabsf(q) * en) / den * 1.000001f +
                       U22 * (__builtin_fabsf(t0) + __builtin_fabsf(q));
    if (!(t >= -terr) || !(t - terr <= r.best_hi)) return;
    push(r, list_slot, tri);
    const bool certain = (__builtin_fminf(__builtin_fminf(a, b), w) >= margin * ad) && (t > terr);
    if (certain) {
        if (ANYHIT) {
            r.cnt |= RT_CNT_OCCLUDED;
            r.best_hi = -INF;
        } else if (t < r.best) {
            r.best = t;
            r.best_hi = t + terr;
        }
    }
}

// Stages tile t (RT_TILE triangles of NPLANES float4 planes) into one LDS buffer.  One LDS-DMA
// instruction moves 64 float4 (1 KiB) per wave: global source per lane, LDS destination =
// wave-uniform base + 16 * lane.
template <int NPLANES>
__device__ __forceinline__ void stage_tile(const float4* __restrict__ planes, long plane_stride,
                                           int t, float4 (*buf)[RT_TILE], int wave, int lane)
{
#pragma unroll
    for (int c = 0; c < (NPLANES * (RT_TILE / 64) + 3) / 4; ++c) {
        const int chunk = c * 4 + wave;                 // chunk = plane * (RT_TILE/64) + part
        if (chunk < NPLANES * (RT_TILE / 64)) {
            const int plane = chunk / (RT_TILE / 64), part = chunk % (RT_TILE / 64);
            const long g = (long)plane * plane_stride + (long)t * RT_TILE + part * 64 + lane;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(planes + g),
                (__attribute__((address_space(3))) void*)&buf[plane][part * 64], 16, 0, 0);
        }
    }
}

// LDS by its 32-bit address (a float4 of HIP is a class, which a pointer into another address
// space cannot be dereferenced as; the native vector can)
typedef float native_float4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned lds_address(const void* p)
{
    return (unsigned)(size_t)(const __attribute__((address_space(3))) void*)p;
}
__device__ __forceinline__ float4 lds_read_float4(unsigned at)
{
    const native_float4 v = *(const __attribute__((address_space(3))) native_float4*)(size_t)at;
    return make_float4(v.x, v.y, v.z, v.w);
}

// min(a, |b|, |c|) as the one instruction it is (the optimiser otherwise turns the chain of these
// over a lane's pixels into separate two-operand minima)
__device__ __forceinline__ float min3_abs(float a, float b, float c)
{
    float r;
    asm("v_min3_f32 %0, %1, |%2|, |%3|" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// max(|a|, |b|) and min(a, b, c), likewise one instruction each
__device__ __forceinline__ float max_abs(float a, float b)
{
    float r;
    asm("v_max_f32_e64 %0, |%1|, |%2|" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float min3_plain(float a, float b, float c)
{
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// The counterpart of candidate() for an any-hit sweep whose triangle range is shared by several
// workgroups per ray: the candidate count and the occluded flag of a ray live in global memory
// only and are updated atomically.  The order of a ray's list entries is then arbitrary, which
// an any-hit query does not care about (triangle_intersection.cu:169 asks for any dist > 0).
// Returns true when the pair is a certain hit (the ray retires).
__device__ __forceinline__ bool candidate_shared(int32_t* __restrict__ cnt_slot,
                                                 int32_t* __restrict__ list_slot,
                                                 int32_t* __restrict__ occ_slot, int tri, float a,
                                                 float b, float w, float d, float T, float t0,
                                                 float margin, float en, float eT)
{
    const float ad = __builtin_fabsf(d);
    const float den = ad - en;
    bool certain = false;
    if (den > 0.0f) {
        const float q = T / d;
        const float t = t0 + q;
        const float terr = (eT + __builtin_fabsf(q) * en) / den * 1.000001f +
                           U22 * (__builtin_fabsf(t0) + __builtin_fabsf(q));
        if (!(t >= -terr)) return false;
        certain = (__builtin_fminf(__builtin_fminf(a, b), w) >= margin * ad) && (t > terr);
    }
    if (certain) {
        atomicOr(cnt_slot, RT_CNT_OCCLUDED);    // the list no longer matters
        if (occ_slot) *occ_slot = tri;          // for the occluder cache (any certain occluder will do)
        return true;
    }
    const int at = atomicAdd(cnt_slot, 1) & RT_CNT_MASK;
    if (at < RT_LIST_K) list_slot[at] = tri;
    return false;
}

// Running sum over the wave's 64 lanes (inclusive).
__device__ __forceinline__ int wave_scan(int v, int lane)
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(v, off);
        v += lane >= off ? t : 0;
    }
    return v;
}

// The first lane whose running sum `incl` exceeds r (63 when none does).
__device__ __forceinline__ int wave_search(int incl, int r)
{
    int l = 0;
#pragma unroll
    for (int step = 32; step > 0; step >>= 1) {
        const int v = __shfl(incl, l + step - 1);
        l += v <= r ? step : 0;
    }
    return l > 63 ? 63 : l;
}

}  // namespace rt_sweep
