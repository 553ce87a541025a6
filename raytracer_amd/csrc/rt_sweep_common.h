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

}  // namespace rt_sweep
