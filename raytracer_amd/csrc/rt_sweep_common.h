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

// Inclusive scans over the wave's 64 lanes with data-parallel-primitive moves (no LDS traffic):
// shifts by 1, 2, 4, 8 lanes inside every row of 16, then the last lane of rows 0 and 2 broadcast
// into rows 1 and 3, then lane 31 into the upper half.  A lane without a source keeps `identity`.
#define RT_DPP_SCAN(v, identity, OP)                                                                  \
    do {                                                                                              \
        v = OP(v, __builtin_amdgcn_update_dpp((identity), v, 0x111, 0xf, 0xf, false)); /* row_shr:1 */ \
        v = OP(v, __builtin_amdgcn_update_dpp((identity), v, 0x112, 0xf, 0xf, false)); /* row_shr:2 */ \
        v = OP(v, __builtin_amdgcn_update_dpp((identity), v, 0x114, 0xf, 0xf, false)); /* row_shr:4 */ \
        v = OP(v, __builtin_amdgcn_update_dpp((identity), v, 0x118, 0xf, 0xf, false)); /* row_shr:8 */ \
        v = OP(v, __builtin_amdgcn_update_dpp((identity), v, 0x142, 0xa, 0xf, false)); /* row_bcast:15 */ \
        v = OP(v, __builtin_amdgcn_update_dpp((identity), v, 0x143, 0xc, 0xf, false)); /* row_bcast:31 */ \
    } while (0)
__device__ __forceinline__ int rt_op_add(int a, int b) { return a + b; }
__device__ __forceinline__ int rt_op_max(int a, int b) { return a > b ? a : b; }

// Running sum over the wave's 64 lanes (inclusive).
__device__ __forceinline__ int wave_scan(int v, int lane)
{
    (void)lane;
    RT_DPP_SCAN(v, 0, rt_op_add);
    return v;
}

// Running maximum over the wave's 64 lanes (inclusive).
__device__ __forceinline__ int wave_scan_max(int v)
{
    RT_DPP_SCAN(v, (int)0x80000000, rt_op_max);
    return v;
}

// Between a lane's LDS stores and other lanes' loads of them within one wave: no barrier is needed
// in hardware (a wave's LDS accesses complete in program order), but the compiler has to be told
// that the stores are for other lanes and the loads may see theirs.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Dealing items to lanes.  Lane q owns n_q consecutive items, the first one numbered excl_q (the
// running sum of the n before it).  For the window of 64 items from p0 on, returns to every lane
// the lane that owns item p0 + lane: every owner whose first item lies in the window leaves its
// number at that item's place (LDS, 64 ints of the wave's own), and a running maximum carries it
// over the items behind (owners ascend with the items).  carry: the owner of item p0 - 1 (-1 for
// the first window), which owns the window's first items if no run starts there; wave-uniform.
// An item beyond the last one gets the last owner.
// (The lanes exchange values through LDS without a barrier -- a wave's LDS accesses complete in
// program order -- so the accesses are volatile and fenced at wave scope: to the compiler a lane's
// own store and load would otherwise be all there is.)
__device__ __forceinline__ int wave_owner(int excl, int n, int p0, int* flags_, int carry, int lane)
{
    // (an LDS pointer to the compiler too: volatile accesses through a generic pointer stay generic)
    volatile __attribute__((address_space(3))) int* flags = (volatile __attribute__((address_space(3))) int*)flags_;
    flags[lane] = -1;
    const int at = excl - p0;
    if (n > 0 && at >= 0 && at < 64) flags[at] = lane;
    wave_lds_sync();
    const int f = wave_scan_max(flags[lane]);
    return f > carry ? f : carry;
}

}  // namespace rt_sweep
