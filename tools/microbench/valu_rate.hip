// Issue rate of VALU instructions on gfx950 (development aid): 8 independent chains per wave,
// 8 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define KERNEL(name, body)                                                                        \
    __global__ void __launch_bounds__(256) name(float* out, int iters, float a, float b)         \
    {                                                                                             \
        float v0 = threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5,  \
              v6 = v0 + 6, v7 = v0 + 7;                                                            \
        for (int i = 0; i < iters; ++i)                                                           \
            asm volatile(REP8(body)                                                               \
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6),  \
                           "+v"(v7)                                                               \
                         : "v"(a), "v"(b) : "vcc");                                               \
        out[blockIdx.x * 256 + threadIdx.x] = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;              \
    }
KERNEL(k_fma, "v_fma_f32 %0, %0, %8, %9\n" "v_fma_f32 %1, %1, %8, %9\n" "v_fma_f32 %2, %2, %8, %9\n" "v_fma_f32 %3, %3, %8, %9\n" "v_fma_f32 %4, %4, %8, %9\n" "v_fma_f32 %5, %5, %8, %9\n" "v_fma_f32 %6, %6, %8, %9\n" "v_fma_f32 %7, %7, %8, %9\n" )
KERNEL(k_mul, "v_mul_f32 %0, %0, %8\n" "v_mul_f32 %1, %1, %8\n" "v_mul_f32 %2, %2, %8\n" "v_mul_f32 %3, %3, %8\n" "v_mul_f32 %4, %4, %8\n" "v_mul_f32 %5, %5, %8\n" "v_mul_f32 %6, %6, %8\n" "v_mul_f32 %7, %7, %8\n" )
KERNEL(k_add, "v_add_f32 %0, %0, %8\n" "v_add_f32 %1, %1, %8\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %8\n" "v_add_f32 %4, %4, %8\n" "v_add_f32 %5, %5, %8\n" "v_add_f32 %6, %6, %8\n" "v_add_f32 %7, %7, %8\n" )
KERNEL(k_mac, "v_fmac_f32 %0, %8, %9\n" "v_fmac_f32 %1, %8, %9\n" "v_fmac_f32 %2, %8, %9\n" "v_fmac_f32 %3, %8, %9\n" "v_fmac_f32 %4, %8, %9\n" "v_fmac_f32 %5, %8, %9\n" "v_fmac_f32 %6, %8, %9\n" "v_fmac_f32 %7, %8, %9\n" )
KERNEL(k_and, "v_and_b32 %0, %0, %8\n" "v_and_b32 %1, %1, %8\n" "v_and_b32 %2, %2, %8\n" "v_and_b32 %3, %3, %8\n" "v_and_b32 %4, %4, %8\n" "v_and_b32 %5, %5, %8\n" "v_and_b32 %6, %6, %8\n" "v_and_b32 %7, %7, %8\n" )
KERNEL(k_or3, "v_or3_b32 %0, %0, %8, %9\n" "v_or3_b32 %1, %1, %8, %9\n" "v_or3_b32 %2, %2, %8, %9\n" "v_or3_b32 %3, %3, %8, %9\n" "v_or3_b32 %4, %4, %8, %9\n" "v_or3_b32 %5, %5, %8, %9\n" "v_or3_b32 %6, %6, %8, %9\n" "v_or3_b32 %7, %7, %8, %9\n" )
KERNEL(k_and_or, "v_and_or_b32 %0, %0, %8, %9\n" "v_and_or_b32 %1, %1, %8, %9\n" "v_and_or_b32 %2, %2, %8, %9\n" "v_and_or_b32 %3, %3, %8, %9\n" "v_and_or_b32 %4, %4, %8, %9\n" "v_and_or_b32 %5, %5, %8, %9\n" "v_and_or_b32 %6, %6, %8, %9\n" "v_and_or_b32 %7, %7, %8, %9\n" )
KERNEL(k_max3, "v_max3_f32 %0, %0, %8, %9\n" "v_max3_f32 %1, %1, %8, %9\n" "v_max3_f32 %2, %2, %8, %9\n" "v_max3_f32 %3, %3, %8, %9\n" "v_max3_f32 %4, %4, %8, %9\n" "v_max3_f32 %5, %5, %8, %9\n" "v_max3_f32 %6, %6, %8, %9\n" "v_max3_f32 %7, %7, %8, %9\n" )
KERNEL(k_max, "v_max_f32 %0, %0, %8\n" "v_max_f32 %1, %1, %8\n" "v_max_f32 %2, %2, %8\n" "v_max_f32 %3, %3, %8\n" "v_max_f32 %4, %4, %8\n" "v_max_f32 %5, %5, %8\n" "v_max_f32 %6, %6, %8\n" "v_max_f32 %7, %7, %8\n" )
KERNEL(k_min, "v_min_f32 %0, %0, %8\n" "v_min_f32 %1, %1, %8\n" "v_min_f32 %2, %2, %8\n" "v_min_f32 %3, %3, %8\n" "v_min_f32 %4, %4, %8\n" "v_min_f32 %5, %5, %8\n" "v_min_f32 %6, %6, %8\n" "v_min_f32 %7, %7, %8\n" )
KERNEL(k_max_i32, "v_max_i32 %0, %0, %8\n" "v_max_i32 %1, %1, %8\n" "v_max_i32 %2, %2, %8\n" "v_max_i32 %3, %3, %8\n" "v_max_i32 %4, %4, %8\n" "v_max_i32 %5, %5, %8\n" "v_max_i32 %6, %6, %8\n" "v_max_i32 %7, %7, %8\n" )
KERNEL(k_max3_i32, "v_max3_i32 %0, %0, %8, %9\n" "v_max3_i32 %1, %1, %8, %9\n" "v_max3_i32 %2, %2, %8, %9\n" "v_max3_i32 %3, %3, %8, %9\n" "v_max3_i32 %4, %4, %8, %9\n" "v_max3_i32 %5, %5, %8, %9\n" "v_max3_i32 %6, %6, %8, %9\n" "v_max3_i32 %7, %7, %8, %9\n" )
KERNEL(k_add_u32, "v_add_u32 %0, %0, %8\n" "v_add_u32 %1, %1, %8\n" "v_add_u32 %2, %2, %8\n" "v_add_u32 %3, %3, %8\n" "v_add_u32 %4, %4, %8\n" "v_add_u32 %5, %5, %8\n" "v_add_u32 %6, %6, %8\n" "v_add_u32 %7, %7, %8\n" )
KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %8, vcc\n" "v_cndmask_b32 %1, %1, %8, vcc\n" "v_cndmask_b32 %2, %2, %8, vcc\n" "v_cndmask_b32 %3, %3, %8, vcc\n" "v_cndmask_b32 %4, %4, %8, vcc\n" "v_cndmask_b32 %5, %5, %8, vcc\n" "v_cndmask_b32 %6, %6, %8, vcc\n" "v_cndmask_b32 %7, %7, %8, vcc\n" )
KERNEL(k_cmp, "v_cmp_ge_f32 vcc, %0, %8\n" "v_cmp_ge_f32 vcc, %1, %8\n" "v_cmp_ge_f32 vcc, %2, %8\n" "v_cmp_ge_f32 vcc, %3, %8\n" "v_cmp_ge_f32 vcc, %4, %8\n" "v_cmp_ge_f32 vcc, %5, %8\n" "v_cmp_ge_f32 vcc, %6, %8\n" "v_cmp_ge_f32 vcc, %7, %8\n" )
KERNEL(k_mix_2fma_andor, "v_fma_f32 %0, %0, %8, %9\n" "v_fma_f32 %1, %1, %8, %9\n" "v_and_or_b32 %2, %0, %1, %2\n" "v_fma_f32 %3, %3, %8, %9\n" "v_fma_f32 %4, %4, %8, %9\n" "v_and_or_b32 %5, %3, %4, %5\n" "v_fma_f32 %6, %6, %8, %9\n" "v_fma_f32 %7, %7, %8, %9\n" )
KERNEL(k_mix_6fma_max3, "v_fma_f32 %0, %0, %8, %9\n" "v_fma_f32 %1, %1, %8, %9\n" "v_fma_f32 %2, %2, %8, %9\n" "v_fma_f32 %3, %3, %8, %9\n" "v_fma_f32 %4, %4, %8, %9\n" "v_fma_f32 %5, %5, %8, %9\n" "v_max3_f32 %6, %6, %2, %5\n" "v_fma_f32 %7, %7, %8, %9\n" )
KERNEL(k_mix_fma_and, "v_fma_f32 %0, %0, %8, %9\n" "v_and_b32 %1, %1, %8\n" "v_fma_f32 %2, %2, %8, %9\n" "v_and_b32 %3, %3, %8\n" "v_fma_f32 %4, %4, %8, %9\n" "v_and_b32 %5, %5, %8\n" "v_fma_f32 %6, %6, %8, %9\n" "v_and_b32 %7, %7, %8\n" )
KERNEL(k_mix_fma_add, "v_fma_f32 %0, %0, %8, %9\n" "v_add_f32 %1, %1, %8\n" "v_fma_f32 %2, %2, %8, %9\n" "v_add_f32 %3, %3, %8\n" "v_fma_f32 %4, %4, %8, %9\n" "v_add_f32 %5, %5, %8\n" "v_fma_f32 %6, %6, %8, %9\n" "v_add_f32 %7, %7, %8\n" )

#define KERNEL2(name, body)                                                                       \
    __global__ void __launch_bounds__(256) name(float* out, int iters, float a, float b)         \
    {                                                                                             \
        typedef float f2 __attribute__((ext_vector_type(2)));                                     \
        f2 v0 = {(float)threadIdx.x, 1}, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4,     \
           v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7, va = {a, a}, vb = {b, b};                       \
        for (int i = 0; i < iters; ++i)                                                           \
            asm volatile(REP8(body)                                                               \
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6),  \
                           "+v"(v7)                                                               \
                         : "v"(va), "v"(vb));                                                     \
        f2 s = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;                                             \
        out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;                                          \
    }
KERNEL2(k_pk_add, "v_pk_add_f32 %0, %0, %8\n" "v_pk_add_f32 %1, %1, %8\n" "v_pk_add_f32 %2, %2, %8\n" "v_pk_add_f32 %3, %3, %8\n" "v_pk_add_f32 %4, %4, %8\n" "v_pk_add_f32 %5, %5, %8\n" "v_pk_add_f32 %6, %6, %8\n" "v_pk_add_f32 %7, %7, %8\n" )
KERNEL2(k_pk_fma, "v_pk_fma_f32 %0, %0, %8, %9\n" "v_pk_fma_f32 %1, %1, %8, %9\n" "v_pk_fma_f32 %2, %2, %8, %9\n" "v_pk_fma_f32 %3, %3, %8, %9\n" "v_pk_fma_f32 %4, %4, %8, %9\n" "v_pk_fma_f32 %5, %5, %8, %9\n" "v_pk_fma_f32 %6, %6, %8, %9\n" "v_pk_fma_f32 %7, %7, %8, %9\n" )
KERNEL2(k_pk_mul, "v_pk_mul_f32 %0, %0, %8\n" "v_pk_mul_f32 %1, %1, %8\n" "v_pk_mul_f32 %2, %2, %8\n" "v_pk_mul_f32 %3, %3, %8\n" "v_pk_mul_f32 %4, %4, %8\n" "v_pk_mul_f32 %5, %5, %8\n" "v_pk_mul_f32 %6, %6, %8\n" "v_pk_mul_f32 %7, %7, %8\n" )

#define KERNEL1(name, body)                                                                       \
    __global__ void __launch_bounds__(256) name(float* out, int iters, float a, float b)         \
    {                                                                                             \
        float v0 = threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5,  \
              v6 = v0 + 6, v7 = v0 + 7;                                                            \
        for (int i = 0; i < iters; ++i)                                                           \
            asm volatile(body                                                                     \
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6),  \
                           "+v"(v7)                                                               \
                         : "v"(a), "v"(b) : "vcc");                                               \
        out[blockIdx.x * 256 + threadIdx.x] = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;              \
    }
// the hot loop of the pixel-grid sweep for one triangle and 16 pixels, and of the any-hit sweep for
// one triangle and 8 rays (no LDS reads, no branch): what the instruction mix alone costs
KERNEL1(k_raster16, "v_fma_f32 %0, %8, %9, %0\n" "v_fma_f32 %1, %8, %9, %1\n" "v_fma_f32 %2, %8, %9, %0\n" "v_fma_f32 %3, %8, %9, %1\n" "v_and_b32 %4, %2, %3\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_fma_f32 %2, %8, %9, %0\n" "v_fma_f32 %3, %8, %9, %1\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_cmp_gt_i32 vcc, 0, %4\n" )
KERNEL1(k_raster32, "v_fma_f32 %0, %8, %9, %0\n" "v_fma_f32 %1, %8, %9, %1\n" "v_fma_f32 %2, %8, %9, %0\n" "v_fma_f32 %3, %8, %9, %1\n" "v_and_b32 %4, %2, %3\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_fma_f32 %2, %8, %9, %0\n" "v_fma_f32 %3, %8, %9, %1\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_fma_f32 %2, %8, %9, %0\n" "v_fma_f32 %3, %8, %9, %1\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_fma_f32 %2, %8, %9, %0\n" "v_fma_f32 %3, %8, %9, %1\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_cmp_gt_i32 vcc, 0, %4\n" )
KERNEL1(k_raster64, "v_fma_f32 %0, %8, %9, %0\n" "v_fma_f32 %1, %8, %9, %1\n" "v_fma_f32 %2, %8, %9, %0\n" "v_fma_f32 %3, %8, %9, %1\n" "v_and_b32 %4, %2, %3\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_fma_f32 %2, %8, %9, %0\n" "v_fma_f32 %3, %8, %9, %1\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_fma_f32 %2, %8, %9, %0\n" "v_fma_f32 %3, %8, %9, %1\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_fma_f32 %2, %8, %9, %0\n" "v_fma_f32 %3, %8, %9, %1\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_fma_f32 %2, %8, %9, %0\n" "v_fma_f32 %3, %8, %9, %1\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_fma_f32 %2, %8, %9, %0\n" "v_fma_f32 %3, %8, %9, %1\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_fma_f32 %2, %8, %9, %0\n" "v_fma_f32 %3, %8, %9, %1\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_fma_f32 %2, %8, %9, %0\n" "v_fma_f32 %3, %8, %9, %1\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %9\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_cmp_gt_i32 vcc, 0, %4\n" )
KERNEL1(k_apex8, "v_fma_f32 %2, %8, %9, %0\n" "v_fma_f32 %3, %8, %9, %1\n" "v_fmac_f32 %2, %9, %8\n" "v_fmac_f32 %3, %9, %8\n" "v_fma_f32 %5, %2, %3, %0\n" "v_fma_f32 %2, %8, %9, %0\n" "v_fma_f32 %3, %8, %9, %1\n" "v_fmac_f32 %2, %9, %8\n" "v_fmac_f32 %3, %9, %8\n" "v_fma_f32 %6, %2, %3, %0\n" "v_fma_f32 %2, %8, %9, %0\n" "v_fma_f32 %3, %8, %9, %1\n" "v_fmac_f32 %2, %9, %8\n" "v_fmac_f32 %3, %9, %8\n" "v_fma_f32 %7, %2, %3, %0\n" "v_fma_f32 %2, %8, %9, %0\n" "v_fma_f32 %3, %8, %9, %1\n" "v_fmac_f32 %2, %9, %8\n" "v_fmac_f32 %3, %9, %8\n" "v_fma_f32 %5, %2, %3, %0\n" "v_fma_f32 %2, %8, %9, %0\n" "v_fma_f32 %3, %8, %9, %1\n" "v_fmac_f32 %2, %9, %8\n" "v_fmac_f32 %3, %9, %8\n" "v_fma_f32 %6, %2, %3, %0\n" "v_fma_f32 %2, %8, %9, %0\n" "v_fma_f32 %3, %8, %9, %1\n" "v_fmac_f32 %2, %9, %8\n" "v_fmac_f32 %3, %9, %8\n" "v_fma_f32 %7, %2, %3, %0\n" "v_fma_f32 %2, %8, %9, %0\n" "v_fma_f32 %3, %8, %9, %1\n" "v_fmac_f32 %2, %9, %8\n" "v_fmac_f32 %3, %9, %8\n" "v_fma_f32 %5, %2, %3, %0\n" "v_fma_f32 %2, %8, %9, %0\n" "v_fma_f32 %3, %8, %9, %1\n" "v_fmac_f32 %2, %9, %8\n" "v_fmac_f32 %3, %9, %8\n" "v_fma_f32 %6, %2, %3, %0\n" "v_max_f32 %4, %5, %6\n" "v_max3_f32 %4, %4, %7, %5\n" "v_max3_f32 %4, %4, %6, %7\n" "v_max3_f32 %4, %4, %5, %6\n" "v_cmp_le_f32 vcc, 0, %4\n" )
// the any-hit loop if it used the sign tests of the pixel-grid sweep (not adopted: shadow rays have no fixed sign of the determinant)
KERNEL1(k_apex8_sign, "v_fma_f32 %2, %8, %9, %0\n" "v_fma_f32 %3, %8, %9, %1\n" "v_fmac_f32 %2, %9, %8\n" "v_fmac_f32 %3, %9, %8\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_fma_f32 %2, %8, %9, %0\n" "v_fma_f32 %3, %8, %9, %1\n" "v_fmac_f32 %2, %9, %8\n" "v_fmac_f32 %3, %9, %8\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_fma_f32 %2, %8, %9, %0\n" "v_fma_f32 %3, %8, %9, %1\n" "v_fmac_f32 %2, %9, %8\n" "v_fmac_f32 %3, %9, %8\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_fma_f32 %2, %8, %9, %0\n" "v_fma_f32 %3, %8, %9, %1\n" "v_fmac_f32 %2, %9, %8\n" "v_fmac_f32 %3, %9, %8\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_fma_f32 %2, %8, %9, %0\n" "v_fma_f32 %3, %8, %9, %1\n" "v_fmac_f32 %2, %9, %8\n" "v_fmac_f32 %3, %9, %8\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_fma_f32 %2, %8, %9, %0\n" "v_fma_f32 %3, %8, %9, %1\n" "v_fmac_f32 %2, %9, %8\n" "v_fmac_f32 %3, %9, %8\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_fma_f32 %2, %8, %9, %0\n" "v_fma_f32 %3, %8, %9, %1\n" "v_fmac_f32 %2, %9, %8\n" "v_fmac_f32 %3, %9, %8\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_fma_f32 %2, %8, %9, %0\n" "v_fma_f32 %3, %8, %9, %1\n" "v_fmac_f32 %2, %9, %8\n" "v_fmac_f32 %3, %9, %8\n" "v_and_or_b32 %4, %2, %3, %4\n" "v_cmp_gt_i32 vcc, 0, %4\n" )
// candidate first tests of the any-hit sweep on the projection plane of the dominant axis: the
// bounding square (max(|du|, |dv|) <= rho) and the bounding circle (du^2 + dv^2 <= rho^2) of the
// projected triangle, 8 rays per lane
KERNEL1(k_apex8_box, "v_sub_f32 %2, %8, %0\n" "v_sub_f32 %3, %9, %1\n" "v_max_f32_e64 %5, |%2|, |%3|\n" "v_sub_f32 %2, %8, %0\n" "v_sub_f32 %3, %9, %1\n" "v_max_f32_e64 %6, |%2|, |%3|\n" "v_min3_f32 %4, %4, %5, %6\n" "v_sub_f32 %2, %8, %0\n" "v_sub_f32 %3, %9, %1\n" "v_max_f32_e64 %5, |%2|, |%3|\n" "v_sub_f32 %2, %8, %0\n" "v_sub_f32 %3, %9, %1\n" "v_max_f32_e64 %6, |%2|, |%3|\n" "v_min3_f32 %4, %4, %5, %6\n" "v_sub_f32 %2, %8, %0\n" "v_sub_f32 %3, %9, %1\n" "v_max_f32_e64 %5, |%2|, |%3|\n" "v_sub_f32 %2, %8, %0\n" "v_sub_f32 %3, %9, %1\n" "v_max_f32_e64 %6, |%2|, |%3|\n" "v_min3_f32 %4, %4, %5, %6\n" "v_sub_f32 %2, %8, %0\n" "v_sub_f32 %3, %9, %1\n" "v_max_f32_e64 %5, |%2|, |%3|\n" "v_sub_f32 %2, %8, %0\n" "v_sub_f32 %3, %9, %1\n" "v_max_f32_e64 %6, |%2|, |%3|\n" "v_min3_f32 %4, %4, %5, %6\n" "v_cmp_ge_f32 vcc, %7, %4\n" )
KERNEL1(k_apex8_circle, "v_sub_f32 %2, %8, %0\n" "v_sub_f32 %3, %9, %1\n" "v_mul_f32 %5, %2, %2\n" "v_fmac_f32 %5, %3, %3\n" "v_sub_f32 %2, %8, %0\n" "v_sub_f32 %3, %9, %1\n" "v_mul_f32 %6, %2, %2\n" "v_fmac_f32 %6, %3, %3\n" "v_min3_f32 %4, %4, %5, %6\n" "v_sub_f32 %2, %8, %0\n" "v_sub_f32 %3, %9, %1\n" "v_mul_f32 %5, %2, %2\n" "v_fmac_f32 %5, %3, %3\n" "v_sub_f32 %2, %8, %0\n" "v_sub_f32 %3, %9, %1\n" "v_mul_f32 %6, %2, %2\n" "v_fmac_f32 %6, %3, %3\n" "v_min3_f32 %4, %4, %5, %6\n" "v_sub_f32 %2, %8, %0\n" "v_sub_f32 %3, %9, %1\n" "v_mul_f32 %5, %2, %2\n" "v_fmac_f32 %5, %3, %3\n" "v_sub_f32 %2, %8, %0\n" "v_sub_f32 %3, %9, %1\n" "v_mul_f32 %6, %2, %2\n" "v_fmac_f32 %6, %3, %3\n" "v_min3_f32 %4, %4, %5, %6\n" "v_sub_f32 %2, %8, %0\n" "v_sub_f32 %3, %9, %1\n" "v_mul_f32 %5, %2, %2\n" "v_fmac_f32 %5, %3, %3\n" "v_sub_f32 %2, %8, %0\n" "v_sub_f32 %3, %9, %1\n" "v_mul_f32 %6, %2, %2\n" "v_fmac_f32 %6, %3, %3\n" "v_min3_f32 %4, %4, %5, %6\n" "v_cmp_ge_f32 vcc, %7, %4\n" )
// candidate first test of the pixel-grid sweep: each of a lane's 64 pixels against the triangle's
// bounding box in x (the distance from the box's centre steps by one from pixel to pixel; a
// three-way minimum takes two pixels), the lane's row against it in y: 99 instructions
KERNEL1(k_raster64_box, "v_sub_f32 %2, %8, %0\n" "v_sub_f32 %6, %9, %1\n" "v_add_f32 %3, 1.0, %2\n" "v_min3_f32 %4, %4, |%2|, |%3|\n" "v_add_f32 %3, 1.0, %2\n" "v_add_f32 %2, 1.0, %3\n" "v_min3_f32 %4, %4, |%3|, |%2|\n" "v_add_f32 %3, 1.0, %2\n" "v_add_f32 %2, 1.0, %3\n" "v_min3_f32 %4, %4, |%3|, |%2|\n" "v_add_f32 %3, 1.0, %2\n" "v_add_f32 %2, 1.0, %3\n" "v_min3_f32 %4, %4, |%3|, |%2|\n" "v_add_f32 %3, 1.0, %2\n" "v_add_f32 %2, 1.0, %3\n" "v_min3_f32 %4, %4, |%3|, |%2|\n" "v_add_f32 %3, 1.0, %2\n" "v_add_f32 %2, 1.0, %3\n" "v_min3_f32 %4, %4, |%3|, |%2|\n" "v_add_f32 %3, 1.0, %2\n" "v_add_f32 %2, 1.0, %3\n" "v_min3_f32 %4, %4, |%3|, |%2|\n" "v_add_f32 %3, 1.0, %2\n" "v_add_f32 %2, 1.0, %3\n" "v_min3_f32 %4, %4, |%3|, |%2|\n" "v_add_f32 %3, 1.0, %2\n" "v_add_f32 %2, 1.0, %3\n" "v_min3_f32 %4, %4, |%3|, |%2|\n" "v_add_f32 %3, 1.0, %2\n" "v_add_f32 %2, 1.0, %3\n" "v_min3_f32 %4, %4, |%3|, |%2|\n" "v_add_f32 %3, 1.0, %2\n" "v_add_f32 %2, 1.0, %3\n" "v_min3_f32 %4, %4, |%3|, |%2|\n" "v_add_f32 %3, 1.0, %2\n" "v_add_f32 %2, 1.0, %3\n" "v_min3_f32 %4, %4, |%3|, |%2|\n" "v_add_f32 %3, 1.0, %2\n" "v_add_f32 %2, 1.0, %3\n" "v_min3_f32 %4, %4, |%3|, |%2|\n" "v_add_f32 %3, 1.0, %2\n" "v_add_f32 %2, 1.0, %3\n" "v_min3_f32 %4, %4, |%3|, |%2|\n" "v_add_f32 %3, 1.0, %2\n" "v_add_f32 %2, 1.0, %3\n" "v_min3_f32 %4, %4, |%3|, |%2|\n" "v_add_f32 %3, 1.0, %2\n" "v_add_f32 %2, 1.0, %3\n" "v_min3_f32 %4, %4, |%3|, |%2|\n" "v_add_f32 %3, 1.0, %2\n" "v_add_f32 %2, 1.0, %3\n" "v_min3_f32 %4, %4, |%3|, |%2|\n" "v_add_f32 %3, 1.0, %2\n" "v_add_f32 %2, 1.0, %3\n" "v_min3_f32 %4, %4, |%3|, |%2|\n" "v_add_f32 %3, 1.0, %2\n" "v_add_f32 %2, 1.0, %3\n" "v_min3_f32 %4, %4, |%3|, |%2|\n" "v_add_f32 %3, 1.0, %2\n" "v_add_f32 %2, 1.0, %3\n" "v_min3_f32 %4, %4, |%3|, |%2|\n" "v_add_f32 %3, 1.0, %2\n" "v_add_f32 %2, 1.0, %3\n" "v_min3_f32 %4, %4, |%3|, |%2|\n" "v_add_f32 %3, 1.0, %2\n" "v_add_f32 %2, 1.0, %3\n" "v_min3_f32 %4, %4, |%3|, |%2|\n" "v_add_f32 %3, 1.0, %2\n" "v_add_f32 %2, 1.0, %3\n" "v_min3_f32 %4, %4, |%3|, |%2|\n" "v_add_f32 %3, 1.0, %2\n" "v_add_f32 %2, 1.0, %3\n" "v_min3_f32 %4, %4, |%3|, |%2|\n" "v_add_f32 %3, 1.0, %2\n" "v_add_f32 %2, 1.0, %3\n" "v_min3_f32 %4, %4, |%3|, |%2|\n" "v_add_f32 %3, 1.0, %2\n" "v_add_f32 %2, 1.0, %3\n" "v_min3_f32 %4, %4, |%3|, |%2|\n" "v_add_f32 %3, 1.0, %2\n" "v_add_f32 %2, 1.0, %3\n" "v_min3_f32 %4, %4, |%3|, |%2|\n" "v_add_f32 %3, 1.0, %2\n" "v_add_f32 %2, 1.0, %3\n" "v_min3_f32 %4, %4, |%3|, |%2|\n" "v_add_f32 %3, 1.0, %2\n" "v_add_f32 %2, 1.0, %3\n" "v_min3_f32 %4, %4, |%3|, |%2|\n" "v_add_f32 %3, 1.0, %2\n" "v_add_f32 %2, 1.0, %3\n" "v_min3_f32 %4, %4, |%3|, |%2|\n" "v_add_f32 %3, 1.0, %2\n" "v_add_f32 %2, 1.0, %3\n" "v_min3_f32 %4, %4, |%3|, |%2|\n" "v_add_f32 %3, 1.0, %2\n" "v_add_f32 %2, 1.0, %3\n" "v_min3_f32 %4, %4, |%3|, |%2|\n" "v_cmp_ge_f32 vcc, %7, %4\n" "v_cmp_ge_f32 vcc, %5, |%6|\n" )
// the pixel-grid sweep's first test with a lane's 64 pixels as 8 rows x 8 columns: one interval test
// per column (the distance steps by one: additions) and per row (subtractions), a three-way
// minimum per two of either: 26 instructions per 64 pairs
KERNEL1(k_raster64_block, "v_sub_f32 %2, %8, %0\n" "v_add_f32 %3, 1.0, %2\n" "v_min3_f32 %4, %4, |%2|, |%3|\n"
        "v_add_f32 %3, 2.0, %2\n" "v_add_f32 %5, 4.0, %2\n" "v_min3_f32 %4, %4, |%3|, |%5|\n"
        "v_add_f32 %3, 2.0, %2\n" "v_add_f32 %5, 4.0, %2\n" "v_min3_f32 %4, %4, |%3|, |%5|\n"
        "v_add_f32 %3, 2.0, %2\n" "v_add_f32 %5, 4.0, %2\n" "v_min3_f32 %4, %4, |%3|, |%5|\n"
        "v_sub_f32 %3, %9, %1\n" "v_sub_f32 %5, %8, %1\n" "v_min3_f32 %6, %6, |%3|, |%5|\n"
        "v_sub_f32 %3, %9, %1\n" "v_sub_f32 %5, %8, %1\n" "v_min3_f32 %6, %6, |%3|, |%5|\n"
        "v_sub_f32 %3, %9, %1\n" "v_sub_f32 %5, %8, %1\n" "v_min3_f32 %6, %6, |%3|, |%5|\n"
        "v_sub_f32 %3, %9, %1\n" "v_sub_f32 %5, %8, %1\n" "v_min3_f32 %6, %6, |%3|, |%5|\n"
        "v_cmp_ge_f32 vcc, %7, %4\n" "v_cmp_ge_f32 vcc, %7, %6\n")
// ... as 8 rows x 16 columns (128 pixels, 38 instructions) and 16 x 16 (256 pixels, 50 instructions)
KERNEL1(k_raster128_block, "v_sub_f32 %2, %8, %0\n" "v_add_f32 %3, 1.0, %2\n" "v_min3_f32 %4, %4, |%2|, |%3|\n" "v_add_f32 %3, 2.0, %2\n" "v_add_f32 %5, 4.0, %2\n" "v_min3_f32 %4, %4, |%3|, |%5|\n" "v_add_f32 %3, 2.0, %2\n" "v_add_f32 %5, 4.0, %2\n" "v_min3_f32 %4, %4, |%3|, |%5|\n" "v_add_f32 %3, 2.0, %2\n" "v_add_f32 %5, 4.0, %2\n" "v_min3_f32 %4, %4, |%3|, |%5|\n" "v_add_f32 %3, 2.0, %2\n" "v_add_f32 %5, 4.0, %2\n" "v_min3_f32 %4, %4, |%3|, |%5|\n" "v_add_f32 %3, 2.0, %2\n" "v_add_f32 %5, 4.0, %2\n" "v_min3_f32 %4, %4, |%3|, |%5|\n" "v_add_f32 %3, 2.0, %2\n" "v_add_f32 %5, 4.0, %2\n" "v_min3_f32 %4, %4, |%3|, |%5|\n" "v_add_f32 %3, 2.0, %2\n" "v_add_f32 %5, 4.0, %2\n" "v_min3_f32 %4, %4, |%3|, |%5|\n" "v_sub_f32 %3, %9, %1\n" "v_sub_f32 %5, %8, %1\n" "v_min3_f32 %6, %6, |%3|, |%5|\n" "v_sub_f32 %3, %9, %1\n" "v_sub_f32 %5, %8, %1\n" "v_min3_f32 %6, %6, |%3|, |%5|\n" "v_sub_f32 %3, %9, %1\n" "v_sub_f32 %5, %8, %1\n" "v_min3_f32 %6, %6, |%3|, |%5|\n" "v_sub_f32 %3, %9, %1\n" "v_sub_f32 %5, %8, %1\n" "v_min3_f32 %6, %6, |%3|, |%5|\n" "v_cmp_ge_f32 vcc, %7, %4\n" "v_cmp_ge_f32 vcc, %7, %6\n")
KERNEL1(k_raster256_block, "v_sub_f32 %2, %8, %0\n" "v_add_f32 %3, 1.0, %2\n" "v_min3_f32 %4, %4, |%2|, |%3|\n" "v_add_f32 %3, 2.0, %2\n" "v_add_f32 %5, 4.0, %2\n" "v_min3_f32 %4, %4, |%3|, |%5|\n" "v_add_f32 %3, 2.0, %2\n" "v_add_f32 %5, 4.0, %2\n" "v_min3_f32 %4, %4, |%3|, |%5|\n" "v_add_f32 %3, 2.0, %2\n" "v_add_f32 %5, 4.0, %2\n" "v_min3_f32 %4, %4, |%3|, |%5|\n" "v_add_f32 %3, 2.0, %2\n" "v_add_f32 %5, 4.0, %2\n" "v_min3_f32 %4, %4, |%3|, |%5|\n" "v_add_f32 %3, 2.0, %2\n" "v_add_f32 %5, 4.0, %2\n" "v_min3_f32 %4, %4, |%3|, |%5|\n" "v_add_f32 %3, 2.0, %2\n" "v_add_f32 %5, 4.0, %2\n" "v_min3_f32 %4, %4, |%3|, |%5|\n" "v_add_f32 %3, 2.0, %2\n" "v_add_f32 %5, 4.0, %2\n" "v_min3_f32 %4, %4, |%3|, |%5|\n" "v_sub_f32 %3, %9, %1\n" "v_sub_f32 %5, %8, %1\n" "v_min3_f32 %6, %6, |%3|, |%5|\n" "v_sub_f32 %3, %9, %1\n" "v_sub_f32 %5, %8, %1\n" "v_min3_f32 %6, %6, |%3|, |%5|\n" "v_sub_f32 %3, %9, %1\n" "v_sub_f32 %5, %8, %1\n" "v_min3_f32 %6, %6, |%3|, |%5|\n" "v_sub_f32 %3, %9, %1\n" "v_sub_f32 %5, %8, %1\n" "v_min3_f32 %6, %6, |%3|, |%5|\n" "v_sub_f32 %3, %9, %1\n" "v_sub_f32 %5, %8, %1\n" "v_min3_f32 %6, %6, |%3|, |%5|\n" "v_sub_f32 %3, %9, %1\n" "v_sub_f32 %5, %8, %1\n" "v_min3_f32 %6, %6, |%3|, |%5|\n" "v_sub_f32 %3, %9, %1\n" "v_sub_f32 %5, %8, %1\n" "v_min3_f32 %6, %6, |%3|, |%5|\n" "v_sub_f32 %3, %9, %1\n" "v_sub_f32 %5, %8, %1\n" "v_min3_f32 %6, %6, |%3|, |%5|\n" "v_cmp_ge_f32 vcc, %7, %4\n" "v_cmp_ge_f32 vcc, %7, %6\n")
// the any-hit sweep's first test: the u half of the bounding square, 8 rays per lane
KERNEL1(k_apex8_strip, "v_sub_f32 %2, %8, %0\n" "v_sub_f32 %3, %8, %1\n" "v_min3_f32 %4, %4, |%2|, |%3|\n" "v_sub_f32 %2, %8, %0\n" "v_sub_f32 %3, %8, %1\n" "v_min3_f32 %4, %4, |%2|, |%3|\n" "v_sub_f32 %2, %8, %0\n" "v_sub_f32 %3, %8, %1\n" "v_min3_f32 %4, %4, |%2|, |%3|\n" "v_sub_f32 %2, %8, %0\n" "v_sub_f32 %3, %8, %1\n" "v_min3_f32 %4, %4, |%2|, |%3|\n" "v_cmp_ge_f32 vcc, %7, %4\n" )
// ... and 16 rays per lane
KERNEL1(k_apex16_strip, "v_sub_f32 %2, %8, %0\n" "v_sub_f32 %3, %8, %1\n" "v_min3_f32 %4, %4, |%2|, |%3|\n" "v_sub_f32 %2, %8, %0\n" "v_sub_f32 %3, %8, %1\n" "v_min3_f32 %4, %4, |%2|, |%3|\n" "v_sub_f32 %2, %8, %0\n" "v_sub_f32 %3, %8, %1\n" "v_min3_f32 %4, %4, |%2|, |%3|\n" "v_sub_f32 %2, %8, %0\n" "v_sub_f32 %3, %8, %1\n" "v_min3_f32 %4, %4, |%2|, |%3|\n" "v_sub_f32 %2, %8, %0\n" "v_sub_f32 %3, %8, %1\n" "v_min3_f32 %4, %4, |%2|, |%3|\n" "v_sub_f32 %2, %8, %0\n" "v_sub_f32 %3, %8, %1\n" "v_min3_f32 %4, %4, |%2|, |%3|\n" "v_sub_f32 %2, %8, %0\n" "v_sub_f32 %3, %8, %1\n" "v_min3_f32 %4, %4, |%2|, |%3|\n" "v_sub_f32 %2, %8, %0\n" "v_sub_f32 %3, %8, %1\n" "v_min3_f32 %4, %4, |%2|, |%3|\n" "v_cmp_ge_f32 vcc, %7, %4\n" )
template <class K>
void run_mix(const char* name, K k, float* out, int wgs, int instrs, int pairs)
{
    const int iters = 40000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, 0, out, iters / 4, 1.0001f, 0.5f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double waves = (double)wgs * 4, s = ms * 1e-3;
    printf("%-16s %8.3f ms  %d instructions per %d pairs: %.2f cycles per wave-instr at 2.4 GHz, %.3e pairs/s\n",
           name, ms, instrs, pairs, 2.4e9 * 1024 / (waves * iters * instrs / s), waves * iters * 64.0 * pairs / s);
}

template <class K>
void run(const char* name, K k, float* out, int wgs)
{
    const int iters = 40000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, 0, out, iters / 4, 1.0001f, 0.5f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double instr = (double)wgs * 4 * iters * 64;     // wave instructions
    printf("%-16s %8.3f ms  %.3e wave-instr/s  cycles per wave-instr per SIMD at 2.4 GHz: %.2f\n", name, ms,
           instr / (ms * 1e-3), 2.4e9 * 1024 / (instr / (ms * 1e-3)));
}

int main()
{
    float* out;
    const int wgs = 256 * 8;    // 8 workgroups = 32 waves per CU = 8 per SIMD
    (void)hipMalloc(&out, sizeof(float) * 256 * wgs);
    run("fma", k_fma, out, wgs);
    run("fma", k_fma, out, wgs);
    run("mul", k_mul, out, wgs);
    run("add", k_add, out, wgs);
    run("mac", k_mac, out, wgs);
    run("and", k_and, out, wgs);
    run("or3", k_or3, out, wgs);
    run("and_or", k_and_or, out, wgs);
    run("max3", k_max3, out, wgs);
    run("max", k_max, out, wgs);
    run("min", k_min, out, wgs);
    run("max_i32", k_max_i32, out, wgs);
    run("max3_i32", k_max3_i32, out, wgs);
    run("add_u32", k_add_u32, out, wgs);
    run("cndmask", k_cndmask, out, wgs);
    run("cmp", k_cmp, out, wgs);
    run("mix_2fma_andor", k_mix_2fma_andor, out, wgs);
    run("mix_6fma_max3", k_mix_6fma_max3, out, wgs);
    run("mix_fma_and", k_mix_fma_and, out, wgs);
    run("mix_fma_add", k_mix_fma_add, out, wgs);
    run("pk_add", k_pk_add, out, wgs);
    run("pk_fma", k_pk_fma, out, wgs);
    run("pk_mul", k_pk_mul, out, wgs);
    run("add", k_add, out, wgs);
    run_mix("raster16", k_raster16, out, wgs, 51, 16);
    run_mix("raster32", k_raster32, out, wgs, 99, 32);
    run_mix("raster64", k_raster64, out, wgs, 195, 64);
    run_mix("apex8", k_apex8, out, wgs, 45, 8);
    run_mix("apex8_sign", k_apex8_sign, out, wgs, 41, 8);
    run_mix("raster64_rowbox", k_raster64_box, out, wgs, 99, 64);
    run_mix("raster64_box", k_raster64_block, out, wgs, 26, 64);
    run_mix("raster128_box", k_raster128_block, out, wgs, 38, 128);
    run_mix("raster256_box", k_raster256_block, out, wgs, 50, 256);
    run_mix("apex8_box", k_apex8_box, out, wgs, 29, 8);
    run_mix("apex8_strip", k_apex8_strip, out, wgs, 13, 8);
    run_mix("apex16_strip", k_apex16_strip, out, wgs, 25, 16);
    run_mix("apex8_circle", k_apex8_circle, out, wgs, 37, 8);
    (void)hipFree(out);
    return 0;
}
