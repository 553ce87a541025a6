// Issue rate of packed 16-bit integer VALU instructions on gfx950, and of the first tests of the
// two apex sweeps written with them (development aid): 8 waves per SIMD, no LDS reads, no branch.
//   hipcc --offload-arch=gfx950 -O3 -o pk16_rate pk16_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define R2(x) x x
#define R4(x) R2(x) R2(x)
#define R8(x) R4(x) R4(x)
#define R16(x) R8(x) R8(x)
#define R31(x) R16(x) R8(x) R4(x) R2(x) x

#define KERNEL(name, body)                                                                        \
    __global__ void __launch_bounds__(256) name(unsigned* out, int iters, unsigned a, unsigned b) \
    {                                                                                             \
        unsigned v0 = threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4,            \
                 v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7;                                           \
        const unsigned c = __builtin_amdgcn_readfirstlane(a * 3 + 1);                             \
        for (int i = 0; i < iters; ++i)                                                           \
            asm volatile(body                                                                     \
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6),  \
                           "+v"(v7)                                                               \
                         : "v"(a), "v"(b), "s"(c) : "vcc");                                       \
        out[blockIdx.x * 256 + threadIdx.x] = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;              \
    }
#define EIGHT(op, tail) op " %0, %0, " tail "\n" op " %1, %1, " tail "\n" op " %2, %2, " tail "\n" op " %3, %3, " tail "\n" \
                        op " %4, %4, " tail "\n" op " %5, %5, " tail "\n" op " %6, %6, " tail "\n" op " %7, %7, " tail "\n"
KERNEL(k_add_f32, R8(EIGHT("v_add_f32", "%8")))
KERNEL(k_pk_add_u16, R8(EIGHT("v_pk_add_u16", "%8")))
KERNEL(k_pk_add_u16_s, R8(EIGHT("v_pk_add_u16", "%10")))
KERNEL(k_pk_sub_i16, R8(EIGHT("v_pk_sub_i16", "%8")))
KERNEL(k_pk_min_u16, R8(EIGHT("v_pk_min_u16", "%8")))
KERNEL(k_pk_max_i16, R8(EIGHT("v_pk_max_i16", "%8")))
KERNEL(k_pk_add_f16, R8(EIGHT("v_pk_add_f16", "%8")))
KERNEL(k_pk_min_f16, R8(EIGHT("v_pk_min_f16", "%8")))
KERNEL(k_min_u32, R8(EIGHT("v_min_u32", "%8")))
KERNEL(k_min3_u32, R8(EIGHT("v_min3_u32", "%8, %9")))
// alternating, two independent accumulators
KERNEL(k_mix_add_min, R8("v_pk_add_u16 %2, %0, %10\n" "v_pk_min_u16 %4, %4, %2\n" "v_pk_add_u16 %3, %0, %10\n" "v_pk_min_u16 %5, %5, %3\n"
                         "v_pk_add_u16 %2, %1, %10\n" "v_pk_min_u16 %4, %4, %2\n" "v_pk_add_u16 %3, %1, %10\n" "v_pk_min_u16 %5, %5, %3\n"))

// First test of the pixel-grid sweep on 16-bit pixel coordinates, 64 pixels of a row per lane, two
// to a register: a = (x0 - lo) in both halves (1), then for every pair of pixels its offset added
// (a wave-uniform constant) and an unsigned minimum kept (64 - 1), the halves' minima combined
// (1), compared with the box's width (1), the row against the box in y (2): 68 per 64 pairs.
#define PAIR "v_pk_add_u16 %2, %0, %10\n" "v_pk_min_u16 %4, %4, %2\n"
KERNEL(k_raster64_pk16, "v_pk_sub_u16 %0, %8, %1\n" "v_pk_add_u16 %4, %0, %10\n" R31(PAIR)
                        "v_pk_min_u16 %4, %4, %4 op_sel:[0,1] op_sel_hi:[1,0]\n" "v_cmp_ge_u32 vcc, %7, %4\n"
                        "v_sub_f32 %6, %9, %1\n" "v_cmp_ge_f32 vcc, %5, |%6|\n")
// the same with two accumulators (shorter dependency chains)
#define PAIR2 "v_pk_add_u16 %2, %0, %10\n" "v_pk_min_u16 %4, %4, %2\n" "v_pk_add_u16 %3, %0, %10\n" "v_pk_min_u16 %6, %6, %3\n"
KERNEL(k_raster64_pk16_2acc, "v_pk_sub_u16 %0, %8, %1\n" "v_pk_add_u16 %4, %0, %10\n" "v_pk_add_u16 %6, %0, %10\n" R8(PAIR2) R4(PAIR2) R2(PAIR2) PAIR2
                        "v_pk_min_u16 %4, %4, %6\n"
                        "v_pk_min_u16 %4, %4, %4 op_sel:[0,1] op_sel_hi:[1,0]\n" "v_cmp_ge_u32 vcc, %7, %4\n"
                        "v_sub_f32 %3, %9, %1\n" "v_cmp_ge_f32 vcc, %5, |%3|\n")
// First test of the any-hit sweep on 16-bit fixed-point u, 16 rays per lane, two to a register:
// 8 subtractions (u - lo, unsigned wrap), 7 minima, halves combined, compared: 17 per 16 pairs.
KERNEL(k_apex16_pk16, "v_pk_sub_u16 %2, %0, %8\n" "v_pk_sub_u16 %3, %1, %8\n" "v_pk_min_u16 %4, %2, %3\n"
                      R2("v_pk_sub_u16 %2, %5, %8\n" "v_pk_min_u16 %4, %4, %2\n" "v_pk_sub_u16 %3, %6, %8\n" "v_pk_min_u16 %4, %4, %3\n"
                         "v_pk_sub_u16 %2, %7, %8\n" "v_pk_min_u16 %4, %4, %2\n")
                      "v_pk_min_u16 %4, %4, %4 op_sel:[0,1] op_sel_hi:[1,0]\n" "v_cmp_ge_u32 vcc, %9, %4\n")
// 8 rays per lane: 4 + 3 + 2 = 9 per 8 pairs
KERNEL(k_apex8_pk16, "v_pk_sub_u16 %2, %0, %8\n" "v_pk_sub_u16 %3, %1, %8\n" "v_pk_min_u16 %4, %2, %3\n"
                     "v_pk_sub_u16 %2, %5, %8\n" "v_pk_min_u16 %4, %4, %2\n" "v_pk_sub_u16 %3, %6, %8\n" "v_pk_min_u16 %4, %4, %3\n"
                     "v_pk_min_u16 %4, %4, %4 op_sel:[0,1] op_sel_hi:[1,0]\n" "v_cmp_ge_u32 vcc, %9, %4\n")

template <class K>
void run(const char* name, K k, unsigned* out, int wgs, int instrs, int pairs)
{
    const int iters = instrs >= 60 ? 40000 : 80000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, 0, out, iters / 4, 5u, 7u);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, 0, out, iters, 5u, 7u);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double waves = (double)wgs * 4, s = ms * 1e-3;
    printf("%-20s %8.3f ms  %3d instructions", name, ms, instrs);
    if (pairs) printf(" per %2d pairs", pairs);
    printf(": %.2f cycles per wave-instr at 2.4 GHz", 2.4e9 * 1024 / (waves * iters * instrs / s));
    if (pairs) printf(", %.3e pairs/s", waves * iters * 64.0 * pairs / s);
    printf("\n");
    fflush(stdout);
}

int main()
{
    unsigned* out;
    const int wgs = 256 * 8;    // 8 workgroups = 32 waves per CU = 8 per SIMD
    if (hipMalloc(&out, sizeof(unsigned) * 256 * wgs) != hipSuccess) return 1;
    run("add_f32", k_add_f32, out, wgs, 64, 0);
    run("add_f32", k_add_f32, out, wgs, 64, 0);
    run("pk_add_u16", k_pk_add_u16, out, wgs, 64, 0);
    run("pk_add_u16 (sgpr)", k_pk_add_u16_s, out, wgs, 64, 0);
    run("pk_sub_i16", k_pk_sub_i16, out, wgs, 64, 0);
    run("pk_min_u16", k_pk_min_u16, out, wgs, 64, 0);
    run("pk_max_i16", k_pk_max_i16, out, wgs, 64, 0);
    run("pk_add_f16", k_pk_add_f16, out, wgs, 64, 0);
    run("pk_min_f16", k_pk_min_f16, out, wgs, 64, 0);
    run("min_u32", k_min_u32, out, wgs, 64, 0);
    run("min3_u32", k_min3_u32, out, wgs, 64, 0);
    run("mix pk_add/pk_min", k_mix_add_min, out, wgs, 64, 0);
    run("raster64_pk16", k_raster64_pk16, out, wgs, 68, 64);
    run("raster64_pk16_2acc", k_raster64_pk16_2acc, out, wgs, 70, 64);
    run("apex16_pk16", k_apex16_pk16, out, wgs, 17, 16);
    run("apex8_pk16", k_apex8_pk16, out, wgs, 9, 8);
    (void)hipFree(out);
    return 0;
}
