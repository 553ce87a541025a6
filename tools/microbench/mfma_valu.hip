// Do the matrix cores and the vector ALU of a SIMD run side by side on gfx950?  (development aid)
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu mfma_valu.hip
// Three kernels over the same grid (8 waves per SIMD): FMA only, MFMA only (fp32 16x16x4), and
// both interleaved in every wave.  If the pipes overlap, "both" takes about max(fma, mfma), not
// their sum.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE0>    // 1 = FMA, 2 = MFMA, 3 = both in every wave, 4 = odd waves MFMA, even waves FMA
__global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b)
{
    const int MODE = MODE0 == 4 ? ((threadIdx.x >> 6) & 1 ? 2 : 1) : MODE0;
    float v[8];
    f4 m[4];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x + i;
    for (int i = 0; i < 4; ++i) m[i] = (f4){a, b, a, b};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
            if (MODE & 2) {
#pragma unroll
                for (int i = 0; i < 4; ++i) m[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, m[i], 0, 0, 0);
            }
            if (MODE & 1) {
#pragma unroll
                for (int r2 = 0; r2 < 4; ++r2)
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], a, b);
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i];
    for (int i = 0; i < 4; ++i) s += m[i].x + m[i].y + m[i].z + m[i].w;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE0>
float run(float* out, int wgs, int iters)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE0>, dim3(wgs), dim3(256), 0, 0, out, iters / 4, 1.0001f, 0.5f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE0>, dim3(wgs), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    float* out;
    const int wgs = 256 * 8, iters = 4000;
    (void)hipMalloc(&out, sizeof(float) * 256 * wgs);
    const float f = run<1>(out, wgs, iters), m = run<2>(out, wgs, iters), both = run<3>(out, wgs, iters);
    // per iteration and wave: 128 v_fma (64 lanes x 2 flop), 16 MFMA (16x16x4: 2048 flop each)
    const double waves = wgs * 4.0;
    printf("fma only   %8.3f ms  %.1f TFLOP/s\n", f, waves * iters * 128 * 128 / (f * 1e-3) / 1e12);
    printf("mfma only  %8.3f ms  %.1f TFLOP/s\n", m, waves * iters * 16 * 2048.0 / (m * 1e-3) / 1e12);
    printf("both       %8.3f ms  (sum %.3f, max %.3f)\n", both, f + m, f > m ? f : m);
    const float split = run<4>(out, wgs, iters);
    printf("split      %8.3f ms  half the waves each: (sum/2 %.3f, max/2 %.3f)\n", split, (f + m) / 2, (f > m ? f : m) / 2);
    (void)hipFree(out);
    return 0;
}
