// rt_transform.hip -- per-frame vertex transform and construction of the fp32 sweep buffer.
//
// k_transform is mat_mult_points_kernel (mat_mult.cu:8-28): rows 0..2 of a row-major 4x4
// applied to structure-of-arrays fp64 points, sums left to right, one point per lane,
// fully coalesced (24 B in, 24 B out per point).  It is compiled without FMA contraction so
// the result equals the host twin M3d_mat_mult_points (M3d_matrix_tools.c:184-208) bit for bit.
//
// k_pack derives, per triangle, what the sweep consumes -- v0 = A, e1 = B-A, e2 = C-A and,
// depending on the formulation, nn = e2 x e1, g = e2 x v0, h = e1 x v0, k = v0.nn (everything
// formed in fp64, then rounded once to fp32) -- plus two rounding bounds that make the fp32
// test a conservative filter (see rt_sweep.hip):
//     eu bounds |error of U*, V*, det-U-V| for a unit-length ray direction,
//     et bounds |error of T|,
// both proportional to 2^-20 * (4 * largest |coordinate| in eye space) * edge lengths.
#include "rt_internal.h"

namespace {

__global__ void __launch_bounds__(RT_WG)
k_transform(const double* __restrict__ m, const double* x, const double* y, const double* z,
            double* X, double* Y, double* Z, int n, unsigned* __restrict__ max_bits)
{
    const int i = blockIdx.x * RT_WG + threadIdx.x;
    float mx = 0.0f;
    if (i < n) {
        const double xi = x[i], yi = y[i], zi = z[i];
        const double u = m[0] * xi + m[1] * yi + m[2] * zi + m[3];      // mat_mult.cu:20-22
        const double v = m[4] * xi + m[5] * yi + m[6] * zi + m[7];
        const double w = m[8] * xi + m[9] * yi + m[10] * zi + m[11];
        X[i] = u; Y[i] = v; Z[i] = w;
        if (i > 0) {        // slot 0 of the 1-indexed arrays is never a real vertex
            const double a = fmax(fabs(u), fmax(fabs(v), fabs(w)));
            mx = (a == a && a < 3.0e38) ? (float)a : 0.0f;
        }
    }
    if (max_bits) {
        // non-negative floats order like their bit patterns
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
        if ((threadIdx.x & 63) == 0 && mx > 0.0f) atomicMax(max_bits, __float_as_uint(mx));
    }
}

__global__ void __launch_bounds__(RT_WG)
k_pack(const double* __restrict__ x, const double* __restrict__ y, const double* __restrict__ z,
       const int* __restrict__ iA, const int* __restrict__ iB, const int* __restrict__ iC,
       int num_tris, int num_tris_padded, const unsigned* __restrict__ max_bits,
       float4* __restrict__ planes)
{
    const int i = blockIdx.x * RT_WG + threadIdx.x;
    if (i >= num_tris_padded) return;
    float4* p[RT_PLANES];
#pragma unroll
    for (int k = 0; k < RT_PLANES; ++k) p[k] = planes + (long)k * num_tris_padded + i;
    if (i >= num_tris) {        // padding: a null triangle can never pass the filter
#pragma unroll
        for (int k = 0; k < RT_PLANES; ++k) *p[k] = make_float4(0, 0, 0, 0);
        return;
    }
    const int a = iA[i], b = iB[i], c = iC[i];
    const double ax = x[a], ay = y[a], az = z[a];
    const double e1x = x[b] - ax, e1y = y[b] - ay, e1z = z[b] - az;
    const double e2x = x[c] - ax, e2y = y[c] - ay, e2z = z[c] - az;
    const double l1 = sqrt(e1x * e1x + e1y * e1y + e1z * e1z);
    const double l2 = sqrt(e2x * e2x + e2y * e2y + e2z * e2z);
    const double scale = 4.0 * (double)__uint_as_float(*max_bits) * 9.5367431640625e-07;  // 2^-20
    const float eu = (float)(3.0 * scale * fmax(l1, l2));
    const float eu2 = (float)(6.0 * scale * fmax(l1, l2));
    const float et = (float)(scale * l1 * l2);
    // nn = e2 x e1, so that det = e1.(D x e2) = D.nn
    const double nx = e2y * e1z - e2z * e1y, ny = e2z * e1x - e2x * e1z, nz = e2x * e1y - e2y * e1x;
#if RT_SWEEP_FORM == 0
    *p[0] = make_float4((float)ax, (float)ay, (float)az, eu);
    *p[1] = make_float4((float)e1x, (float)e1y, (float)e1z, et);
    *p[2] = make_float4((float)e2x, (float)e2y, (float)e2z, 0.0f);
    (void)eu2; (void)nx; (void)ny; (void)nz;
#elif RT_SWEEP_FORM == 1
    *p[0] = make_float4((float)ax, (float)ay, (float)az, eu);
    *p[1] = make_float4((float)e2x, (float)e2y, (float)e2z, eu2);
    *p[2] = make_float4((float)nx, (float)ny, (float)nz, 0.0f);
    *p[3] = make_float4((float)e1x, (float)e1y, (float)e1z, et);
#else
    // moment form: g = e2 x v0, h = e1 x v0, k = v0.nn -- all formed in fp64, rounded once
    const double gx = e2y * az - e2z * ay, gy = e2z * ax - e2x * az, gz = e2x * ay - e2y * ax;
    const double hx = e1y * az - e1z * ay, hy = e1z * ax - e1x * az, hz = e1x * ay - e1y * ax;
    const double k = ax * nx + ay * ny + az * nz;
    *p[0] = make_float4((float)nx, (float)ny, (float)nz, eu2);
    *p[1] = make_float4((float)gx, (float)gy, (float)gz, eu);
    *p[2] = make_float4((float)e2x, (float)e2y, (float)e2z, et);
    *p[3] = make_float4((float)e1x, (float)e1y, (float)e1z, (float)k);
    *p[4] = make_float4((float)hx, (float)hy, (float)hz, 0.0f);
#endif
}

}  // namespace

int rt_launch_transform(const double* m16, const double* x, const double* y, const double* z,
                        double* X, double* Y, double* Z, int n, unsigned* max_bits, hipStream_t st)
{
    if (n <= 0) return RT_OK;
    hipLaunchKernelGGL(k_transform, dim3((n + RT_WG - 1) / RT_WG), dim3(RT_WG), 0, st, m16, x, y, z,
                       X, Y, Z, n, max_bits);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        rt_set_error("transform launch failed: %s", hipGetErrorString(e));
        return RT_E_HIP;
    }
    return RT_OK;
}

int rt_launch_pack(rt_scene* sc, hipStream_t st)
{
    const int np = sc->num_tris_padded;
    hipLaunchKernelGGL(k_pack, dim3((np + RT_WG - 1) / RT_WG), dim3(RT_WG), 0, st,
                       sc->ex.as<double>(), sc->ey.as<double>(), sc->ez.as<double>(),
                       sc->iA.as<int>(), sc->iB.as<int>(), sc->iC.as<int>(), sc->num_tris, np,
                       sc->scratch_max.as<unsigned>(), sc->planes.as<float4>());
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        rt_set_error("pack launch failed: %s", hipGetErrorString(e));
        return RT_E_HIP;
    }
    return RT_OK;
}
