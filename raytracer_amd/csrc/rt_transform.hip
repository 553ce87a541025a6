// rt_transform.hip -- per-frame vertex transform and construction of the fp32 sweep buffer.
//
// k_transform is mat_mult_points_kernel (mat_mult.cu:8-28): rows 0..2 of a row-major 4x4
// applied to structure-of-arrays fp64 points, sums left to right, one point per lane,
// fully coalesced (24 B in, 24 B out per point).  It is compiled without FMA contraction so
// the result equals the host twin M3d_mat_mult_points (M3d_matrix_tools.c:184-208) bit for bit.
//
// k_pack derives, per triangle, what the sweep consumes: v0 = A, e1 = B-A, e2 = C-A
// (differences taken in fp64, then rounded once to fp32) plus two rounding bounds that make
// the fp32 test a conservative filter (see rt_sweep.hip):
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
       float4* __restrict__ v0, float4* __restrict__ e1, float4* __restrict__ e2,
       float4* __restrict__ nn)
{
    const int i = blockIdx.x * RT_WG + threadIdx.x;
    if (i >= num_tris_padded) return;
    if (i >= num_tris) {        // padding: a null triangle can never pass the filter
        v0[i] = make_float4(0, 0, 0, 0);
        e1[i] = make_float4(0, 0, 0, 0);
        e2[i] = make_float4(0, 0, 0, 0);
        nn[i] = make_float4(0, 0, 0, 0);
        return;
    }
    const int a = iA[i], b = iB[i], c = iC[i];
    const double ax = x[a], ay = y[a], az = z[a];
    const double e1x = x[b] - ax, e1y = y[b] - ay, e1z = z[b] - az;
    const double e2x = x[c] - ax, e2y = y[c] - ay, e2z = z[c] - az;
    const double l1 = sqrt(e1x * e1x + e1y * e1y + e1z * e1z);
    const double l2 = sqrt(e2x * e2x + e2y * e2y + e2z * e2z);
    const double scale = 4.0 * (double)__uint_as_float(*max_bits) * 9.5367431640625e-07;  // 2^-20
    const double eu = 3.0 * scale * fmax(l1, l2);
    const double et = scale * l1 * l2;
    v0[i] = make_float4((float)ax, (float)ay, (float)az, (float)eu);
    e1[i] = make_float4((float)e1x, (float)e1y, (float)e1z, (float)et);
    e2[i] = make_float4((float)e2x, (float)e2y, (float)e2z, (float)(2.0 * eu));
    // nn = e2 x e1, so that det = e1.(D x e2) = D.nn
    nn[i] = make_float4((float)(e2y * e1z - e2z * e1y), (float)(e2z * e1x - e2x * e1z),
                        (float)(e2x * e1y - e2y * e1x), 0.0f);
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
                       sc->scratch_max.as<unsigned>(), sc->pv0.as<float4>(), sc->pe1.as<float4>(),
                       sc->pe2.as<float4>(), sc->pnn.as<float4>());
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        rt_set_error("pack launch failed: %s", hipGetErrorString(e));
        return RT_E_HIP;
    }
    return RT_OK;
}
