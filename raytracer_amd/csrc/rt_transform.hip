// rt_transform.hip -- per-frame vertex transform and construction of the fp32 sweep buffer.
//
// k_transform is mat_mult_points_kernel (mat_mult.cu:8-28): rows 0..2 of a row-major 4x4
// applied to structure-of-arrays fp64 points, sums left to right, one point per lane,
// fully coalesced (24 B in, 24 B out per point).  It is compiled without FMA contraction so
// the result equals the host twin M3d_mat_mult_points (M3d_matrix_tools.c:184-208) bit for bit.
//
// k_pack derives, per triangle, what the general-position sweep consumes -- with v0 = A,
// e1 = B-A, e2 = C-A: nn = e2 x e1, g = e2 x v0, h = e1 x v0, k = v0.nn (everything formed in
// fp64, then rounded once to fp32) -- plus two rounding bounds that make the fp32 test a
// conservative filter (see rt_sweep.hip):
//     eu bounds |error of U*, V*, det-U-V| for a unit-length ray direction,
//     et bounds |error of T|,
// both proportional to 2^-20 * (4 * largest |coordinate| in eye space) * edge lengths.
#include "rt_internal.h"

namespace {

constexpr int kTransformPoints = 8;     // points per lane of k_transform

// M: the matrix, rows 0..2 -- in device memory (const double*) or passed with the launch (RtMat12:
// no upload, nothing of the caller's the device still reads when the launch call returns)
template <class M>
__global__ void __launch_bounds__(RT_WG)
k_transform(const M m, const double* x, const double* y, const double* z,
            double* X, double* Y, double* Z, int n, unsigned* __restrict__ max_bits)
{
    // kTransformPoints points per lane, so that the largest coordinate costs one atomic per
    // workgroup and a few thousand per launch (one per wave and point made three million vertices
    // wait 0.5 ms for 47 000 atomics on one address)
    float mx = 0.0f;
#pragma unroll
    for (int k = 0; k < kTransformPoints; ++k) {
        const int i = (blockIdx.x * kTransformPoints + k) * RT_WG + threadIdx.x;
        if (i < n) {
            const double xi = x[i], yi = y[i], zi = z[i];
            const double u = m[0] * xi + m[1] * yi + m[2] * zi + m[3];      // mat_mult.cu:20-22
            const double v = m[4] * xi + m[5] * yi + m[6] * zi + m[7];
            const double w = m[8] * xi + m[9] * yi + m[10] * zi + m[11];
            X[i] = u; Y[i] = v; Z[i] = w;
            if (i > 0) {        // slot 0 of the 1-indexed arrays is never a real vertex
                const double a = fmax(fabs(u), fmax(fabs(v), fabs(w)));
                mx = fmaxf(mx, (a == a && a < 3.0e38) ? (float)a : 0.0f);
            }
        }
    }
    if (max_bits) {
        // non-negative floats order like their bit patterns
        __shared__ float wave_max[RT_WG / 64];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
        if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = mx;
        __syncthreads();
        if (threadIdx.x == 0) {
#pragma unroll
            for (int w = 1; w < RT_WG / 64; ++w) mx = fmaxf(mx, wave_max[w]);
            if (mx > 0.0f) atomicMax(max_bits, __float_as_uint(mx));
        }
    }
}

__global__ void __launch_bounds__(RT_WG)
k_pack(const double* __restrict__ x, const double* __restrict__ y, const double* __restrict__ z,
       const int* __restrict__ iA, const int* __restrict__ iB, const int* __restrict__ iC,
       int num_tris, int num_tris_padded, const unsigned* __restrict__ max_bits,
       float4* __restrict__ planes, double* __restrict__ tri_verts)
{
    // planes (or nullptr): the general sweep's planes; tri_verts (or nullptr): the exact pass's copy
    const int i = blockIdx.x * RT_WG + threadIdx.x;
    if (i >= num_tris_padded) return;
    float4* p[RT_PLANES];
#pragma unroll
    for (int k = 0; k < RT_PLANES; ++k) p[k] = planes + (long)k * num_tris_padded + i;
    if (i >= num_tris) {        // padding: a null triangle can never pass the filter
        if (planes) {
#pragma unroll
            for (int k = 0; k < RT_PLANES; ++k) *p[k] = make_float4(0, 0, 0, 0);
        }
        return;
    }
    const int a = iA[i], b = iB[i], c = iC[i];
    const double ax = x[a], ay = y[a], az = z[a];
    const double bx = x[b], by = y[b], bz = z[b];
    const double cx = x[c], cy = y[c], cz = z[c];
    if (tri_verts) {       // the exact pass's copy of the triangle
        double* t = tri_verts + 9 * (long)i;
        t[0] = ax; t[1] = ay; t[2] = az;
        t[3] = bx; t[4] = by; t[5] = bz;
        t[6] = cx; t[7] = cy; t[8] = cz;
    }
    if (!planes) return;
    const double e1x = bx - ax, e1y = by - ay, e1z = bz - az;
    const double e2x = cx - ax, e2y = cy - ay, e2z = cz - az;
    const double l1 = sqrt(e1x * e1x + e1y * e1y + e1z * e1z);
    const double l2 = sqrt(e2x * e2x + e2y * e2y + e2z * e2z);
    const double scale = 4.0 * (double)__uint_as_float(*max_bits) * 9.5367431640625e-07;  // 2^-20
    const float eu = (float)(3.0 * scale * fmax(l1, l2));
    const float eu2 = (float)(6.0 * scale * fmax(l1, l2));
    const float et = (float)(scale * l1 * l2);
    // nn = e2 x e1, so that det = e1.(D x e2) = D.nn
    const double nx = e2y * e1z - e2z * e1y, ny = e2z * e1x - e2x * e1z, nz = e2x * e1y - e2y * e1x;
    // moment form: g = e2 x v0, h = e1 x v0, k = v0.nn -- all formed in fp64, rounded once
    const double gx = e2y * az - e2z * ay, gy = e2z * ax - e2x * az, gz = e2x * ay - e2y * ax;
    const double hx = e1y * az - e1z * ay, hy = e1z * ax - e1x * az, hz = e1x * ay - e1y * ax;
    const double k = ax * nx + ay * ny + az * nz;
    *p[0] = make_float4((float)nx, (float)ny, (float)nz, eu2);
    *p[1] = make_float4((float)-gx, (float)-gy, (float)-gz, eu);       // (-g: see in_u_slab, rt_sweep.hip)
    *p[2] = make_float4((float)e2x, (float)e2y, (float)e2z, et);
    *p[3] = make_float4((float)e1x, (float)e1y, (float)e1z, (float)k);
    *p[4] = make_float4((float)hx, (float)hy, (float)hz, 0.0f);
}

// Planes for sweeps of rays that all pass through one point P (the apex): the eye for primary
// rays, a light for its shadow rays.  With a_i = v_i - P, e1 = a1 - a0, e2 = a2 - a0 and any
// direction D of such a ray, the determinants of triangle_intersection.cu:59-94 reduce to dot
// products with four per-triangle vectors (the tvec x D term of the general case vanishes):
//     U  = D.p,   p  = a0 x e2        (u  = U / det)
//     W' = D.q,   q  = e2 x a1        (W' = det - U, so u <= 1  <=>  W'/det >= 0)
//     V  = D.h,   h  = e1 x a0        (v  = V / det)
//     det = D.nn, nn = e2 x e1,  and the ray parameter from P is k / det with k = a0.nn.
// Stored per triangle (z components multiplied by zs, see the launcher):
//     plane 0 = (p, c)   plane 1 = (q, m)   plane 2 = (h, k)   plane 3 = (nn, en)
// Rounding bounds, with delta = 2^-20 * dmax * rho * l  (rho = distance of the farthest vertex
// from P, l = longer edge; the true bound on the fp32 error of U, W', V is about 5 * 2^-24 of
// that product):
//     c  = delta * (dmax*|nn| + delta): a pair inside the u slab has U, W' of det's sign and
//          |U| + |W'| = |det|, hence computed U~ * W~' >= -c.  (The slab product was the any-hit
//          sweep's first test before the bounding squares below; the sweeps no longer read c.)
//     m  = 32 * delta: margin of the full inside test (per unit of |det|).
//     en = 2^-20 * dmax * |nn|: bound on the error of det.
// A triangle with an exactly zero normal (repeated or proportional vertices) has a zero
// determinant in the reference for every ray and can never be hit: it is stored as a null
// entry (c = -1).  A nearly degenerate one gets m = +inf, which sends every pair inside its
// slab to the exact pass.
#ifndef RT_PACK_ROTATE
#define RT_PACK_ROTATE 1
#endif
__global__ void __launch_bounds__(RT_WG)
k_pack_apex(const double* __restrict__ x, const double* __restrict__ y, const double* __restrict__ z,
            const int* __restrict__ iA, const int* __restrict__ iB, const int* __restrict__ iC,
            int num_tris, int num_tris_padded, const unsigned* __restrict__ max_bits,
            double px, double py, double pz, double zs, double dmax, int sign_layout,
            const int32_t* __restrict__ order, const double* __restrict__ tri_verts,
            float4* __restrict__ planes, float4* __restrict__ squares)
{
    const int i = blockIdx.x * RT_WG + threadIdx.x;      // entry of the plane arrays
    if (i >= num_tris_padded) return;
    const int src = order ? order[i] : (i < num_tris ? i : -1);     // the triangle it holds
    float4* p[RT_APEX_PLANES];
#pragma unroll
    for (int k = 0; k < RT_APEX_PLANES; ++k) p[k] = planes + (long)k * num_tris_padded + i;
    const float4 null_entry = make_float4(0, 0, 0, -1.0f);
    bool null_tri = src < 0 || src >= num_tris;
    double a0x = 0, a0y = 0, a0z = 0, a1x = 0, a1y = 0, a1z = 0, a2x = 0, a2y = 0, a2z = 0;
    if (!null_tri) {
        if (tri_verts) {
            // the triangle's vertices from the copy k_pack keeps for the exact pass: 72 bytes in one
            // place (with a scattered order, three indices and nine coordinates fetched through
            // them are twelve cache lines)
            const double* t = tri_verts + 9 * (long)src;
            a0x = t[0] - px; a0y = t[1] - py; a0z = t[2] - pz;
            a1x = t[3] - px; a1y = t[4] - py; a1z = t[5] - pz;
            a2x = t[6] - px; a2y = t[7] - py; a2z = t[8] - pz;
        } else {
            const int a = iA[src], b = iB[src], c = iC[src];
            a0x = x[a] - px; a0y = y[a] - py; a0z = z[a] - pz;
            a1x = x[b] - px; a1y = y[b] - py; a1z = z[b] - pz;
            a2x = x[c] - px; a2y = y[c] - py; a2z = z[c] - pz;
        }
#if RT_PACK_ROTATE
        // The sweep's first test is the u slab: the strip between the line through a0 and a2 and
        // its parallel through a1.  Which vertex is called a0 does not matter to the filter
        // (the inside test and its bounds are symmetric under a rotation of the labels, and the
        // exact pass works from the reference's own order), so the labels are rotated to make
        // that strip the narrowest of the three: the edge a0-a2 becomes the one that looks
        // longest from P, and fewer rays lie in the slab.
        const double r0s = a0x * a0x + a0y * a0y + a0z * a0z;
        const double r1s = a1x * a1x + a1y * a1y + a1z * a1z;
        const double r2s = a2x * a2x + a2y * a2y + a2z * a2z;
        auto sin2 = [](double ux, double uy, double uz, double vx, double vy, double vz, double us, double vs) {
            const double cx = uy * vz - uz * vy, cy = uz * vx - ux * vz, cz = ux * vy - uy * vx;
            return (cx * cx + cy * cy + cz * cz) / (us * vs);
        };
        if (r0s > 0.0 && r1s > 0.0 && r2s > 0.0) {
            const double s02 = sin2(a0x, a0y, a0z, a2x, a2y, a2z, r0s, r2s);
            const double s01 = sin2(a0x, a0y, a0z, a1x, a1y, a1z, r0s, r1s);
            const double s12 = sin2(a1x, a1y, a1z, a2x, a2y, a2z, r1s, r2s);
            if (s01 > s02 && s01 >= s12) {          // (a0, a1, a2) <- (a1, a2, a0)
                const double tx = a0x, ty = a0y, tz = a0z;
                a0x = a1x; a0y = a1y; a0z = a1z;
                a1x = a2x; a1y = a2y; a1z = a2z;
                a2x = tx; a2y = ty; a2z = tz;
            } else if (s12 > s02 && s12 > s01) {    // (a0, a1, a2) <- (a2, a0, a1)
                const double tx = a2x, ty = a2y, tz = a2z;
                a2x = a1x; a2y = a1y; a2z = a1z;
                a1x = a0x; a1y = a0y; a1z = a0z;
                a0x = tx; a0y = ty; a0z = tz;
            }
        }
#endif
    }
    const double e1x = a1x - a0x, e1y = a1y - a0y, e1z = a1z - a0z;
    const double e2x = a2x - a0x, e2y = a2y - a0y, e2z = a2z - a0z;
    const double nx = e2y * e1z - e2z * e1y, ny = e2z * e1x - e2x * e1z, nz = e2x * e1y - e2y * e1x;
    const double l1 = sqrt(e1x * e1x + e1y * e1y + e1z * e1z);
    const double l2 = sqrt(e2x * e2x + e2y * e2y + e2z * e2z);
    const double nn = sqrt(nx * nx + ny * ny + nz * nz);
    if (!(nn > 0.0) || !(l1 > 0.0) || !(l2 > 0.0)) null_tri = true;      // also catches NaN
    if (null_tri) {     // never passes: c = -1 in the product layout, positive constants in the sign layout
#pragma unroll
        for (int k = 0; k < RT_APEX_PLANES; ++k) *p[k] = sign_layout ? make_float4(0, 0, 1.0f, 0) : null_entry;
        if (squares && sign_layout) squares[i] = make_float4(0, -1.0f, 0, -1.0f);
        if (squares && !sign_layout) {
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) squares[(long)ax * num_tris_padded + i] = make_float4(0, 0, -1.0f, 0);
        }
        return;
    }
    // p = a0 x e2, q = e2 x a1, h = e1 x a0, k = a0.nn -- formed in fp64, rounded once
    const double qx_ = e2y * a1z - e2z * a1y, qy_ = e2z * a1x - e2x * a1z, qz_ = e2x * a1y - e2y * a1x;
    const double ux = a0y * e2z - a0z * e2y, uy = a0z * e2x - a0x * e2z, uz = a0x * e2y - a0y * e2x;
    const double hx = e1y * a0z - e1z * a0y, hy = e1z * a0x - e1x * a0z, hz = e1x * a0y - e1y * a0x;
    const double k = a0x * nx + a0y * ny + a0z * nz;
    const double r0 = sqrt(a0x * a0x + a0y * a0y + a0z * a0z);
    const double r1 = sqrt(a1x * a1x + a1y * a1y + a1z * a1z);
    const double r2 = sqrt(a2x * a2x + a2y * a2y + a2z * a2z);
    const double u20 = 9.5367431640625e-07;      // 2^-20
    // the floor keeps the bound above the reference's own fp64 rounding for a triangle at P
    const double big = (double)__uint_as_float(*max_bits) + fabs(px) + fabs(py) + fabs(pz);
    const double rho = fmax(r0, fmax(r1, r2)) + u20 * big;
    const double delta = u20 * dmax * rho * fmax(l1, l2);
    double c = delta * (dmax * nn + delta) * 1.001;
    if (c < 1.2e-38) c = 1.2e-38;
    double m = 32.0 * delta;
    if (nn <= 9.3132257461547852e-10 * l1 * l2) m = (double)__builtin_huge_valf();     // 2^-30
    const double en = u20 * dmax * nn;
    if (sign_layout) {
        // Layout for the pixel-grid sweep, whose rays all leave P (t > 0): a hit has
        // t = k / det > 0, so det has the sign of k, and so have U and W' (0 <= u <= 1).  With
        // every vector multiplied by s = sign(k) the u slab is U >= 0 and W' >= 0, which the
        // sweep reads off two sign bits:
        //     plane 0 = (-c s p.x, -s p.y, -s p.z zs - delta, delta)    D.plane0 = -(U + delta)
        //     plane 1 = (-c s q.x, -s q.y, -s q.z zs - delta, m)        D.plane1 = -(W' + delta)
        //     plane 2 = s (h, k)     plane 3 = (s nn, en)
        // for D = (X / c, Y, 1), c = RT_RASTER_COLS (the pixels of a lane are c columns apart).  delta (16 * 2^-24 dmax rho l) covers the rounding of the stored
        // constants, of the two FMAs and of the seven additions that step along a lane's pixels.
        // When k is too small for its sign to be trusted (P within rounding of the triangle's
        // plane, which includes the nearly degenerate triangles) no fp32 verdict is possible and
        // every pixel passes the triangle on to the exact pass.
        const bool ambiguous = !(fabs(k) > 9.3132257461547852e-10 * rho * l1 * l2);     // 2^-30
        if (squares) {
            // Bounding box of the triangle on the pixel grid, the sweep's first test of a pair: a
            // pixel (X, Y) can only see the triangle if it lies in the triangle's central
            // projection onto the image plane z = zs.  Stored: (cx / c, rx / c, cy, ry) with
            // c = RT_RASTER_COLS (the sweep works with X / c), grown by more than the fp32
            // rounding of the test (the centre and two subtractions, at magnitudes up to dmax)
            // and than the reference's own fp64 rounding can move a hit; r = -1: behind the eye;
            // r = +inf: no bound -- the triangle reaches the eye's plane z = 0, or the fp32
            // filter has no verdict for it (nearly degenerate, or the eye within rounding of its
            // plane: the reference's rounding then decides hits anywhere along a line).
            // (The reference's fp64 determinants carry errors of about 8 * 2^-53 * rho * l1 * l2 * |D|,
            // which move a reported hit by up to 8 * 2^-53 * rho^2 * l1 * l2 / |k| in space: with
            // |k| > 2^-26 rho l1 l2 that is 6e-8 rho, 2e-4 pixel at 4K -- far inside the 0.01.)
            const float inf = __builtin_huge_valf();
            float4 box = make_float4(0, inf, 0, inf);
            const double near_eye = 1e-6 * rho;
            const bool verdict = m < 1e300 && fabs(k) > 1.4901161193847656e-08 * rho * l1 * l2;      // 2^-26
            if (a0z < -near_eye && a1z < -near_eye && a2z < -near_eye && verdict) {
                box = make_float4(0, -1.0f, 0, -1.0f);
            } else if (a0z > near_eye && a1z > near_eye && a2z > near_eye && verdict) {
                const double X0 = zs * a0x / a0z, X1 = zs * a1x / a1z, X2 = zs * a2x / a2z;
                const double Y0 = zs * a0y / a0z, Y1 = zs * a1y / a1z, Y2 = zs * a2y / a2z;
                // only the image matters: |X|, |Y| <= dmax
                const double x0 = fmax(fmin(X0, fmin(X1, X2)), -dmax), x1 = fmin(fmax(X0, fmax(X1, X2)), dmax);
                const double y0 = fmax(fmin(Y0, fmin(Y1, Y2)), -dmax), y1 = fmin(fmax(Y0, fmax(Y1, Y2)), dmax);
                if (x0 > x1 || y0 > y1) {
                    box = make_float4(0, -1.0f, 0, -1.0f);
                } else if (x0 == x0 && x1 == x1 && y0 == y0 && y1 == y1) {
                    const double allow = 0.01 + 9.5367431640625e-07 * dmax;     // pixels
                    const double xs = (double)RT_RASTER_COLS;
                    box = make_float4((float)(0.5 * (x0 + x1) / xs),
                                      (float)((0.5 * (x1 - x0) * 1.0001 + allow) / xs) * 1.0000002f,
                                      (float)(0.5 * (y0 + y1)),
                                      (float)(0.5 * (y1 - y0) * 1.0001 + allow) * 1.0000002f);
                }
            }
            squares[i] = box;
        }
        if (ambiguous) {
            *p[0] = make_float4(0, 0, -1.0f, 0);
            *p[1] = make_float4(0, 0, -1.0f, __builtin_huge_valf());
            *p[2] = make_float4(0, 0, 0, 0);
            *p[3] = make_float4(0, 0, 0, 0);
            return;
        }
        const double sg = k < 0.0 ? -1.0 : 1.0;
        const float df = (float)delta * 1.0000002f;
        const double xs = (double)RT_RASTER_COLS;
        *p[0] = make_float4((float)(-xs * sg * ux), (float)(-sg * uy), (float)(-sg * zs * uz - delta), df);
        *p[1] = make_float4((float)(-xs * sg * qx_), (float)(-sg * qy_), (float)(-sg * zs * qz_ - delta), (float)m);
        *p[2] = make_float4((float)(sg * hx), (float)(sg * hy), (float)(sg * zs * hz), (float)(sg * k));
        *p[3] = make_float4((float)(sg * nx), (float)(sg * ny), (float)(sg * zs * nz), (float)en * 1.0000002f);
        return;
    }
    *p[0] = make_float4((float)ux, (float)uy, (float)(zs * uz), (float)c * 1.0000002f);
    *p[1] = make_float4((float)qx_, (float)qy_, (float)(zs * qz_), (float)m);
    *p[2] = make_float4((float)hx, (float)hy, (float)(zs * hz), (float)k);
    *p[3] = make_float4((float)nx, (float)ny, (float)(zs * nz), (float)en * 1.0000002f);
    if (!squares) return;
    // Bounding squares for the any-hit sweep's first test, one per dominant axis.  A ray of axis
    // kk has the direction D' = (u, v) on the plane D_kk = 1 through P, |u|, |v| <= 1, and can
    // only hit the triangle if (u, v) lies in the triangle's central projection onto that plane.
    // Stored: (cu, cv, rho, 0) = the bounding square of that projection, clipped to the rays'
    // square and grown by more than the fp32 rounding of the test and than the reference's own
    // fp64 rounding can move a hit; rho = -1: no ray of this axis can hit (the triangle lies
    // beyond one face of the double pyramid |D_u|, |D_v| <= |D_kk|); rho = +inf: no bound -- the
    // triangle crosses or touches the plane D_kk = 0 inside the pyramid, or it is one of those
    // for which the fp32 filter has no verdict (nearly degenerate, or P within rounding of its
    // plane: the reference's rounding then decides hits anywhere along a line).
    // (The reference's fp64 determinants carry errors of about 8 * 2^-53 * rho * l1 * l2 * |D|, which
    // move a reported hit by up to 8 * 2^-53 * rho^2 * l1 * l2 / |k| in space: with
    // |k| > 2^-26 rho l1 l2 that is 6e-8 rho, 1.2e-7 on the unit plane -- inside the 1e-6 below.)
    const bool no_verdict = !(m < 1e300) || !(fabs(k) > 1.4901161193847656e-08 * rho * l1 * l2);     // 2^-26
    const double A0[3] = {a0x, a0y, a0z}, A1[3] = {a1x, a1y, a1z}, A2[3] = {a2x, a2y, a2z};
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        const int iu = ax == 0 ? 1 : 0, iv = ax == 2 ? 1 : 2;       // the (u, v) components of dot_ax
        const double wk[3] = {A0[ax], A1[ax], A2[ax]}, wu[3] = {A0[iu], A1[iu], A2[iu]}, wv[3] = {A0[iv], A1[iv], A2[iv]};
        float4 sq = make_float4(0, 0, __builtin_huge_valf(), 0);
        const double grow = 1.000001;
        bool beyond = false;        // all three vertices beyond the same face of the double pyramid
        {
            bool up = true, un = true, vp = true, vn = true;
#pragma unroll
            for (int n = 0; n < 3; ++n) {
                const double lim = fabs(wk[n]) * grow + 1e-9 * rho;
                up = up && wu[n] > lim;
                un = un && -wu[n] > lim;
                vp = vp && wv[n] > lim;
                vn = vn && -wv[n] > lim;
            }
            beyond = up || un || vp || vn;
        }
        const bool one_side = (wk[0] > 0 && wk[1] > 0 && wk[2] > 0) || (wk[0] < 0 && wk[1] < 0 && wk[2] < 0);
        const double nearest = fmin(fabs(wk[0]), fmin(fabs(wk[1]), fabs(wk[2])));
        if (beyond && !no_verdict) {
            sq.z = -1.0f;
        } else if (!no_verdict && one_side && nearest > 1e-6 * rho) {
            double u0 = 1e300, u1 = -1e300, v0 = 1e300, v1 = -1e300;
#pragma unroll
            for (int n = 0; n < 3; ++n) {
                const double pu = wu[n] / wk[n], pv = wv[n] / wk[n];
                u0 = fmin(u0, pu); u1 = fmax(u1, pu);
                v0 = fmin(v0, pv); v1 = fmax(v1, pv);
            }
            // only |u|, |v| <= 1 matter
            const double lim = 1.0 + 1e-6;
            u0 = fmax(u0, -lim); u1 = fmin(u1, lim);
            v0 = fmax(v0, -lim); v1 = fmin(v1, lim);
            if (u0 > u1 || v0 > v1) {
                sq.z = -1.0f;
            } else if (u0 == u0 && u1 == u1 && v0 == v0 && v1 == v1) {
                const double cu = 0.5 * (u0 + u1), cv = 0.5 * (v0 + v1);
                const double half = fmax(0.5 * (u1 - u0), 0.5 * (v1 - v0));
                // the centre is rounded to fp32 (<= 2^-24), the ray's u, v were (<= 2^-24 each), the
                // two subtractions round (<= 2^-23 each): 1e-6 covers them; 1e-4 of the extent
                // covers the reference's own rounding for every triangle that has a verdict
                const float r = (float)(half * 1.0001 + 1e-6) * 1.0000002f;
                sq = make_float4((float)cu, (float)cv, r, 0);
            }
        }
        squares[(long)ax * num_tris_padded + i] = sq;
    }
}

}  // namespace

int rt_launch_pack_apex(rt_scene* sc, const double apex[3], double zs, double dmax, float4* dst,
                        bool sign_layout, const int32_t* order, hipStream_t st, float4* squares)
{
    const int np = sc->num_tris_padded;
    if (!dst) {
        rt_set_error("apex pack: no plane buffer");
        return RT_E_ARG;
    }
    hipLaunchKernelGGL(k_pack_apex, dim3((np + RT_WG - 1) / RT_WG), dim3(RT_WG), 0, st,
                       sc->ex.as<double>(), sc->ey.as<double>(), sc->ez.as<double>(),
                       sc->iA.as<int>(), sc->iB.as<int>(), sc->iC.as<int>(), sc->num_tris, np,
                       sc->scratch_max.as<unsigned>(), apex[0], apex[1], apex[2], zs, dmax,
                       sign_layout ? 1 : 0, order, sc->transformed ? sc->tri_verts.as<double>() : nullptr, dst, squares);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        rt_set_error("apex pack launch failed: %s", hipGetErrorString(e));
        return RT_E_HIP;
    }
    return RT_OK;
}

int rt_launch_transform(const double* m16, const double* x, const double* y, const double* z,
                        double* X, double* Y, double* Z, int n, unsigned* max_bits, hipStream_t st)
{
    if (n <= 0) return RT_OK;
    const int per_wg = RT_WG * kTransformPoints;
    hipLaunchKernelGGL(k_transform<const double*>, dim3((n + per_wg - 1) / per_wg), dim3(RT_WG), 0, st, m16, x, y, z,
                       X, Y, Z, n, max_bits);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        rt_set_error("transform launch failed: %s", hipGetErrorString(e));
        return RT_E_HIP;
    }
    return RT_OK;
}

int rt_launch_transform(const RtMat12& m, const double* x, const double* y, const double* z,
                        double* X, double* Y, double* Z, int n, unsigned* max_bits, hipStream_t st)
{
    if (n <= 0) return RT_OK;
    const int per_wg = RT_WG * kTransformPoints;
    hipLaunchKernelGGL(k_transform<RtMat12>, dim3((n + per_wg - 1) / per_wg), dim3(RT_WG), 0, st, m, x, y, z,
                       X, Y, Z, n, max_bits);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        rt_set_error("transform launch failed: %s", hipGetErrorString(e));
        return RT_E_HIP;
    }
    return RT_OK;
}

int rt_launch_pack(rt_scene* sc, hipStream_t st, bool planes, bool verts)
{
    const int np = sc->num_tris_padded;
    hipLaunchKernelGGL(k_pack, dim3((np + RT_WG - 1) / RT_WG), dim3(RT_WG), 0, st,
                       sc->ex.as<double>(), sc->ey.as<double>(), sc->ez.as<double>(),
                       sc->iA.as<int>(), sc->iB.as<int>(), sc->iC.as<int>(), sc->num_tris, np,
                       sc->scratch_max.as<unsigned>(), planes ? sc->planes.as<float4>() : nullptr,
                       verts ? sc->tri_verts.as<double>() : nullptr);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        rt_set_error("pack launch failed: %s", hipGetErrorString(e));
        return RT_E_HIP;
    }
    return RT_OK;
}
