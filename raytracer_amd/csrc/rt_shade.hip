// rt_shade.hip -- shading epilogue of the pixel loop (driver.c:136-190), fp64, one pixel
// per lane, compiled without FMA contraction so every sum and product is rounded as in
// the reference's host code.
//
//   k_shadow_setup : for each hit pixel and light, the part of Light_Model_Single up to the
//                    shadow-ray call (light_model.c:36-63) -- emits the shadow ray
//                    (S' = p + 0.001*L, E' = light) or marks the slot unused.
//   [any-hit sweep over the shadow rays, rt_sweep.hip + rt_exact.hip]
//   k_shade_final  : interpolate_normal_vector (light_model.c:156-186), material selection
//                    (driver.c:143-182), Light_Model (light_model.c:124-151) with the
//                    occlusion bits, clamp, and the (Uint8)(c*255) store
//                    (graphic_tools.c:106-113).
//
// A light whose evaluation returns 0 in the reference (zero-length normal, L or V:
// light_model.c:39-40,47-48,82) contributes nothing here; the reference adds an
// uninitialised array in that case.
#include "rt_internal.h"

namespace {

constexpr double K_AMBIENT = 0.2;     // light_model.c:13
constexpr double K_DIFFUSE = 0.5;     // light_model.c:14
constexpr double K_SPECPOW = 50.0;    // light_model.c:15

__device__ __forceinline__ double dot3(const double a[3], const double b[3])
{
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];     // M3d_matrix_tools.c:238-242
}

// M3d_matrix_tools.c:244-255
__device__ __forceinline__ bool norm3(double res[3], const double a[3])
{
    const double mag = sqrt(dot3(a, a));
    if (mag == 0) return false;
    res[0] = a[0] / mag; res[1] = a[1] / mag; res[2] = a[2] / mag;
    return true;
}

// driver.c:121-133 + triangle_intersection.cu:185-187
__device__ __forceinline__ void hit_point(const ShadeArgs& a, long p, double point[3])
{
    const int py = a.rows.image_row((int)(p / a.W)), px = (int)(p % a.W);
    const double S[3] = {0, 0, 0};
    const double E[3] = {(double)(px - a.W / 2), (double)(py - a.H / 2),
                         (double)(a.W / 2) / a.tan_half};
    const double t = a.tuv[3 * p];
    point[0] = S[0] + t * (E[0] - S[0]);
    point[1] = S[1] + t * (E[1] - S[1]);
    point[2] = S[2] + t * (E[2] - S[2]);
}

// light_model.c:156-186
__device__ __forceinline__ void interp_normal(const ShadeArgs& a, int tri, double u, double v,
                                              double normal[3])
{
    double An[3], Bn[3], Cn[3];
    if (a.tn) {
        // the three vertex normals from the per-triangle copy (k_pack_normals): neighbouring pixels
        // of a fine mesh hit different triangles, and three indices with nine coordinates fetched
        // through them are twelve cache lines a pixel where the copy is two
        const double* t = a.tn + 9 * (long)tri;
        An[0] = t[0]; An[1] = t[1]; An[2] = t[2];
        Bn[0] = t[3]; Bn[1] = t[4]; Bn[2] = t[5];
        Cn[0] = t[6]; Cn[1] = t[7]; Cn[2] = t[8];
    } else {
        const int ia = a.An[tri], ib = a.Bn[tri], ic = a.Cn[tri];
        An[0] = a.nx[ia]; An[1] = a.ny[ia]; An[2] = a.nz[ia];
        Bn[0] = a.nx[ib]; Bn[1] = a.ny[ib]; Bn[2] = a.nz[ib];
        Cn[0] = a.nx[ic]; Cn[1] = a.ny[ic]; Cn[2] = a.nz[ic];
    }
    double Nt[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) Nt[k] = (1 - u - v) * An[k] + u * Bn[k] + v * Cn[k];
    const DevFrame& f = *a.frame;
#pragma unroll
    for (int k = 0; k < 3; ++k)
        normal[k] = Nt[0] * f.obinv[0][k] + Nt[1] * f.obinv[1][k] + Nt[2] * f.obinv[2][k];
}

// What Light_Model_Single computes before it fires the shadow ray (light_model.c:36-63).
// status: 0 = the reference returns 0 here (no contribution), 1 = beyond light_max_dist
// (rgb = 0, no shadow ray), 2 = shadow ray needed.
struct LightGeom {
    double N[3], L[3], len, NdotL, attenuation, shadow_S[3];
    int status;
};

__device__ __forceinline__ void light_geom(const DevFrame& f, int l, const double p[3],
                                           const double n[3], LightGeom& g)
{
    g.status = 0;
    double len = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    if (len == 0) return;
    if (!norm3(g.N, n)) return;
    g.L[0] = f.light_pos[l][0] - p[0];
    g.L[1] = f.light_pos[l][1] - p[1];
    g.L[2] = f.light_pos[l][2] - p[2];
    len = sqrt(g.L[0] * g.L[0] + g.L[1] * g.L[1] + g.L[2] * g.L[2]);
    if (len == 0) return;
    if (!norm3(g.L, g.L)) return;
    g.len = len;
    g.NdotL = dot3(g.N, g.L);
    if (len > f.light_max[l]) { g.status = 1; return; }
    g.attenuation = 1 + f.light_att[l][0] * len + f.light_att[l][1] * len * len;
    g.shadow_S[0] = p[0] + g.L[0] * 0.001;
    g.shadow_S[1] = p[1] + g.L[1] * 0.001;
    g.shadow_S[2] = p[2] + g.L[2] * 0.001;
    g.status = 2;
}

__global__ void __launch_bounds__(RT_WG) k_shadow_setup(const ShadeArgs a)
{
    const long npix = (long)a.W * a.rows.rows;
    const long p = (long)blockIdx.x * RT_WG + threadIdx.x;
    if (p >= npix) return;
    const DevFrame& f = *a.frame;
    const int s = a.hit[p];
    // a pixel that hit nothing has no shadow rays: its slots are not even written, and every
    // reader of the ray arrays looks at hit[] first (k_bin_rays, k_resolve_rays)
    if (s == -1) return;
    double point[3], normal[3];
    hit_point(a, p, point);
    interp_normal(a, s, a.tuv[3 * p + 1], a.tuv[3 * p + 2], normal);
    const double qnan = __builtin_nan("");
    for (int l = 0; l < a.num_lights; ++l) {
        double* S = a.shS + 3 * ((long)l * npix + p);
        double* E = a.shE + 3 * ((long)l * npix + p);
        LightGeom g;
        light_geom(f, l, point, normal, g);
        int axis = -1;
        if (g.status == 2) {
            S[0] = g.shadow_S[0]; S[1] = g.shadow_S[1]; S[2] = g.shadow_S[2];
            E[0] = f.light_pos[l][0]; E[1] = f.light_pos[l][1]; E[2] = f.light_pos[l][2];
            // what k_bin_rays would otherwise read both end points again for: the dominant axis of
            // the direction, or no ray at all (unused slot, zero-length or non-finite direction)
            const double sx = S[0];
            const double dx = fabs(E[0] - sx), dy = fabs(E[1] - S[1]), dz = fabs(E[2] - S[2]);
            if (sx == sx && dx + dy + dz > 0 && dx + dy + dz < 1e300) axis = dx >= dy && dx >= dz ? 0 : (dy >= dz ? 1 : 2);
        } else {
            S[0] = S[1] = S[2] = qnan;      // unused slot: the sweep skips it
            E[0] = E[1] = E[2] = 0;
        }
        if (a.sh_axis) a.sh_axis[(long)l * npix + p] = (int8_t)axis;
    }
}

// graphic_tools.c:161-197 on a tightly packed RGB8 image; the linear texel index is clamped
// into the image (the reference indexes out of bounds there, which is undefined).
__device__ __forceinline__ void tex_rgb(const ShadeArgs& a, int tex, int tri, double u, double v,
                                        double rgb[3])
{
    const DevTexture t = a.tex[tex];
    const int ia = a.At[tri], ib = a.Bt[tri], ic = a.Ct[tri];
    double x = (1 - u - v) * a.tu[ia] + u * a.tu[ib] + v * a.tu[ic];
    double y = (1 - u - v) * a.tv[ia] + u * a.tv[ib] + v * a.tv[ic];
    y = 1 - y;
    x *= (double)t.w; y *= (double)t.h;
    const int xi = (int)x, yi = (int)y;
    long idx = (long)yi * t.w + xi;
    const long last = (long)t.w * t.h - 1;
    if (idx < 0) idx = 0;
    if (idx > last) idx = last;
    const uint8_t* px = t.rgb + 3 * idx;
    rgb[0] = (px[0] * 1.0) / 255;
    rgb[1] = (px[1] * 1.0) / 255;
    rgb[2] = (px[2] * 1.0) / 255;
}

__global__ void __launch_bounds__(RT_WG) k_shade_final(const ShadeArgs a)
{
    const long npix = (long)a.W * a.rows.rows;
    const long p = (long)blockIdx.x * RT_WG + threadIdx.x;
    if (p >= npix) return;
    const DevFrame& f = *a.frame;
    const int s = a.hit[p];
    double argb[3];
    if (s == -1) {
        argb[0] = argb[1] = argb[2] = 0.5;                          // driver.c:136-139
    } else {
        const double u = a.tuv[3 * p + 1], v = a.tuv[3 * p + 2];
        double point[3], normal[3];
        hit_point(a, p, point);
        interp_normal(a, s, u, v, normal);                          // driver.c:142
        double Ka[3], Kd[3], Ks[3];
        const int m = a.mtl ? a.mtl[s] : -1;
        if (m == -1) {                                              // driver.c:177-182
#pragma unroll
            for (int k = 0; k < 3; ++k) Ka[k] = Kd[k] = Ks[k] = 1.0;
        } else {                                                    // driver.c:143-175
            const DevMaterial& mt = a.mats[m];
            if (mt.map_Ka >= 0) tex_rgb(a, mt.map_Ka, s, u, v, Ka);
            else { Ka[0] = mt.Ka[0]; Ka[1] = mt.Ka[1]; Ka[2] = mt.Ka[2]; }
            if (mt.map_Kd >= 0) tex_rgb(a, mt.map_Kd, s, u, v, Kd);
            else { Kd[0] = mt.Kd[0]; Kd[1] = mt.Kd[1]; Kd[2] = mt.Kd[2]; }
            if (mt.map_Ks >= 0) tex_rgb(a, mt.map_Ks, s, u, v, Ks);
            else { Ks[0] = mt.Ks[0]; Ks[1] = mt.Ks[1]; Ks[2] = mt.Ks[2]; }
        }
        // Light_Model, light_model.c:124-151
#pragma unroll
        for (int k = 0; k < 3; ++k) argb[k] = K_AMBIENT * Ka[k];
        const double origin[3] = {0, 0, 0};
        for (int l = 0; l < a.num_lights; ++l) {
            LightGeom g;
            light_geom(f, l, point, normal, g);
            if (g.status == 0) continue;
            double rgb[3] = {0, 0, 0};
            const bool blocked = g.status == 2 && a.occ[(long)l * npix + p] != 0;   // :70-76
            if (g.status == 2 && !blocked) {
                double V[3] = {origin[0] - point[0], origin[1] - point[1], origin[2] - point[2]};
                if (!norm3(V, V)) continue;                         // :82
                double diffuse[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {                       // :89-96
                    diffuse[k] = g.NdotL * Kd[k] * K_DIFFUSE * f.light_rgb[l][k];
                    if (diffuse[k] < 0) diffuse[k] = 0;
                }
                double R[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) R[k] = 2 * g.NdotL * g.N[k] - g.L[k];   // :103-105
                const double RdotV = dot3(R, V);
                double fs;
                if (RdotV > 0) fs = pow(RdotV, K_SPECPOW);          // :108-109
                else fs = 0;
                const double spec_w = 1.0 - K_DIFFUSE - K_AMBIENT;
#pragma unroll
                for (int k = 0; k < 3; ++k) {                       // :112-119
                    const double specular = fs * Ks[k] * spec_w * f.light_rgb[l][k];
                    rgb[k] = (diffuse[k] + specular) / g.attenuation;
                }
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) argb[k] += rgb[k];          // :143-145
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) if (argb[k] > 1) argb[k] = 1;   // :147-149
    }
    if (a.argb) { a.argb[3 * p] = argb[0]; a.argb[3 * p + 1] = argb[1]; a.argb[3 * p + 2] = argb[2]; }
    if (a.rgb8) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            double c = argb[k];
            c *= 255;                                               // graphic_tools.c:108-110
            a.rgb8[3 * p + k] = (uint8_t)(int)c;
        }
    }
}

// driver.c:186-190: pixel (x_pix, y_pix) lands on screen row H - y_pix; y_pix = 0 is off-screen.
__global__ void __launch_bounds__(RT_WG)
k_place_rows(const uint8_t* __restrict__ rgb8, int W, int H, RowMap rows,
             uint8_t* __restrict__ screen)
{
    const long n = (long)W * rows.rows * 3;
    const long i = (long)blockIdx.x * RT_WG + threadIdx.x;
    if (i >= n) return;
    const long row = i / ((long)W * 3), col = i - row * ((long)W * 3);
    const int y_pix = rows.image_row((int)row);
    if (y_pix < 1 || y_pix >= H) return;
    screen[(long)(H - y_pix) * W * 3 + col] = rgb8[i];
}

// The vertex normals of every triangle side by side (object space, as uploaded): 9 doubles.
__global__ void __launch_bounds__(RT_WG)
k_pack_normals(const double* __restrict__ nx, const double* __restrict__ ny, const double* __restrict__ nz,
               const int* __restrict__ An, const int* __restrict__ Bn, const int* __restrict__ Cn, int num_tris,
               double* __restrict__ out)
{
    const int i = blockIdx.x * RT_WG + threadIdx.x;
    if (i >= num_tris) return;
    const int ia = An[i], ib = Bn[i], ic = Cn[i];
    double* t = out + 9 * (long)i;
    t[0] = nx[ia]; t[1] = ny[ia]; t[2] = nz[ia];
    t[3] = nx[ib]; t[4] = ny[ib]; t[5] = nz[ib];
    t[6] = nx[ic]; t[7] = ny[ic]; t[8] = nz[ic];
}

int launched(const char* what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        rt_set_error("%s launch failed: %s", what, hipGetErrorString(e));
        return RT_E_HIP;
    }
    return RT_OK;
}

}  // namespace

int rt_launch_pack_normals(const double* nx, const double* ny, const double* nz, const int* An, const int* Bn,
                           const int* Cn, int num_tris, double* out, hipStream_t st)
{
    if (num_tris <= 0) return RT_OK;
    hipLaunchKernelGGL(k_pack_normals, dim3((unsigned)((num_tris + RT_WG - 1) / RT_WG)), dim3(RT_WG), 0, st, nx, ny,
                       nz, An, Bn, Cn, num_tris, out);
    return launched("pack_normals");
}

int rt_launch_shadow_setup(const ShadeArgs& a, hipStream_t st)
{
    const long npix = (long)a.W * a.rows.rows;
    if (npix <= 0 || a.num_lights <= 0) return RT_OK;
    hipLaunchKernelGGL(k_shadow_setup, dim3((unsigned)((npix + RT_WG - 1) / RT_WG)), dim3(RT_WG), 0,
                       st, a);
    return launched("shadow_setup");
}

int rt_launch_shade_final(const ShadeArgs& a, hipStream_t st)
{
    const long npix = (long)a.W * a.rows.rows;
    if (npix <= 0) return RT_OK;
    hipLaunchKernelGGL(k_shade_final, dim3((unsigned)((npix + RT_WG - 1) / RT_WG)), dim3(RT_WG), 0,
                       st, a);
    return launched("shade_final");
}

int rt_launch_place_rows(const uint8_t* rgb8, int W, int H, const RowMap& rows, uint8_t* screen,
                         hipStream_t st)
{
    const long n = (long)W * rows.rows * 3;
    if (n <= 0) return RT_OK;
    hipLaunchKernelGGL(k_place_rows, dim3((unsigned)((n + RT_WG - 1) / RT_WG)), dim3(RT_WG), 0, st,
                       rgb8, W, H, rows, screen);
    return launched("place_rows");
}
