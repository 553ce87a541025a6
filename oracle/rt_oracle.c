/*
 * rt_oracle.c -- CPU oracle for the ray-triangle hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the checker, never the product: only tests/, __graft_entry__.smoke() and
 * the cpu_baseline leg of bench.py may load it.  The shipped path (raytracer_amd/) never
 * links or calls anything in here and fails loudly when its HIP library is missing.
 *
 * It is a plain-C, strict IEEE fp64 restatement (build with -O2 -ffp-contract=off) of the
 * arithmetic of the reference, statement order preserved so that results are reproducible
 * bit for bit.  Every function cites the reference lines it follows
 * (paths relative to the reference checkout).
 *
 * Pinning status:
 *   - orc_det_3x3, orc_mat_mult_points, the 4x4 helpers and orc_view are checked bit-for-bit
 *     against the reference's own M3d_matrix_tools.c compiled unmodified into
 *     oracle/_ref/libm3d_ref.so (tests/test_oracle_vs_ref.py).
 *   - The intersection test, closest-hit reduction, light model, normal interpolation, ray
 *     generation and pixel store have no golden vectors in the reference (it ships no tests,
 *     no images and no meshes) and the files that hold them cannot be built here
 *     (triangle_intersection.cu needs nvcc + cuda_runtime.h, light_model.c/driver.c need
 *     SDL2 headers): for those parts PARITY IS UNPINNED -- they follow the source text only,
 *     plus first-principles known-answer tests (tests/test_oracle_kat.py).
 *
 * Deliberate departures from the reference text (each an evident defect there):
 *   - vertex arrays are read at their full length; the reference uploads only
 *     8*num_tris bytes of x/y/z (triangle_intersection.cu:102,130-137).
 *   - a light whose Light_Model_Single call returns 0 (zero-length normal, L or V) adds
 *     nothing; the reference adds an uninitialised rgb[] (light_model.c:39-48,82,142-145).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "rt_oracle.h"

/* ------------------------------------------------------------------ determinants */

/* triangle_intersection.cu:6-11 (device) == M3d_matrix_tools.c:257-262 (host twin) */
static double det_2x2(const double A[2], const double B[2])
{
    double det = A[0] * B[1] - A[1] * B[0];
    return det;
}

/* triangle_intersection.cu:13-29 == M3d_matrix_tools.c:264-280: first-row cofactors,
 * accumulated as 0 + A0*d1 + (-B0)*d2 + C0*d3. */
double orc_det_3x3(const double A[3], const double B[3], const double C[3])
{
    double det = 0;
    double tempA[2], tempB[2];
    tempA[0] = B[1]; tempA[1] = B[2];
    tempB[0] = C[1]; tempB[1] = C[2];
    det += A[0] * det_2x2(tempA, tempB);
    tempA[0] = A[1]; tempA[1] = A[2];
    tempB[0] = C[1]; tempB[1] = C[2];
    det += -B[0] * det_2x2(tempA, tempB);
    tempA[0] = A[1]; tempA[1] = A[2];
    tempB[0] = B[1]; tempB[1] = B[2];
    det += C[0] * det_2x2(tempA, tempB);
    return det;
}

/* ------------------------------------------------------------------ one ray, one triangle */

/* triangle_intersection.cu:31-96, body of intersect_single_triangle_device for one i.
 * Returns dist (t, or -1 for a miss); uv[] receives what the kernel leaves in its uv slots. */
double orc_intersect_single_triangle(const double S[3], const double E[3],
                                     const double A[3], const double B[3], const double C[3],
                                     double uv[2])
{
    double AB[3], AC[3], ES[3], AS[3];
    double t;
    int k;
    for (k = 0; k < 3; k++) {          /* :59-62 */
        AB[k] = B[k] - A[k];
        AC[k] = C[k] - A[k];
        ES[k] = S[k] - E[k];
        AS[k] = S[k] - A[k];
    }
    double den = orc_det_3x3(AB, AC, ES);                 /* :65 */
    if (den == 0) return -1;                              /* :67-70 */
    double topt, topu, topv;
    topt = orc_det_3x3(AB, AC, AS);                       /* :72 */
    t = topt / den;                                       /* :73 */
    if (t < 0) return -1;                                 /* :74-77 */
    topu = orc_det_3x3(AS, AC, ES);                       /* :78 */
    uv[0] = topu / den;                                   /* :79 */
    if ((uv[0] < 0) || (uv[0] > 1)) return -1;            /* :81-84 */
    topv = orc_det_3x3(AB, AS, ES);                       /* :85 */
    uv[1] = topv / den;                                   /* :86 */
    if ((uv[1] < 0) || (uv[1] > 1)) return -1;            /* :87-90 */
    if (uv[0] + uv[1] > 1) return -1;                     /* :91-94 */
    return t;                                             /* :95 */
}

/* ------------------------------------------------------------------ one ray, all triangles */

/* triangle_intersection.cu:98-190 with the kernel launch replaced by a loop over i.
 * Closest hit = smallest dist with dist > 0, strict '<' so the lowest index wins a tie
 * (:167-176); uv and point are written only on a hit (:171,183-188). */
int orc_intersect_all_triangles(const double S[3], const double E[3],
                                double uv[2], double point[3],
                                const int* iA, const int* iB, const int* iC,
                                const double* x, const double* y, const double* z,
                                int num_tris)
{
    int i;
    int closest = -1;
    double t = 1e50;                                      /* :108 */
    for (i = 0; i < num_tris; i++) {
        double A[3], B[3], C[3], cand_uv[2];
        A[0] = x[iA[i]]; A[1] = y[iA[i]]; A[2] = z[iA[i]];   /* :53-55 */
        B[0] = x[iB[i]]; B[1] = y[iB[i]]; B[2] = z[iB[i]];
        C[0] = x[iC[i]]; C[1] = y[iC[i]]; C[2] = z[iC[i]];
        double dist = orc_intersect_single_triangle(S, E, A, B, C, cand_uv);
        if ((dist > 0) && (dist < t)) {                   /* :169 */
            uv[0] = cand_uv[0]; uv[1] = cand_uv[1];       /* :171 */
            t = dist;
            closest = i;
        }
    }
    if (closest != -1) {                                  /* :183-188 */
        point[0] = S[0] + t * (E[0] - S[0]);
        point[1] = S[1] + t * (E[1] - S[1]);
        point[2] = S[2] + t * (E[2] - S[2]);
    }
    return closest;
}

/* Same reduction, additionally reporting t (the reference keeps it local, :172). */
static int intersect_all_t(const OrcScene* sc, const double S[3], const double E[3],
                           double uv[2], double point[3], double* t_out)
{
    int i;
    int closest = -1;
    double t = 1e50;
    const double *x = sc->x, *y = sc->y, *z = sc->z;
    for (i = 0; i < sc->num_tris; i++) {
        double A[3], B[3], C[3], cand_uv[2];
        int a = sc->A[i], b = sc->B[i], c = sc->C[i];
        A[0] = x[a]; A[1] = y[a]; A[2] = z[a];
        B[0] = x[b]; B[1] = y[b]; B[2] = z[b];
        C[0] = x[c]; C[1] = y[c]; C[2] = z[c];
        double dist = orc_intersect_single_triangle(S, E, A, B, C, cand_uv);
        if ((dist > 0) && (dist < t)) {
            uv[0] = cand_uv[0]; uv[1] = cand_uv[1];
            t = dist;
            closest = i;
        }
    }
    if (closest != -1) {
        point[0] = S[0] + t * (E[0] - S[0]);
        point[1] = S[1] + t * (E[1] - S[1]);
        point[2] = S[2] + t * (E[2] - S[2]);
        if (t_out) *t_out = t;
    }
    return closest;
}

int orc_scene_intersect(const OrcScene* sc, const double S[3], const double E[3],
                        double uv[2], double point[3], double* t_out)
{
    return intersect_all_t(sc, S, E, uv, point, t_out);
}

/* ------------------------------------------------------------------ vertex transform */

/* mat_mult.cu:8-28 (kernel) == M3d_matrix_tools.c:184-208 (host twin): rows 0..2 of a
 * row-major 4x4, left-to-right sums, safe in place. */
void orc_mat_mult_points(double* X, double* Y, double* Z, const double m[4][4],
                         const double* x, const double* y, const double* z, int numpoints)
{
    int i;
    for (i = 0; i < numpoints; i++) {
        double u = m[0][0] * x[i] + m[0][1] * y[i] + m[0][2] * z[i] + m[0][3];
        double v = m[1][0] * x[i] + m[1][1] * y[i] + m[1][2] * z[i] + m[1][3];
        double w = m[2][0] * x[i] + m[2][1] * y[i] + m[2][2] * z[i] + m[2][3];
        X[i] = u; Y[i] = v; Z[i] = w;
    }
}

/* ------------------------------------------------------------------ 4x4 helpers */

/* M3d_matrix_tools.c:57-69 */
static void mat_identity(double a[4][4])
{
    int r, c;
    for (r = 0; r < 4; r++) for (c = 0; c < 4; c++) a[r][c] = (r == c) ? 1.0 : 0.0;
}

/* M3d_matrix_tools.c:133-157: res = a*b, sum starts at 0.0 and adds k = 0..3 in order;
 * aliasing-safe through a temporary. */
void orc_mat_mult(double res[4][4], const double a[4][4], const double b[4][4])
{
    double tmp[4][4];
    int r, c, k;
    for (r = 0; r < 4; r++)
        for (c = 0; c < 4; c++) {
            double sum = 0.0;
            for (k = 0; k < 4; k++) sum = sum + a[r][k] * b[k][c];
            tmp[r][c] = sum;
        }
    memcpy(res, tmp, sizeof tmp);
}

/* M3d_matrix_tools.c:163-178 */
void orc_mat_mult_pt(double P[3], const double m[4][4], const double Q[3])
{
    double u = m[0][0] * Q[0] + m[0][1] * Q[1] + m[0][2] * Q[2] + m[0][3];
    double v = m[1][0] * Q[0] + m[1][1] * Q[1] + m[1][2] * Q[2] + m[1][3];
    double t = m[2][0] * Q[0] + m[2][1] * Q[1] + m[2][2] * Q[2] + m[2][3];
    P[0] = u; P[1] = v; P[2] = t;
}

/* M3d_matrix_tools.c:75-80, :86-91 */
static void mat_translation(double a[4][4], double dx, double dy, double dz)
{
    mat_identity(a); a[0][3] = dx; a[1][3] = dy; a[2][3] = dz;
}
static void mat_scaling(double a[4][4], double sx, double sy, double sz)
{
    mat_identity(a); a[0][0] = sx; a[1][1] = sy; a[2][2] = sz;
}

/* M3d_matrix_tools.c:93-127; axis 0/1/2 = x/y/z rotation from a known cosine and sine */
static void mat_rotation_cs(double a[4][4], int axis, double cs, double sn)
{
    mat_identity(a);
    if (axis == 0)      { a[1][1] = cs; a[1][2] = -sn; a[2][1] = sn;  a[2][2] = cs; }
    else if (axis == 1) { a[0][0] = cs; a[0][2] = sn;  a[2][0] = -sn; a[2][2] = cs; }
    else                { a[0][0] = cs; a[0][1] = -sn; a[1][0] = sn;  a[1][1] = cs; }
}

/* M3d_matrix_tools.c:305-471.  Codes: 0-2 scale x/y/z, 3-5 rotate x/y/z (degrees),
 * 6-8 translate x/y/z, 9-11 negate x/y/z.  mat = T_k * ... * T_0, inv = T_0^-1 * ... * T_k^-1.
 * Returns 1, or 0 on an unknown code. */
int orc_movement_sequence(double mat[4][4], double inv[4][4], int n,
                          const int* types, const double* params)
{
    double tmp[4][4];
    int i;
    mat_identity(mat);
    mat_identity(inv);
    for (i = 0; i < n; i++) {
        int m = types[i];
        double p = params[i];
        double s[3] = {1.0, 1.0, 1.0};
        if (m >= 0 && m <= 2) {                                     /* :354-376 */
            s[m] = p;       mat_scaling(tmp, s[0], s[1], s[2]);  orc_mat_mult(mat, tmp, mat);
            s[m] = 1.0 / p; mat_scaling(tmp, s[0], s[1], s[2]);  orc_mat_mult(inv, inv, tmp);
        } else if (m >= 3 && m <= 5) {                              /* :381-403 */
            mat_rotation_cs(tmp, m - 3, cos(p * M_PI / 180), sin(p * M_PI / 180));
            orc_mat_mult(mat, tmp, mat);
            mat_rotation_cs(tmp, m - 3, cos(-p * M_PI / 180), sin(-p * M_PI / 180));
            orc_mat_mult(inv, inv, tmp);
        } else if (m >= 6 && m <= 8) {                              /* :408-430 */
            double d[3] = {0.0, 0.0, 0.0};
            d[m - 6] = p;  mat_translation(tmp, d[0], d[1], d[2]);  orc_mat_mult(mat, tmp, mat);
            d[m - 6] = -p; mat_translation(tmp, d[0], d[1], d[2]);  orc_mat_mult(inv, inv, tmp);
        } else if (m >= 9 && m <= 11) {                             /* :436-455 */
            s[m - 9] = -1.0; mat_scaling(tmp, s[0], s[1], s[2]);
            orc_mat_mult(mat, tmp, mat);
            orc_mat_mult(inv, inv, tmp);
        } else {
            return 0;                                               /* :457-462 */
        }
    }
    return 1;
}

/* M3d_matrix_tools.c:484-527 (AA: |b| < p) and :536-581 (BB) share everything but the first
 * two rotations; :591-628 picks between them. */
int orc_view(double view[4][4], double view_inverse[4][4],
             const double eye[3], const double coi[3], const double up[3])
{
    double a = coi[0] - eye[0];
    double b = coi[1] - eye[1];
    double c = coi[2] - eye[2];
    double p = sqrt(a * a + c * c);
    double r = sqrt(a * a + b * b + c * c);
    int aa = fabs(b) < p;                                           /* :610 */
    double A[4][4], B[4][4], C[4][4], D[4][4];
    double Ai[4][4], Bi[4][4], Ci[4][4], Di[4][4];
    double Up[3], Uplen;

    if (!aa) p = sqrt(b * b + c * c);                               /* :547 */
    if ((p == 0) || (r == 0)) return 0;                             /* :495, :549 */

    mat_translation(A, -eye[0], -eye[1], -eye[2]);
    if (aa) {                                                       /* :498-499 */
        mat_rotation_cs(B, 1, c / p, -a / p);
        mat_rotation_cs(C, 0, p / r, b / r);
    } else {                                                        /* :552-553 */
        mat_rotation_cs(B, 0, c / p, b / p);
        mat_rotation_cs(C, 1, p / r, -a / r);
    }
    mat_identity(view);
    orc_mat_mult(view, A, view);
    orc_mat_mult(view, B, view);
    orc_mat_mult(view, C, view);

    orc_mat_mult_pt(Up, view, up);
    Uplen = sqrt(Up[0] * Up[0] + Up[1] * Up[1]);
    if (Uplen == 0) return 0;
    mat_rotation_cs(D, 2, Up[1] / Uplen, Up[0] / Uplen);
    orc_mat_mult(view, D, view);

    mat_translation(Ai, eye[0], eye[1], eye[2]);
    if (aa) {                                                       /* :516-517 */
        mat_rotation_cs(Bi, 1, c / p, a / p);
        mat_rotation_cs(Ci, 0, p / r, -b / r);
    } else {                                                        /* :570-571 */
        mat_rotation_cs(Bi, 0, c / p, -b / p);
        mat_rotation_cs(Ci, 1, p / r, a / r);
    }
    mat_rotation_cs(Di, 2, Up[1] / Uplen, -Up[0] / Uplen);
    mat_identity(view_inverse);
    orc_mat_mult(view_inverse, Di, view_inverse);
    orc_mat_mult(view_inverse, Ci, view_inverse);
    orc_mat_mult(view_inverse, Bi, view_inverse);
    orc_mat_mult(view_inverse, Ai, view_inverse);
    return 1;
}

/* driver.c:34-107: camera orbit, the two lights, object matrix for one frame.
 * lights_world[l] = world position; outputs everything the pixel loop needs. */
void orc_driver_frame_setup(int frame_number, int frames_per_turn, OrcFrame* f)
{
    double eangle = 2 * M_PI * frame_number / frames_per_turn;      /* :34 */
    double eye[3], coi[3], up[3], vm[4][4], vi[4][4], m[4][4], mi[4][4], lw[3];
    int types[1];
    double params[1];
    eye[0] = 15.0 * cos(eangle); eye[1] = 10.0; eye[2] = 15.0 * sin(eangle);   /* :38-40 */
    coi[0] = coi[1] = coi[2] = 0;
    up[0] = eye[0]; up[1] = eye[1] + 1; up[2] = eye[2];             /* :52-54 */
    orc_view(vm, vi, eye, coi, up);

    f->num_lights = 0;
    lw[0] = 0.0; lw[1] = 15.0 + frame_number; lw[2] = 0.0;          /* :63-65 */
    f->light_rgb[0][0] = 1.0; f->light_rgb[0][1] = 1.0; f->light_rgb[0][2] = 1.0;
    f->light_attenuation[0][0] = 0.045; f->light_attenuation[0][1] = 0.0075;
    f->light_max_dist[0] = 100;
    orc_mat_mult_pt(f->light_in_eye_space[0], vm, lw);
    f->num_lights++;
    f->light_rgb[1][0] = 0.1; f->light_rgb[1][1] = 0.0; f->light_rgb[1][2] = 1.0;   /* :79-81 */
    lw[0] = -10; lw[1] = 10; lw[2] = 0;
    f->light_attenuation[1][0] = 0.09; f->light_attenuation[1][1] = 0.032;
    f->light_max_dist[1] = 50;
    orc_mat_mult_pt(f->light_in_eye_space[1], vm, lw);
    f->num_lights++;

    types[0] = 3; params[0] = 270;                                  /* :103, RX = 3 */
    orc_movement_sequence(m, mi, 1, types, params);
    orc_mat_mult(f->obmat, vm, m);                                  /* :106 */
    orc_mat_mult(f->obinv, mi, vi);                                 /* :107 */
}

/* obj_reader.c:239-289: centroid to the origin, largest bounding-box extent to 15 units.
 * Vertices are 1-indexed; slot 0 is transformed too, as the reference does (:288). */
void orc_center_and_scale(double* x, double* y, double* z, int num_v, double m_out[4][4])
{
    double xc = 0, yc = 0, zc = 0;
    double x_min, y_min, z_min, x_max, y_max, z_max;
    double t[4][4], s[4][4], m[4][4];
    int i;
    x_min = x_max = x[1]; y_min = y_max = y[1]; z_min = z_max = z[1];
    for (i = 1; i <= num_v; i++) {
        xc += x[i]; yc += y[i]; zc += z[i];
        if (x[i] < x_min) x_min = x[i];
        if (y[i] < y_min) y_min = y[i];
        if (z[i] < z_min) z_min = z[i];
        if (x[i] > x_max) x_max = x[i];
        if (y[i] > y_max) y_max = y[i];
        if (z[i] > z_max) z_max = z[i];
    }
    xc /= num_v; yc /= num_v; zc /= num_v;
    double xd = x_max - x_min, yd = y_max - y_min, zd = z_max - z_min;
    double delta;
    if ((xd > yd) && (xd > zd)) delta = xd;                         /* :275-277 */
    else if (yd > zd) delta = yd;
    else delta = zd;
    double sf = 15 / delta;
    mat_translation(t, -xc, -yc, -zc);
    mat_scaling(s, sf, sf, sf);
    orc_mat_mult(m, s, t);
    orc_mat_mult_points(x, y, z, m, x, y, z, num_v + 1);
    if (m_out) memcpy(m_out, m, sizeof m);
}

/* ------------------------------------------------------------------ shading */

/* M3d_matrix_tools.c:238-255 */
static double dot3(const double a[3], const double b[3])
{
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}
static int norm3(double res[3], const double a[3])
{
    double mag = sqrt(dot3(a, a));
    if (mag == 0) return 0;
    res[0] = a[0] / mag; res[1] = a[1] / mag; res[2] = a[2] / mag;
    return 1;
}

/* the two helpers above as the light model uses them, exported so that the tests can hold them
 * against the reference's compiled M3d_dot_product / M3d_norm */
double orc_dot3(const double a[3], const double b[3]) { return dot3(a, b); }
int orc_norm3(double res[3], const double a[3]) { return norm3(res, a); }

#define ORC_AMBIENT 0.2    /* light_model.c:13 */
#define ORC_DIFFUSE 0.5    /* light_model.c:14 */
#define ORC_SPECPOW 50.0   /* light_model.c:15 */

/* light_model.c:17-122.  Returns 1 and fills rgb, or 0 (rgb untouched) on a zero-length
 * normal, L or V.  The shadow ray (:60-76) starts 0.001 along L and ends AT the light, with
 * no upper bound on t: an occluder beyond the light still shadows. */
static int light_model_single(const OrcScene* sc, const OrcFrame* f,
                              const double Ka[3], const double Kd[3], const double Ks[3],
                              const double s[3], const double p[3], const double n[3],
                              double rgb[3], int light_num, long* shadow_rays)
{
    /* the globals are doubles in the reference, so no expression here is a compile-time
     * constant there; volatile keeps gcc from folding (1.0 - DIFFUSE - AMBIENT) differently */
    volatile double AMBIENT = ORC_AMBIENT, DIFFUSE = ORC_DIFFUSE, SPECPOW = ORC_SPECPOW;
    double len;
    double N[3];
    (void)Ka;
    len = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);           /* :38 */
    if (len == 0) return 0;
    if (norm3(N, n) == 0) return 0;

    double L[3];
    const double* lp = f->light_in_eye_space[light_num];
    L[0] = lp[0] - p[0]; L[1] = lp[1] - p[1]; L[2] = lp[2] - p[2]; /* :43-45 */
    len = sqrt(L[0] * L[0] + L[1] * L[1] + L[2] * L[2]);
    if (len == 0) return 0;
    if (norm3(L, L) == 0) return 0;
    double NdotL = dot3(N, L);                                      /* :49 */

    if (len > f->light_max_dist[light_num]) {                       /* :51-56 */
        rgb[0] = rgb[1] = rgb[2] = 0;
        return 1;
    }
    double attenuation = 1 + f->light_attenuation[light_num][0] * len
                           + f->light_attenuation[light_num][1] * len * len;   /* :57 */

    double shadow_S[3];
    shadow_S[0] = p[0] + L[0] * 0.001;                              /* :61-63 */
    shadow_S[1] = p[1] + L[1] * 0.001;
    shadow_S[2] = p[2] + L[2] * 0.001;
    double uv[2], point[3];
    if (shadow_rays) (*shadow_rays)++;
    if (intersect_all_t(sc, shadow_S, lp, uv, point, NULL) != -1) { /* :70-76 */
        rgb[0] = rgb[1] = rgb[2] = 0;
        return 1;
    }

    double V[3];
    V[0] = s[0] - p[0]; V[1] = s[1] - p[1]; V[2] = s[2] - p[2];     /* :79-81 */
    if (norm3(V, V) == 0) return 0;

    double diffuse[3];
    int i;
    for (i = 0; i < 3; i++) {                                       /* :89-96 */
        diffuse[i] = NdotL * Kd[i] * DIFFUSE * f->light_rgb[light_num][i];
        if (diffuse[i] < 0) diffuse[i] = 0;
    }
    double R[3];
    R[0] = 2 * NdotL * N[0] - L[0];                                 /* :103-105 */
    R[1] = 2 * NdotL * N[1] - L[1];
    R[2] = 2 * NdotL * N[2] - L[2];
    double RdotV = dot3(R, V);
    double fs;
    if (RdotV > 0) fs = pow(RdotV, SPECPOW);                        /* :108-109 */
    else fs = 0;
    for (i = 0; i < 3; i++) {                                       /* :112-119 */
        double specular = fs * Ks[i] * (1.0 - DIFFUSE - AMBIENT) * f->light_rgb[light_num][i];
        rgb[i] = (diffuse[i] + specular) / attenuation;
    }
    return 1;
}

/* light_model.c:124-151 */
void orc_light_model(const OrcScene* sc, const OrcFrame* f,
                     const double Ka[3], const double Kd[3], const double Ks[3],
                     const double s[3], const double p[3], const double n[3],
                     double argb[3], long* shadow_rays)
{
    volatile double AMBIENT = ORC_AMBIENT;
    int i, j;
    for (j = 0; j < 3; j++) argb[j] = AMBIENT * Ka[j];              /* :134-139 */
    for (i = 0; i < f->num_lights; i++) {
        double rgb[3];
        if (light_model_single(sc, f, Ka, Kd, Ks, s, p, n, rgb, i, shadow_rays))
            for (j = 0; j < 3; j++) argb[j] += rgb[j];              /* :143-145 */
    }
    for (j = 0; j < 3; j++) if (argb[j] > 1) argb[j] = 1;           /* :147-149 */
}

/* light_model.c:156-186: barycentric blend (u with B, v with C) then the upper 3x3 of obinv
 * transposed, i.e. normal_j = sum_i Ntemp_i * obinv[i][j]. */
void orc_interpolate_normal(const OrcScene* sc, int tri, const double uv[2],
                            const double obinv[4][4], double normal[3])
{
    int a = sc->An[tri], b = sc->Bn[tri], c = sc->Cn[tri];
    double An[3] = {sc->xn[a], sc->yn[a], sc->zn[a]};
    double Bn[3] = {sc->xn[b], sc->yn[b], sc->zn[b]};
    double Cn[3] = {sc->xn[c], sc->yn[c], sc->zn[c]};
    double Nt[3];
    int k;
    for (k = 0; k < 3; k++)                                         /* :178-180 */
        Nt[k] = (1 - uv[0] - uv[1]) * An[k] + uv[0] * Bn[k] + uv[1] * Cn[k];
    for (k = 0; k < 3; k++)                                         /* :183-185 */
        normal[k] = Nt[0] * obinv[0][k] + Nt[1] * obinv[1][k] + Nt[2] * obinv[2][k];
}

/* graphic_tools.c:161-197 for a tightly packed 8-bit RGB image (3 bytes per texel, rows of
 * `w` texels): nearest texel at (int)(x*w), (int)((1-y)*h).  The reference does no bounds
 * handling (an index outside the surface is undefined behaviour there); the linear texel
 * index is clamped into the image here so the oracle is total. */
void orc_get_rgb(const OrcTexture* tex, const OrcScene* sc, int tri, const double uv[2],
                 double rgb[3])
{
    int ia = sc->At[tri], ib = sc->Bt[tri], ic = sc->Ct[tri];
    double At[2] = {sc->u[ia], sc->v[ia]};
    double Bt[2] = {sc->u[ib], sc->v[ib]};
    double Ct[2] = {sc->u[ic], sc->v[ic]};
    double width = (double)tex->w, height = (double)tex->h;
    double x = (1 - uv[0] - uv[1]) * At[0] + uv[0] * Bt[0] + uv[1] * Ct[0];   /* :179-180 */
    double y = (1 - uv[0] - uv[1]) * At[1] + uv[0] * Bt[1] + uv[1] * Ct[1];
    y = 1 - y;                                                      /* :183 */
    x *= width; y *= height;
    int xi = (int)x, yi = (int)y;                                   /* :187 */
    long idx = (long)yi * tex->w + xi;                              /* :129 with pitch = 3*w */
    long last = (long)tex->w * tex->h - 1;
    if (idx < 0) idx = 0;
    if (idx > last) idx = last;
    const uint8_t* px = tex->rgb + 3 * idx;
    rgb[0] = (px[0] * 1.0) / 255;                                   /* :194-196 */
    rgb[1] = (px[1] * 1.0) / 255;
    rgb[2] = (px[2] * 1.0) / 255;
}

/* driver.c:143-182: material selection for a hit triangle */
static void select_material(const OrcScene* sc, int tri, const double uv[2],
                            double Ka[3], double Kd[3], double Ks[3])
{
    int m = sc->mtl ? sc->mtl[tri] : -1;
    int k;
    if (m == -1) {                                                  /* :177-182 */
        for (k = 0; k < 3; k++) Ka[k] = Kd[k] = Ks[k] = 1.0;
        return;
    }
    const OrcMaterial* mat = &sc->mats[m];
    if (mat->map_Ka >= 0) orc_get_rgb(&sc->textures[mat->map_Ka], sc, tri, uv, Ka);
    else for (k = 0; k < 3; k++) Ka[k] = mat->Ka[k];
    if (mat->map_Kd >= 0) orc_get_rgb(&sc->textures[mat->map_Kd], sc, tri, uv, Kd);
    else for (k = 0; k < 3; k++) Kd[k] = mat->Kd[k];
    if (mat->map_Ks >= 0) orc_get_rgb(&sc->textures[mat->map_Ks], sc, tri, uv, Ks);
    else for (k = 0; k < 3; k++) Ks[k] = mat->Ks[k];
}

/* graphic_tools.c:106-113: c*255 then the implicit double -> Uint8 conversion of
 * SDL_SetRenderDrawColor's arguments, a truncation for the [0,255] values that occur. */
uint8_t orc_quantize(double c)
{
    c *= 255;
    return (uint8_t)c;
}

/* driver.c:126-193 for the pixels (x_pix, y_pix), y_pix = y0, y0+ystep, ... < y1.
 * Outputs are indexed [y_pix*W + x_pix]; rows not visited are left untouched.
 *   hit   : winning triangle index or -1            (driver.c:135)
 *   tuv   : t,u,v of the hit (3 doubles per pixel), untouched on a miss
 *   argb  : the three colour values handed to set_rgb (driver.c:189)
 *   rgb8  : their quantisation (graphic_tools.c:106-113), 3 bytes per pixel
 * Returns the number of shadow rays fired. */
long orc_render_rows(const OrcScene* sc, const OrcFrame* f, int W, int H, double tan_half,
                     int y0, int y1, int ystep, int shade, int nthreads,
                     int32_t* hit, double* tuv, double* argb, uint8_t* rgb8)
{
    long shadow_total = 0;
    if (ystep < 1) ystep = 1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
    const int nrows = (y1 > y0) ? (y1 - y0 + ystep - 1) / ystep : 0;
    const long total = (long)nrows * W;
    long k;
#pragma omp parallel for schedule(dynamic, 16) reduction(+ : shadow_total)
    for (k = 0; k < total; k++) {
        const int y_pix = y0 + (int)(k / W) * ystep;
        const int x_pix = (int)(k % W);
        double origin[3] = {0, 0, 0}, screen_pt[3];              /* :121-123 */
        double uv[2], point[3], normal[3], col[3], t = 0;
        long shadows = 0;
        size_t o = (size_t)y_pix * W + x_pix;
        screen_pt[0] = x_pix - W / 2;                            /* :131-133, integer W/2 */
        screen_pt[1] = y_pix - H / 2;
        screen_pt[2] = (W / 2) / tan_half;
        int s = intersect_all_t(sc, origin, screen_pt, uv, point, &t);
        if (hit) hit[o] = s;
        if (s != -1 && tuv) { tuv[3 * o] = t; tuv[3 * o + 1] = uv[0]; tuv[3 * o + 2] = uv[1]; }
        if (!shade) continue;
        if (s == -1) {
            col[0] = col[1] = col[2] = 0.5;                      /* :136-139 */
        } else {
            double Ka[3], Kd[3], Ks[3];
            orc_interpolate_normal(sc, s, uv, f->obinv, normal); /* :142 */
            select_material(sc, s, uv, Ka, Kd, Ks);
            orc_light_model(sc, f, Ka, Kd, Ks, origin, point, normal, col, &shadows);
        }
        shadow_total += shadows;
        if (argb) { argb[3 * o] = col[0]; argb[3 * o + 1] = col[1]; argb[3 * o + 2] = col[2]; }
        if (rgb8) {
            rgb8[3 * o] = orc_quantize(col[0]);
            rgb8[3 * o + 1] = orc_quantize(col[1]);
            rgb8[3 * o + 2] = orc_quantize(col[2]);
        }
    }
    return shadow_total;
}

/* driver.c:31-32,186-190 + graphic_tools.c:115-121: the window is cleared to (0,0,0), pixel
 * (x_pix, y_pix) is drawn at screen row H - y_pix, so y_pix = 0 falls off-screen and screen
 * row 0 keeps the clear colour.  rgb8 is indexed [y_pix*W + x_pix]; screen is W*H*3 bytes,
 * top row first. */
void orc_place_pixels(const uint8_t* rgb8, int W, int H, uint8_t* screen)
{
    int y_pix;
    memset(screen, 0, (size_t)W * H * 3);
    for (y_pix = 1; y_pix < H; y_pix++)
        memcpy(screen + (size_t)(H - y_pix) * W * 3, rgb8 + (size_t)y_pix * W * 3, (size_t)W * 3);
}

/* n independent rays through orc_scene_intersect (OpenMP over rays): S, E as 3 doubles per
 * ray; hit[i] = index or -1; tuv (3 per ray) and point (3 per ray) written on a hit. */
void orc_trace_rays(const OrcScene* sc, long n, const double* S, const double* E,
                    int32_t* hit, double* tuv, double* point, int nthreads)
{
    long i;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
#pragma omp parallel for schedule(dynamic, 64)
    for (i = 0; i < n; i++) {
        double uv[2] = {0, 0}, p[3] = {0, 0, 0}, t = 0;
        int s = intersect_all_t(sc, S + 3 * i, E + 3 * i, uv, p, &t);
        hit[i] = s;
        if (s != -1) {
            if (tuv) { tuv[3 * i] = t; tuv[3 * i + 1] = uv[0]; tuv[3 * i + 2] = uv[1]; }
            if (point) { point[3 * i] = p[0]; point[3 * i + 1] = p[1]; point[3 * i + 2] = p[2]; }
        }
    }
}

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
