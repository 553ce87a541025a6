/* rt_oracle.h -- interface of the CPU oracle (test infrastructure only; see rt_oracle.c). */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_LIGHTS 100   /* light_model.c:9-12 */

/* 8-bit RGB image, 3 bytes per texel, rows of w texels, top row first. */
typedef struct OrcTexture {
    int w, h;
    const uint8_t* rgb;
} OrcTexture;

/* The fields of the reference's Material that the pixel loop reads (obj_reader.h:7-23,
 * driver.c:143-175); map_* are indices into OrcScene.textures, -1 = no map. */
typedef struct OrcMaterial {
    double Ka[3], Kd[3], Ks[3];
    int map_Ka, map_Kd, map_Ks;
} OrcMaterial;

/* Scene in structure-of-arrays form.  Vertex, normal and texture-coordinate arrays are
 * 1-indexed like the reference's globals (obj_reader.h:41-46); x,y,z hold whatever space
 * the caller has transformed them into (eye space inside the pixel loop). */
typedef struct OrcScene {
    int num_tris;
    const int *A, *B, *C;        /* vertex indices         (Triangle.A/B/C)    */
    const int *An, *Bn, *Cn;     /* normal indices         (Triangle.An/Bn/Cn) */
    const int *At, *Bt, *Ct;     /* texture-vertex indices (Triangle.At/Bt/Ct), may be NULL */
    const int *mtl;              /* material index per triangle, -1 = none; NULL = all -1 */
    const double *x, *y, *z;
    const double *xn, *yn, *zn;
    const double *u, *v;         /* may be NULL when no material has a map */
    int num_mats;
    const OrcMaterial* mats;
    int num_textures;
    const OrcTexture* textures;
} OrcScene;

/* Per-frame state of driver.c:56-107 and the light_model.h globals. */
typedef struct OrcFrame {
    int num_lights;
    double light_in_eye_space[ORC_MAX_LIGHTS][3];
    double light_rgb[ORC_MAX_LIGHTS][3];
    double light_attenuation[ORC_MAX_LIGHTS][2];
    double light_max_dist[ORC_MAX_LIGHTS];
    double obmat[4][4];
    double obinv[4][4];
} OrcFrame;

double orc_det_3x3(const double A[3], const double B[3], const double C[3]);
double orc_intersect_single_triangle(const double S[3], const double E[3],
                                     const double A[3], const double B[3], const double C[3],
                                     double uv[2]);
int orc_intersect_all_triangles(const double S[3], const double E[3],
                                double uv[2], double point[3],
                                const int* iA, const int* iB, const int* iC,
                                const double* x, const double* y, const double* z,
                                int num_tris);
int orc_scene_intersect(const OrcScene* sc, const double S[3], const double E[3],
                        double uv[2], double point[3], double* t_out);
void orc_mat_mult_points(double* X, double* Y, double* Z, const double m[4][4],
                         const double* x, const double* y, const double* z, int numpoints);
void orc_mat_mult(double res[4][4], const double a[4][4], const double b[4][4]);
void orc_mat_mult_pt(double P[3], const double m[4][4], const double Q[3]);
int orc_movement_sequence(double mat[4][4], double inv[4][4], int n,
                          const int* types, const double* params);
int orc_view(double view[4][4], double view_inverse[4][4],
             const double eye[3], const double coi[3], const double up[3]);
void orc_driver_frame_setup(int frame_number, int frames_per_turn, OrcFrame* f);
void orc_center_and_scale(double* x, double* y, double* z, int num_v, double m_out[4][4]);
void orc_light_model(const OrcScene* sc, const OrcFrame* f,
                     const double Ka[3], const double Kd[3], const double Ks[3],
                     const double s[3], const double p[3], const double n[3],
                     double argb[3], long* shadow_rays);
void orc_interpolate_normal(const OrcScene* sc, int tri, const double uv[2],
                            const double obinv[4][4], double normal[3]);
void orc_get_rgb(const OrcTexture* tex, const OrcScene* sc, int tri, const double uv[2],
                 double rgb[3]);
double orc_dot3(const double a[3], const double b[3]);        /* M3d_matrix_tools.c:238-242 */
int orc_norm3(double res[3], const double a[3]);              /* M3d_matrix_tools.c:244-255 */
uint8_t orc_quantize(double c);
long orc_render_rows(const OrcScene* sc, const OrcFrame* f, int W, int H, double tan_half,
                     int y0, int y1, int ystep, int shade, int nthreads,
                     int32_t* hit, double* tuv, double* argb, uint8_t* rgb8);
void orc_place_pixels(const uint8_t* rgb8, int W, int H, uint8_t* screen);
void orc_trace_rays(const OrcScene* sc, long n, const double* S, const double* E,
                    int32_t* hit, double* tuv, double* point, int nthreads);
int orc_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
