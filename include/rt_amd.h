/*
 * rt_amd.h -- C ABI of librt_amd.so, the MI355X (gfx950) implementation of the
 * ray-triangle hot path of SimonBuan/Raytracer.
 *
 * Two groups of entry points:
 *
 *  1. The reference's own device API, unchanged (kernel.h:10-17): mat_mult_device and
 *     intersect_all_triangles_device, host pointers in and out, synchronous, no error
 *     channel.  A program built against the reference's kernel.h links against this library
 *     instead of mat_mult.cu / triangle_intersection.cu with no source change.
 *
 *  2. An additive batched API (rt_*) that replaces the reference's host-serial pixel loop
 *     (driver.c:126-193) by whole-band launches over a device-resident scene.  Every rt_*
 *     function returns 0 on success or a negative RT_E_* code; rt_last_error() gives the
 *     text.  Nothing here aborts the process.
 *
 * Plain pointers and sizes only.  "host" / "device" in a parameter comment says which
 * address space the pointer must belong to.  All calls are blocking unless stated, and the
 * library is single-caller like the reference (one host thread per process and GPU).
 */
#ifndef RT_AMD_H
#define RT_AMD_H

#include <stdint.h>
#include "rt_reference_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ status codes */
#define RT_OK            0
#define RT_E_ARG        -1   /* bad argument (null pointer, negative size, index out of range) */
#define RT_E_HIP        -2   /* a HIP runtime call failed; see rt_last_error()                 */
#define RT_E_STATE      -3   /* call order (e.g. tracing a scene that was never transformed)   */
#define RT_E_IO         -4   /* file could not be opened / parsed                              */
#define RT_E_NODEVICE   -5   /* no gfx950 device visible                                       */
#define RT_E_LIMIT      -6   /* a documented size limit was exceeded                           */

#define RT_MAX_LIGHTS 100    /* light_model.c:9-12 */

/* ------------------------------------------------------------------ 1. reference API */

/* kernel.h:10-14 / mat_mult.cu:31-78.  (X,Y,Z)[i] = rows 0..2 of m applied to (x,y,z,1)[i]
 * for i in [0,numpoints), fp64, left-to-right sums; outputs may alias inputs.  Host pointers. */
void mat_mult_device(double* X, double* Y, double* Z, double m[4][4],
                     double* x, double* y, double* z, int numpoints);

/* kernel.h:16-17 / triangle_intersection.cu:98-190.  One ray S->E against num_tris
 * triangles.  Returns the index of the triangle with the smallest t > 0 (lowest index on a
 * tie) or -1; on a hit uv[] and point[] are written, on a miss they are left alone.  Host
 * pointers.  The device copy of the mesh is kept between calls and refreshed when
 * mat_mult_device rewrites x/y/z or rt_legacy_invalidate() is called. */
int intersect_all_triangles_device(double S[3], double E[3], double uv[2], double point[3],
                                   Triangle* tris, double* x, double* y, double* z,
                                   int num_tris);

/* Drop the cached device copy used by intersect_all_triangles_device (call after editing
 * tris or x/y/z by any means other than mat_mult_device). */
void rt_legacy_invalidate(void);
/* sizeof(Triangle) of the caller's build when it differs from this header's (0 = default). */
void rt_set_triangle_stride(int bytes);

/* ------------------------------------------------------------------ 2. batched API */

int         rt_init(int device_ordinal);         /* selects the GPU for this process */
void        rt_shutdown(void);
int         rt_device_count(void);               /* number of visible HIP devices (0 if none) */
const char* rt_last_error(void);
const char* rt_version(void);
int         rt_set_stream(void* hip_stream);     /* NULL = the default stream; waits for the work left on the stream it replaces */

typedef struct rt_scene rt_scene;                /* device-resident mesh, opaque */
typedef struct rt_obj rt_obj;                    /* a loaded .obj file (host side), opaque */

/* Lights and matrices of one frame: the light_model.h:4-8 globals plus the obinv of
 * driver.c:107 that interpolate_normal_vector consumes. */
typedef struct rt_frame {
    int    num_lights;
    double light_in_eye_space[RT_MAX_LIGHTS][3];
    double light_rgb[RT_MAX_LIGHTS][3];
    double light_attenuation[RT_MAX_LIGHTS][2];
    double light_max_dist[RT_MAX_LIGHTS];
    double obmat[4][4];      /* object -> eye  (driver.c:106) */
    double obinv[4][4];      /* eye -> object  (driver.c:107) */
} rt_frame;

int rt_scene_create(rt_scene** out);
int rt_scene_destroy(rt_scene* sc);

/* Geometry upload (host pointers).  Arrays are 1-indexed like obj_reader.h:41-46: length
 * count+1, slot 0 ignored.  Vertex, normal and texture indices must lie in [1, count], material
 * indices in [-1, num_mats), texture numbers in [-1, num_textures): the setters check what they
 * can see and rt_render_rows* checks the tables against each other (RT_E_STATE), so that no
 * kernel reads out of bounds.  A material with a texture needs At/Bt/Ct and texture
 * coordinates. */
int rt_scene_set_vertices(rt_scene* sc, int num_v, const double* x, const double* y, const double* z);
int rt_scene_set_normals(rt_scene* sc, int num_vn, const double* xn, const double* yn, const double* zn);
int rt_scene_set_texcoords(rt_scene* sc, int num_vt, const double* u, const double* v);
/* At/Bt/Ct and mtl may be NULL (no texture coordinates / every triangle mtl.index = -1). */
int rt_scene_set_triangles(rt_scene* sc, int num_tris,
                           const int* A, const int* B, const int* C,
                           const int* An, const int* Bn, const int* Cn,
                           const int* At, const int* Bt, const int* Ct,
                           const int* mtl);
/* Same from the reference's array-of-structures (obj_reader.h:25-39); the per-triangle
 * Material copies are folded back into a table keyed by mtl.index. */
int rt_scene_set_triangles_aos(rt_scene* sc, const Triangle* tris, int num_tris);
/* Ka/Kd/Ks: 3*num_mats doubles each.  map_*: texture index per material or -1; may be NULL. */
int rt_scene_set_materials(rt_scene* sc, int num_mats,
                           const double* Ka, const double* Kd, const double* Ks,
                           const int* map_Ka, const int* map_Kd, const int* map_Ks);
/* Textures as tightly packed 8-bit RGB, top row first (what get_rgb reads through
 * SDL_GetRGB, graphic_tools.c:161-197). */
int rt_scene_set_textures(rt_scene* sc, int num_textures, const int* w, const int* h,
                          const uint8_t* const* rgb);

/* mat_mult.cu:8-28 on the device-resident master copy: eye = m * object for every vertex
 * (slot 0 included, as driver.c:110 does), then the per-triangle fp32 sweep buffer
 * (v0, e1, e2 planes) is rebuilt.  The object-space master is never overwritten, so no
 * inverse pass (driver.c:202) is needed and no drift accumulates over frames.
 * The call queues the work on the library's stream and returns (m is read before it does); the
 * calls that use the eye-space geometry are ordered behind it, and a failure of the queued work
 * is reported by the next call that waits for the device. */
int rt_scene_transform(rt_scene* sc, const double m[4][4]);
/* Applies m to the master copy itself (what obj_reader.c:288 does at load time). */
int rt_scene_bake(rt_scene* sc, const double m[4][4]);
/* Copies the current eye-space vertices back (host, num_v+1 each). */
int rt_scene_get_eye_vertices(rt_scene* sc, double* x, double* y, double* z);
int rt_scene_num_tris(const rt_scene* sc);

/* Primary rays of driver.c:121-135 for the pixel rows y_pix in [y0,y1): S = 0,
 * E = (x_pix - W/2, y_pix - H/2, (W/2)/tan_half) with integer W/2, H/2.
 * Outputs (device), indexed [(y_pix - y0)*W + x_pix]:
 *   hit : int32 winning triangle or -1
 *   tuv : 3 doubles per pixel (t,u,v), written only where hit != -1; may be NULL */
int rt_trace_primary(rt_scene* sc, int W, int H, double tan_half, int y0, int y1,
                     int32_t* hit, double* tuv);

/* n arbitrary rays, S and E as 3 doubles per ray (device).  Same outputs, plus point
 * (3 doubles per ray, written on a hit; may be NULL).  intersect_all_triangles_device
 * semantics per ray. */
int rt_trace_rays(rt_scene* sc, int n, const double* S, const double* E,
                  int32_t* hit, double* tuv, double* point);

/* rt_trace_rays for callers that cannot allocate device memory (tight serving loops, harnesses
 * that forbid hipMalloc, code that runs between the captures of HIP graphs; the call itself is
 * blocking -- it ends with a stream synchronisation -- so it cannot run on a capturing stream).
 * No counterpart in the reference, whose
 * intersect_all_triangles_device allocates and frees on every call
 * (triangle_intersection.cu:104-165).  Same arguments, outputs and results as rt_trace_rays, but
 * the per-ray workspace (candidate counts and lists, overflow queue, counters: 72 bytes per ray)
 * is a static device array of RT_STATIC_MAX_RAYS entries (about 4.8 MB, held by every GPU the
 * library is initialised on until the process ends) instead of the scene's grow-only buffers:
 * the call performs no hipMalloc / hipFree and no profiling read-back.  n > RT_STATIC_MAX_RAYS
 * returns RT_E_LIMIT.  The array used is the copy on the scene's own device, which must be the
 * current device (RT_E_STATE otherwise, also when rt_init was never run on it).  All calls on one GPU share that array: calls that overlap on a GPU
 * (several host threads or streams) race; calls in sequence do not (each call clears what it
 * needs zeroed and rewrites the rest). */
#define RT_STATIC_MAX_RAYS 65536
int rt_trace_rays_static(rt_scene* sc, int n, const double* S, const double* E,
                         int32_t* hit, double* tuv, double* point);

/* occ[i] = 1 if ray i hits any triangle at t > 0 (what light_model.c:70 asks), else 0. */
int rt_occluded(rt_scene* sc, int n, const double* S, const double* E, uint8_t* occ);

/* The whole pixel loop of driver.c:126-193 for rows [y0,y1): primary sweep, normal
 * interpolation, material selection, Light_Model with its shadow rays, clamp.
 * Outputs (device, any may be NULL), indexed [(y_pix - y0)*W + x_pix]:
 *   hit  : int32      winning triangle or -1
 *   tuv  : 3 doubles  t,u,v on a hit
 *   argb : 3 doubles  the colour handed to set_rgb (driver.c:189)
 *   rgb8 : 3 bytes    (Uint8)(c*255), graphic_tools.c:106-113 */
int rt_render_rows(rt_scene* sc, const rt_frame* f, int W, int H, double tan_half,
                   int y0, int y1, int32_t* hit, double* tuv, double* argb, uint8_t* rgb8);

/* driver.c:186-190: rows [y0,y1) of rgb8 (as produced above) go to screen rows H - y_pix
 * of a W*H*3 image whose row 0 keeps the clear colour (0,0,0) of driver.c:31-32; y_pix = 0
 * falls off-screen.  Both pointers device.  The caller zeroes `screen` once per frame. */
int rt_place_rows(const uint8_t* rgb8, int W, int H, int y0, int y1, uint8_t* screen);

/* Row band of rank r out of n: contiguous, H / n rows each, the first H % n ranks one more. */
int rt_band(int H, int nranks, int rank, int* y0, int* y1);

/* A row set that interleaves parts: the rows [y0,y1) are cut into blocks of `block` rows and the
 * blocks are dealt to the parts in rounds of `parts`, forwards (block round*parts + part) or
 * backwards (block round*parts + parts-1-part) after the Thue-Morse sequence F B B F B F F B ...
 * (backwards when the round number has an odd number of one bits), so that a cost that grows
 * down the image, steadily or faster, is shared evenly.  The rows left after the last complete
 * round (fewer than block*parts) are not dealt as whole blocks, which would leave some parts one
 * block short: they are cut into `parts` runs of nearly equal length (run s starts at row
 * tail*s/parts of them), and the part takes the run of its slot in that round.  The rows of a part are numbered
 * locally 0 .. rt_rows_count()-1 in ascending image order, and the outputs of the *_part calls
 * are indexed [local_row * W + x_pix].  parts = 1 is the contiguous band.  Interleaving is how
 * the pixel loop of driver.c:126-193 is split over GPUs when the cost per row is uneven (the
 * shadow rays of light_model.c:70 are): every part sees every region of the image. */
typedef struct rt_rows {
    int y0, y1;      /* image rows the set is cut from */
    int block;       /* rows per block (16 matches the sweeps' pixel tiles) */
    int parts, part;
} rt_rows;
int rt_rows_count(const rt_rows* rows);                    /* rows in the part, or RT_E_ARG */
int rt_rows_image_row(const rt_rows* rows, int local_row); /* image row of a local row, or RT_E_ARG */
int rt_trace_primary_part(rt_scene* sc, int W, int H, double tan_half, const rt_rows* rows,
                          int32_t* hit, double* tuv);
int rt_render_rows_part(rt_scene* sc, const rt_frame* f, int W, int H, double tan_half,
                        const rt_rows* rows, int32_t* hit, double* tuv, double* argb, uint8_t* rgb8);
int rt_place_rows_part(const uint8_t* rgb8, int W, int H, const rt_rows* rows, uint8_t* screen);

/* One frame on several GPUs of one node, from one process.  The reference drives a single GPU
 * (triangle_intersection.cu:98-190); its pixel loop (driver.c:126-193) shards by rows because
 * pixels are independent.  An rt_multi holds one scene per device (every device keeps the full
 * mesh), renders part i of n_devices interleaved 16-row parts on device i (one host thread per
 * device inside the call; the interface stays blocking and single-caller), moves the finished
 * 8-bit rows to the first device over RCCL (one group of ncclSend / ncclRecv, xGMI point to
 * point) and places them there (driver.c:186-190).  RCCL is loaded at run time, and only for more
 * than one device; where it cannot be loaded or initialised the rows travel by peer copies
 * instead (rt_multi_exchange tells which, and why).  Not usable while a user stream is set
 * (rt_set_stream).
 *   devices        n_devices HIP ordinals, or NULL for 0 .. n_devices-1
 *   for_each_scene calls fn(scene of part i, i, user) with that scene's device current, one after
 *                  the other, for the rt_scene_set_* uploads; rt_multi_set_obj does it for a loaded
 *                  OBJ file
 *   transform      rt_scene_transform(obmat) on every device (driver.c:110)
 *   render         screen: W*H*3 bytes on the FIRST device, top row first; cleared and filled */
typedef struct rt_multi rt_multi;
int rt_multi_create(rt_multi** out, int n_devices, const int* devices);
int rt_multi_destroy(rt_multi* m);
int rt_multi_num_devices(const rt_multi* m);
const char* rt_multi_exchange(const rt_multi* m);   /* "rccl", "copy" or "copy (<why RCCL is not used>)" */
int rt_multi_for_each_scene(rt_multi* m, int (*fn)(rt_scene* scene, int part, void* user), void* user);
int rt_multi_set_obj(rt_multi* m, const rt_obj* o);
int rt_multi_transform(rt_multi* m, const double obmat[4][4]);
int rt_render_frame_multi(rt_multi* m, const rt_frame* f, int W, int H, double tan_half,
                          uint8_t* screen);

/* The frame loop of driver.c:28-203 as one pipelined call.  Frames first_frame,
 * first_frame + frame_step, ... (count of them) of the orbit of rt_driver_frame_setup(frame,
 * frames_per_turn): transform, render every row, place.  With bmp_pattern != NULL (a printf
 * pattern with one %d, e.g. "pic/pic%04d.bmp" as driver.c:197) each picture is copied to the
 * host on a side stream and written by background threads while the next frames render.
 * frame_step > 1 is how ranks split a sequence without any exchange (rank r of n:
 * first_frame + r, frame_step = n). */
typedef struct rt_sequence_stats {
    int    frames;                       /* frames rendered */
    double seconds_total;                /* wall clock of the call, files closed */
    double seconds_render;               /* of which: set-up + transform + pixel loop + placement */
    double seconds_waiting_for_writers;  /* of which: the GPU side stalled on a full ring */
    double seconds_writing;              /* summed over the writer threads (overlapped) */
} rt_sequence_stats;
int rt_render_sequence(rt_scene* sc, int first_frame, int count, int frame_step, int frames_per_turn,
                       int W, int H, double tan_half, const char* bmp_pattern,
                       rt_sequence_stats* stats);

/* ------------------------------------------------------------------ measurement */
typedef struct rt_profile {
    double   sweep_ms;        /* sum of HIP-event durations of closest-hit sweep launches */
    uint64_t sweep_launches;
    uint64_t sweep_tests;     /* ray-triangle pairs those launches evaluated */
    double   shadow_ms;       /* same for any-hit sweeps */
    uint64_t shadow_launches;
    uint64_t shadow_tests;    /* pairs actually evaluated (early-out accounted per tile) */
    uint64_t rays_primary;
    uint64_t rays_shadow;
    uint64_t exact_candidates;  /* fp64 confirmations */
    uint64_t exact_overflow;    /* rays that fell back to the full fp64 sweep */
    uint64_t rays_live;         /* generic-ray launches: rays in use (unused slots excluded) */
    /* any-hit phases that ran as a sweep over lattices of direction cells (their time, launches and
     * ray-triangle pairs are part of shadow_ms / shadow_launches / shadow_tests as well) */
    double   cell_ms;
    uint64_t cell_launches;
    uint64_t cell_pairs;        /* (cell, triangle) pairs: one first test each, what the hot loop executes */
    uint64_t cell_ray_pairs;    /* (ray, triangle) pairs behind it, tested one by one */
    uint64_t cell_rays;         /* rays swept that way */
    uint64_t cells_per_lane;    /* kernel shape of the most recent such launch */
    uint64_t cell_lookups;      /* cell-table look-ups of the rare path */
    uint64_t cell_fine;         /* fine cells (the grid the rays are sorted and looked up by) per cell, most recent launch */
    uint64_t cell_entries;      /* (wave footprint, triangle) pairs that went on to the rare path */
} rt_profile;
/* Compile-time shape of the sweep kernel for rays in general position (rt_trace_rays,
 * rt_occluded): formulation (2 = moment form), triangles per LDS tile, rays per lane, and the
 * fp32 flops the hot loop executes per ray-triangle pair (FMA = 2). */
int rt_sweep_config(int* form, int* tile_tris, int* rays_per_lane, int* flop_per_pair);
/* Same for the two sweeps of the pixel loop (rt_trace_primary, rt_render_rows): the
 * pixel-grid sweep of the primary rays and the per-light any-hit sweep of the shadow rays. */
int rt_apex_sweep_config(int* tile_tris, int* primary_pixels_per_lane,
                         double* primary_flop_per_pair, int* shadow_rays_per_lane,
                         double* shadow_flop_per_pair, int* primary_chunks);
/* VALU instructions of the pixel-grid sweep's hot loop per lane and triangle, i.e. per
 * *pixels_per_lane pixel-triangle pairs (of the most recent primary sweep): fp32 FMAs (none),
 * fp32 additions / subtractions, three-way minima (reported as `logic`) and compares.  The loop
 * is bound by instruction issue, not by flops: its arithmetic is additions that step the
 * distance from the centre of a triangle's bounding box from one pixel of the lane to the
 * next, and minima that collect them. */
int rt_raster_sweep_mix(int* pixels_per_lane, int* fma, int* add, int* logic, int* compare);
/* Whether the most recent primary sweep ran the form of the kernel that steps from row to row by a
 * constant (every aligned group of 16 rows of the call's row set consecutive image rows: whole
 * bands, and interleaved parts whose block height is a multiple of 16) or the one that holds a
 * coordinate per row. */
int rt_raster_sweep_form(int* regular_rows);
/* The launch shape the primary sweep would take for a W x rows band over `triangles` triangles
 * (host logic only, no device work): pixels per lane (4096 NL and more = the wave-block form with
 * blocks of 4096 x 64 NL pixels, NL = 1, 2, 4, 8, 16, 36; up to 2048 = the lattice form), the
 * triangle chunks, the wave layout (-1 = a workgroup per block, 0 = waves dealt row by row, 1 / 2 /
 * 4 = waves side by side), and the workgroups of pixels that shape needs.  regular_rows: whether
 * the band's rows are regular (rt_raster_sweep_form).  Honours RT_AMD_RASTER_FORCE like the sweep. */
int rt_raster_shape_query(int W, int rows, int triangles, int regular_rows, int* pixels_per_lane,
                          int* chunks, int* waves_x, long* pixel_workgroups);
/* Tile culling (off by default).  The first test of every (ray, triangle) pair is the ray against
 * the bounding box of the triangle's projection (on the image plane for primary rays, on the
 * plane of the dominant axis through the light for shadow rays).  With culling on, every
 * triangle's box is first compared once with a wave's whole bundle -- the pixel rectangle of a
 * primary-ray wave, the direction box of a shadow-ray wave -- and only the survivors go through
 * the per-ray tests.  Every wave still visits every triangle and the results are bit for bit the
 * same (the candidates handed to the fp64 pass are the same); what changes is that most pairs
 * are decided a few thousand at a time.  The default keeps the all-pairs sweep of
 * triangle_intersection.cu:167-176, which is what the throughput figures in tests per second
 * refer to.  on: 1 / 0, or -1 to follow the environment variable RT_AMD_CULL again. */
int rt_set_tile_culling(int on);
int rt_get_tile_culling(void);
int rt_profile_enable(int on);     /* wraps sweep launches in HIP events and reads the device
                                      counters back (adds a sync per launch); the ray and
                                      candidate counts above are only collected while on */
int rt_profile_reset(void);
int rt_profile_get(rt_profile* out);

/* ------------------------------------------------------------------ host side (no GPU needed) */

/* driver.c:34-107 for one frame: eye orbit of radius 15 at height 10, the two lights,
 * RX(270 deg) object matrix. */
int rt_driver_frame_setup(int frame_number, int frames_per_turn, rt_frame* out);
/* M3d_matrix_tools.c:591-628, :305-471, :133-157, :163-178 */
int rt_view(double view[4][4], double view_inverse[4][4],
            const double eye[3], const double coi[3], const double up[3]);
int rt_movement_sequence(double mat[4][4], double inv[4][4], int n,
                         const int* types, const double* params);
int rt_mat_mult(double res[4][4], const double a[4][4], const double b[4][4]);
int rt_mat_mult_pt(double P[3], const double m[4][4], const double Q[3]);

/* Wavefront .obj/.mtl loader with the reference's semantics (obj_reader.c:27-237,457-726):
 * triangles and quads, four face formats, quads split (a,b,c)+(a,c,d), face normals
 * synthesised when the file has none, centre + scale to a 15-unit box.  The result is
 * structure-of-arrays in page-locked host memory when a GPU is present. */
int rt_obj_load(const char* path, rt_obj** out);
int rt_obj_free(rt_obj* o);
typedef struct rt_obj_view {
    int num_v, num_vn, num_vt, num_tris, num_mats;
    const double *x, *y, *z;          /* num_v+1  */
    const double *xn, *yn, *zn;       /* num_vn+1 */
    const double *u, *v;              /* num_vt+1 */
    const int *A, *B, *C, *An, *Bn, *Cn, *At, *Bt, *Ct, *mtl;   /* num_tris */
    const double *Ka, *Kd, *Ks;       /* 3*num_mats */
    const int *map_Ka, *map_Kd, *map_Ks;                        /* num_mats, -1 = none */
    int pinned;                       /* 1 if the arrays are page-locked */
} rt_obj_view;
int rt_obj_get(const rt_obj* o, rt_obj_view* out);
/* The image maps the .mtl files named (map_Ka / map_Kd / map_Ks, obj_reader.c:352-381), decoded
 * to tightly packed 8-bit RGB, top row first: PNG (all colour types and bit depths, interlaced
 * or not), baseline JPEG (8-bit, grey or YCbCr 4:4:4 / 4:2:2 / 4:2:0, decoded like libjpeg does by
 * default), binary PPM, uncompressed 24/32-bit BMP.  A file in any other format (progressive
 * JPEG, ...) or one that fails to decode leaves the material's map at -1, as a failed IMG_Load
 * leaves it NULL.  The pixels belong to the rt_obj. */
int rt_obj_num_textures(const rt_obj* o);
int rt_obj_texture(const rt_obj* o, int index, int* w, int* h, const uint8_t** rgb);
int rt_scene_set_obj(rt_scene* sc, const rt_obj* o);

/* graphic_tools.c:19-23,115-121: 32-bit BMP with the ARGB8888 masks, alpha 0xFF, from a
 * W*H*3 RGB screen image (host), top row first. */
int rt_save_bmp(const char* path, const uint8_t* screen_rgb, int W, int H);

#ifdef __cplusplus
}
#endif
#endif /* RT_AMD_H */
