// rt_internal.h -- shared declarations of the library's translation units (not installed).
#pragma once

#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <string>
#include <vector>

#include "rt_amd.h"

// ---------------------------------------------------------------- error plumbing
void rt_set_error(const char* fmt, ...);

#define RT_HIP(call)                                                                    \
    do {                                                                                \
        hipError_t rt_e_ = (call);                                                      \
        if (rt_e_ != hipSuccess) {                                                      \
            rt_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #call,             \
                         hipGetErrorString(rt_e_));                                     \
            return RT_E_HIP;                                                            \
        }                                                                               \
    } while (0)

#define RT_TRY(expr)                        \
    do {                                    \
        int rt_rc_ = (expr);                \
        if (rt_rc_ != RT_OK) return rt_rc_; \
    } while (0)

#define RT_ARG(cond, msg)                                                               \
    do {                                                                                \
        if (!(cond)) {                                                                  \
            rt_set_error("%s: %s", __func__, msg);                                      \
            return RT_E_ARG;                                                            \
        }                                                                               \
    } while (0)

// ---------------------------------------------------------------- tunables
constexpr int RT_WG = 256;          // threads per workgroup of every kernel here (4 waves)
#ifndef RT_TILE_TRIS
#define RT_TILE_TRIS 128
#endif
#ifndef RT_RAYS_PER_LANE
#define RT_RAYS_PER_LANE 4  // rays per lane of the general-position sweep
#endif
#ifndef RT_MIN_WAVES
#define RT_MIN_WAVES 1      // __launch_bounds__ waves per SIMD the sweeps' register allocation must allow
#endif
// general-position sweep (rt_sweep.hip, moment form): float4 planes per triangle in the sweep
// buffer and in each LDS tile, and the fp32 flops its hot loop executes per pair before the
// lazily evaluated parts (3 mul + 9 FMA)
constexpr int RT_PLANES = 5;
constexpr int RT_FLOP_PER_PAIR = 21;
constexpr int RT_TILE = RT_TILE_TRIS;      // triangles per LDS tile of the sweep
constexpr int RT_RPL = RT_RAYS_PER_LANE;   // rays per lane in the sweep
constexpr int RT_LIST_K = 16;       // candidate slots per ray handed to the fp64 confirmation
constexpr int RT_RASTER_LIST_K = 32;    // same for a pixel of the primary sweep, whose triangle chunks share one list
constexpr int RT_TILE_W = 16 * RT_RPL;   // pixel tile of one workgroup: 2x2 waves of (8*RPL)x8
constexpr int RT_TILE_H = 16;
constexpr int RT_RAYS_PER_WG = RT_WG * RT_RPL;

// Sweeps of rays through a common point (rt_sweep_apex.hip): the eye for primary rays, a
// light for its shadow rays.  4 float4 planes per triangle, packed per apex by k_pack_apex.
#ifndef RT_RASTER_RPL
#define RT_RASTER_RPL 2048  // most pixels per lane of the primary (pixel-grid) sweep
#endif
#define RT_RASTER_COLS 4     // pixel columns a lane's pixels are apart (a wave is COLS R x 64/COLS pixels)
#ifndef RT_APEX_RPL
#define RT_APEX_RPL 16      // most rays per lane of the any-hit apex sweep (8, 4, 2, 1 for smaller batches)
#endif
constexpr int RT_APEX_PLANES = 4;
// a light's plane set: the four apex planes and the bounding squares for the three dominant axes
constexpr int RT_LIGHT_PLANES = RT_APEX_PLANES + 3;
constexpr int RT_APEX_LIGHTS = 4;   // lights swept per any-hit launch (plane sets kept side by side)
constexpr int RT_APEX_SETS = 3 * RT_APEX_LIGHTS;    // ray sets per launch: light x dominant axis
// Wave-block form of the pixel-grid sweep (regular rows, no tile culling): a workgroup's block is 4096
// columns by 64 NL rows, NL = 1, 2, 4, 8, 16 or 36, and "pixels per lane" = 4096 NL names the shape.
constexpr int RT_WAVE_BLOCK_RPL = 4096;
constexpr bool rt_wave_block_shape(int rpl)
{
    return rpl == 4096 || rpl == 8192 || rpl == 16384 || rpl == 32768 || rpl == 65536 || rpl == 36 * 4096;
}
// A lane's rpl pixels in the lattice form are a block of this many rows (by rpl / rows columns); in the
// wave-block form: the rows of the block.
constexpr int rt_raster_lane_rows(int rpl) { return rpl >= RT_WAVE_BLOCK_RPL ? rpl / 64 : (rpl >= 2048 ? 32 : (rpl >= 256 ? 16 : (rpl >= 64 ? 8 : (rpl >= 16 ? 4 : (rpl >= 4 ? 2 : 1))))); }
// A wave's 64 lanes cover this many local rows (two 16-row blocks with 2048 pixels per lane) ...
constexpr int rt_raster_wave_rows(int rpl) { return rpl >= RT_WAVE_BLOCK_RPL ? rpl / 64 : (rpl >= 2048 ? 2 * RT_TILE_H : RT_TILE_H); }
// most pixels per lane in the regular-row forms (the wave-block form's tallest block)
constexpr int RT_RASTER_RPL_REGULAR = 36 * 4096;
// ... by this many pixel columns
constexpr int rt_raster_wave_cols(int rpl) { return 64 * rpl / rt_raster_wave_rows(rpl); }
// Hot loop of the pixel-grid sweep, per lane and triangle (rpl pixels = rows x columns): the
// distance of the first column from the box's centre and one addition for each further column,
// one subtraction per row, a three-way minimum per two columns and per two rows (`logic`) and
// two compares: 146 instructions per 2048 pairs (32 rows x 64 columns), 3/64 of a flop per pair.  Any-hit apex
// sweep, per pair: one subtraction (the u half of the bounding-square test), a three-way minimum
// per two pairs and a compare per eight: 13 instructions per 8 pairs, one flop per pair.
struct RasterMix { int fma, add, logic, compare; };
constexpr RasterMix rt_raster_mix(int rpl)
{
    // wave-block form: 64 columns and NL rows per lane (104 instructions per wave, triangle and 4096 x 256 pixels)
    if (rpl >= RT_WAVE_BLOCK_RPL) return RasterMix{0, 64 + rpl / 4096, 32 + rpl / 8192, 2};
    return RasterMix{0, rpl / rt_raster_lane_rows(rpl) + rt_raster_lane_rows(rpl),
                     (rpl / rt_raster_lane_rows(rpl) > 1 ? rpl / rt_raster_lane_rows(rpl) / 2 : 0) +
                         (rt_raster_lane_rows(rpl) > 1 ? rt_raster_lane_rows(rpl) / 2 : 0),
                     2};
}
// fp32 flops per pair (FMA = 2) of that loop
constexpr double rt_raster_flop_per_pair(int rpl)
{
    return (2.0 * rt_raster_mix(rpl).fma + rt_raster_mix(rpl).add) / rpl;
}
constexpr double RT_APEX_FLOP_PER_PAIR = 1.0;
constexpr double RT_APEX_INSTR_PER_PAIR = 25.0 / 16.0;     // 16 rays per lane: 16 subtractions, 8 minima, 1 compare

// per-ray candidate counter written by the sweep: low bits = candidates pushed, plus
constexpr int RT_CNT_OCCLUDED = 0x40000000;   // any-hit: a certain hit was found
constexpr int RT_CNT_RETIRED = 0x20000000;    // unused ray slot (NaN origin, zero length, padding)
constexpr int RT_CNT_MASK = 0x0FFFFFFF;

// ---------------------------------------------------------------- device buffer helper
// Grow-only device allocation; keeps steady-state frames free of hipMalloc/hipFree.
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes);   // returns RT_OK / RT_E_HIP
    void release();
    template <class T> T* as() const { return static_cast<T*>(p); }
};

// ---------------------------------------------------------------- scene
struct rt_scene {
    int num_v = 0, num_vn = 0, num_vt = 0, num_tris = 0, num_mats = 0, num_tex = 0;
    int num_tris_padded = 0;
    bool transformed = false;
    double radius = 0;          // largest |coordinate| of the eye-space vertices (host copy)
    bool radius_ready = false;  // ... fetched from scratch_max when rays in general position are first traced after a transform

    // object-space master + eye-space working copy, fp64 SoA, 1-indexed (slot 0 carried along)
    DevBuf ox, oy, oz, ex, ey, ez;
    DevBuf nx, ny, nz;          // normals, object space, fp64
    DevBuf tu, tv;              // texture coordinates
    DevBuf iA, iB, iC, iAn, iBn, iCn, iAt, iBt, iCt, imtl;   // int32 per triangle
    DevBuf mats;                // DevMaterial[num_mats]
    DevBuf tex_desc;            // DevTexture[num_tex]
    std::vector<DevBuf> tex_data;
    bool has_tex_idx = false;
    int device = 0;             // HIP device the scene's buffers live on (current at rt_scene_create)
    std::vector<unsigned char> frame_host;      // staging copy of the frame constants
    // host-side extremes of the index arrays, checked against the tables before a render
    int mtl_max = -1;           // largest mtl[] entry
    int map_max = -1;           // largest map_Ka/Kd/Ks entry of any material (-1: no textured material)

    // fp32 sweep buffer: RT_PLANES float4 planes of num_tris_padded entries each, back to back
    // (contents per formulation: k_pack in rt_transform.hip)
    DevBuf planes;
    // same for the apex sweeps: RT_APEX_PLANES planes, repacked for the eye and for each light
    DevBuf planes_apex, tri_order;
    // Order of the current lights' plane sets: 0 = triangle order, 1 = tri_order (a list per light, by
    // the solid angle subtended at it), 2 = tri_scatter (one fixed pseudo-random list for every light)
    int lights_order = 0;
    DevBuf tri_scatter;
    bool tri_scatter_ready = false; // computed for the current triangle count
    // ... and for the lights of one any-hit launch (RT_APEX_LIGHTS sets side by side), sorted and
    // packed on a side stream while the primary sweep runs
    DevBuf planes_light, ws_sort_tri;
    hipStream_t side_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    DevBuf tri_verts;           // 9 doubles per triangle, eye space (see ExactScene::tv)
    DevBuf tri_normals;         // 9 doubles per triangle: its vertex normals, object space (ShadeArgs::tn)
    bool tri_normals_ready = false;
    bool planes_ready = false;  // the general sweep's planes hold the current transform
    DevBuf scratch_max;         // 1 float: max |coordinate| of the eye-space vertices
    DevBuf mat_dev;             // the 4x4 of the current transform

    // survivor counts of the any-hit ranges travel to page-locked host memory behind the launch
    // stream, one slot and one event per range (rt_api.hip: anyhit_ranges)
    int* h_counts = nullptr;
    std::vector<hipEvent_t> count_events;

    // per-call workspaces (grow-only)
    DevBuf ws_cnt, ws_list, ws_ovf, ws_counters, ws_besthi, ws_partials, ws_frame, ws_axis;
    DevBuf ws_hit, ws_tuv, ws_shS, ws_shE, ws_occ, ws_act0, ws_act1, ws_act2, ws_tails, ws_live, ws_bin_pad, ws_occid, ws_sort;
    // the cell-lattice any-hit sweep (rt_sweep_cells.hip): sort space and ray records, the cell
    // table, the triangles' boxes in cell units
    DevBuf ws_cell_rays, ws_cell_table, ws_cell_boxes, ws_cell_wide;
};

struct DevMaterial {
    double Ka[3], Kd[3], Ks[3];
    int map_Ka, map_Kd, map_Ks, pad;
};
struct DevTexture {
    const uint8_t* rgb;
    int w, h;
};

struct DevFrame {           // what the shading kernels read
    int num_lights;
    int pad;
    double obinv[4][4];
    double light_pos[RT_MAX_LIGHTS][3];
    double light_rgb[RT_MAX_LIGHTS][3];
    double light_att[RT_MAX_LIGHTS][2];
    double light_max[RT_MAX_LIGHTS];
};

// ---------------------------------------------------------------- global state (rt_api.hip)
struct RtGlobal {
    bool inited = false;
    int device = -1;
    hipStream_t stream = nullptr;
    bool profiling = false;
    int tile_culling = -1;                  // rt_set_tile_culling; -1: not set, RT_AMD_CULL decides
    std::atomic<int> last_raster_rpl{0};    // pixels per lane of the most recent primary sweep
    std::atomic<int> last_raster_chunks{0}; // and its triangle chunks
    std::atomic<int> last_raster_regular{0};    // and whether it ran the regular-row form
    rt_profile prof{};
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
};
RtGlobal& rt_g();
bool rt_tile_culling();     // the setting in force (rt_set_tile_culling, else RT_AMD_CULL, else off)
int rt_ensure_init();
int rt_device_cus();        // compute units of the device (cached; every device of a node is alike)

// ---------------------------------------------------------------- launchers (defined per .hip file)
// rt_transform.hip
int rt_launch_transform(const double* m16, const double* x, const double* y, const double* z,
                        double* X, double* Y, double* Z, int n, unsigned* max_bits, hipStream_t st);
// ... with rows 0..2 of the matrix passed by value
struct RtMat12 {
    double v[12];
    __host__ __device__ double operator[](int i) const { return v[i]; }
};
int rt_launch_transform(const RtMat12& m, const double* x, const double* y, const double* z,
                        double* X, double* Y, double* Z, int n, unsigned* max_bits, hipStream_t st);
// planes: the general sweep's planes (rt_sweep.hip); verts: the exact pass's copy of the triangles.
// A transform writes the copy only; the planes follow when rays in general position are traced
// (rt_scene::planes_ready).
int rt_launch_pack(rt_scene* sc, hipStream_t st, bool planes, bool verts);
// apex planes for rays through `apex`; zs is folded into the z components (the constant z of
// pixel-grid directions, 1 otherwise); dmax bounds |D| of every ray of the sweep
// squares (or nullptr): three more planes of num_tris_padded entries, the bounding squares of the
// any-hit sweep's first test for the dominant axes 0, 1, 2 (product layout only)
int rt_launch_pack_apex(rt_scene* sc, const double apex[3], double zs, double dmax, float4* dst,
                        bool sign_layout, const int32_t* order, hipStream_t st, float4* squares = nullptr);
// into the RT_APEX_PLANES planes of num_tris_padded entries at dst; order (or nullptr): entry k
// holds triangle order[k]

// The pixel rows one call works on: one block of `block` rows out of every `parts` (rt_rows in
// the public header); parts = 1 is the contiguous band [y0, y0 + rows).  Outputs are indexed by
// local row lr (0 .. rows-1).
struct RowMap {
    int y0, rows, block, parts, part;
    int full_rounds, tail;   // complete rounds of `parts` blocks, and the rows left after them
    // blocks are dealt to the parts in rounds, forwards or backwards after the Thue-Morse
    // sequence (F B B F B F F B ...: backwards when the round number has an odd number of one
    // bits), so that a cost that grows down the image -- steadily or faster -- is shared evenly;
    // the rows left after the last complete round are cut into `parts` nearly equal runs, one a
    // part, instead of whole blocks for some parts and none for the others
    static __host__ __device__ bool backwards(int round)
    {
        unsigned x = (unsigned)round;
        x ^= x >> 16; x ^= x >> 8; x ^= x >> 4; x ^= x >> 2; x ^= x >> 1;
        return x & 1u;
    }
    __host__ __device__ int slot(int round) const { return backwards(round) ? parts - 1 - part : part; }
    // first row, within the tail, of the run that slot s takes
    __host__ __device__ int tail_begin(int s) const { return (int)((long)tail * s / parts); }
    __host__ __device__ int image_row(int lr) const
    {
        const int round = lr / block;
        if (round < full_rounds) return y0 + (round * parts + slot(round)) * block + lr % block;
        return y0 + full_rounds * parts * block + tail_begin(slot(full_rounds)) + (lr - full_rounds * block);
    }
    // the map of part `part` of `parts` over `total` rows from y0; rows is filled in
    static __host__ __device__ RowMap of(int y0, int total, int block, int parts, int part)
    {
        RowMap m{y0, 0, block, parts, part, total / (block * parts), total % (block * parts)};
        const int s = m.slot(m.full_rounds);
        m.rows = m.full_rounds * block + m.tail_begin(s + 1) - m.tail_begin(s);
        return m;
    }
};
inline RowMap rt_band_rows(int y0, int y1) { return RowMap::of(y0, y1 - y0, y1 - y0 > 0 ? y1 - y0 : 1, 1, 0); }

// rt_sweep.hip
struct SweepArgs {
    const float4* planes;   // RT_PLANES planes, plane p starts at planes + p * plane_stride
    long plane_stride;
    int num_tris, num_tris_padded;
    // primary mode
    int W, H;
    RowMap rows;
    double ez;              // (W/2)/tan_half, integer W/2 (driver.c:133)
    // generic mode
    int n_rays;
    const double *S, *E;    // 3 doubles per ray
    const int32_t* active;  // indexed mode: n_rays ray numbers to work on (else nullptr)
    float4* tails;          // apex any-hit sweep: scratch beside `active`, one entry per listed ray -- (len, sig, the
                            // ray's number as bits, -), written by every workgroup for its own rays before its tile
                            // loop and read back in the rare path (what would otherwise be two dependent fetches
                            // and an fp64 square root there, or 24 registers more)
    double scene_radius;    // largest |coordinate| of the eye-space vertices (general-position sweep)
    // triangle range of this launch, tile-aligned; resume = pick up cnt[] left by an earlier range
    int tri_begin, tri_end;
    int resume;
    // outputs
    int32_t* cnt;           // per ray: candidates pushed | RT_CNT_* flags
    int32_t* list;          // RT_LIST_K candidate triangle indices per ray
    // apex any-hit sweep only: `lights` ray sets swept by one launch, set s = blockIdx.y =
    // light * 3 + dominant axis of the rays' direction.  Light l has its planes
    // light_plane_stride float4 after light l-1 and its rays, counts and lists
    // light_ray_stride rays after; set s has its active entries at active + s *
    // light_ray_stride, live_counts[s] of them (n_rays is the largest of those counts, what the
    // grid is sized for; total_rays their sum).
    int lights;
    long light_ray_stride, light_plane_stride;
    long list_stride;       // pixel-grid sweep: entry c of pixel p is list[c * list_stride + p] (neighbouring pixels' entries side by side)
    const int* live_counts;
    long total_rays;
    long est_rays;          // the host's estimate of the rays really alive (<= total_rays; 0: unknown):
                            // picks rays per lane and triangle chunks, never an array bound
    int32_t* occ_id;        // per ray (laid out like cnt): the triangle of its certain hit, or nullptr
    // entry k of light l's planes holds triangle tri_order[l * tri_order_stride + k] (nullptr: k;
    // the stride is num_tris_padded, or 0 when all lights share one list)
    const int32_t* tri_order;
    long tri_order_stride;
    // pixel-grid sweep only: the triangle range is cut into `chunks` equal parts that run as
    // separate workgroups; the chunks of a pixel share its count, list and bound
    int chunks;
    int waves_x;            // pixel-grid sweep only: waves side by side in a workgroup tile (4, 2, 1),
                            // or 0: the waves of the band are numbered row by row and dealt four to a workgroup
    int regular;            // pixel-grid sweep only: every aligned group of 16 local rows is 16 consecutive image rows
    // pixel-grid sweep only: per pixel, the upper bound of the closest certain hit's t so far
    // (+inf before the sweep); cnt[] must be zero before the sweep
    float* best_hi;
    unsigned long long* tests;   // pairs evaluated, accumulated per workgroup (may be nullptr)
    int cull;               // apex sweeps: test every tile's triangles against the wave's whole bundle first
    int debug_skip;         // measurement only: the pixel-grid sweep leaves its rare path out (wrong results)
};
int rt_launch_sweep_rays(const SweepArgs& a, bool anyhit, hipStream_t st);
// rt_sweep_apex.hip: a.planes = apex planes (RT_APEX_PLANES of them)
int rt_launch_sweep_raster(const SweepArgs& a, hipStream_t st);   // primary rays, closest hit
// launch shape of the pixel-grid sweep for a band of W x rows pixels and `tiles` triangle tiles:
// pixels per lane (<= RT_RASTER_RPL) and the number of triangle chunks (<= RT_RASTER_MAX_CHUNKS)
constexpr int RT_RASTER_MAX_CHUNKS = 2048;
// and how the four waves of a workgroup are laid out: waves_x (4, 2 or 1) side by side, the rest
// below, i.e. workgroup tiles of (waves_x * RT_RASTER_COLS * pixels_per_lane) x (64 / waves_x) pixels;
// or waves_x = 0: the wave-sized pixel blocks of the band, numbered row by row, four consecutive
// ones to a workgroup (at most three idle waves in the whole launch; the launcher's choice for the
// form with a coordinate per row); or waves_x = -1: a workgroup per wave-sized block, its four waves
// sweeping a quarter of every tile of triangles each (the launcher's choice for the regular-row form)
// regular: every aligned group of 16 local rows is 16 consecutive image rows (rt_rows_regular)
void rt_raster_shape(int W, int rows, int tiles, bool regular, int* pixels_per_lane, int* chunks, int* waves_x);
inline bool rt_rows_regular(const RowMap& m) { return m.parts == 1 || m.block % 16 == 0; }
// pixel-triangle pairs the working waves of that shape evaluate (a wave covers
// RT_RASTER_COLS * pixels_per_lane x RT_TILE_H pixels; waves wholly off the image idle)
inline long rt_raster_pairs(int W, int rows, int pixels_per_lane, long tris)
{
    const int ww = rt_raster_wave_cols(pixels_per_lane), wh = rt_raster_wave_rows(pixels_per_lane);
    return (long)((W + ww - 1) / ww) * ((rows + wh - 1) / wh) * (64L * pixels_per_lane) * tris;
}
// workgroup tiles that shape needs to cover W x rows pixels
inline long rt_raster_tiles(int W, int rows, int pixels_per_lane, int waves_x)
{
    if (waves_x < 0) {          // a workgroup per wave-sized block
        const int ww = rt_raster_wave_cols(pixels_per_lane), wh = rt_raster_wave_rows(pixels_per_lane);
        return (long)((W + ww - 1) / ww) * ((rows + wh - 1) / wh);
    }
    if (waves_x == 0) {
        const int ww = rt_raster_wave_cols(pixels_per_lane), wh = rt_raster_wave_rows(pixels_per_lane);
        return ((long)((W + ww - 1) / ww) * ((rows + wh - 1) / wh) + 3) / 4;
    }
    const int tw = waves_x * rt_raster_wave_cols(pixels_per_lane), th = rt_raster_wave_rows(pixels_per_lane) * (4 / waves_x);
    return (long)((W + tw - 1) / tw) * ((rows + th - 1) / th);
}
int rt_launch_sweep_apex(const SweepArgs& a, hipStream_t st);     // any-hit over binned ray lists
// shadow rays [light][pixel] -> ray-number lists per (light, dominant axis), stride entries apart
// Only the rays of pixels whose column and local row are multiples of step_in, and not both
// multiples of step_out (0: no exclusion): the levels of the occluder cache.
// hit (or nullptr): the primary hit of every local pixel; pixels that hit nothing have no rays and
// their slots in S and E are not read
// padded_counts: RT_BIN_COUNT_STRIDE * RT_APEX_SETS ints of scratch (the launch counts there, one
// counter per 256 bytes, and copies the totals to out_counts, which it overwrites)
constexpr int RT_BIN_COUNT_STRIDE = 64;
int rt_launch_bin_rays(const double* S, const double* E, const int8_t* axis, const int32_t* hit, int W, int rows, int lights,
                       long stride, int step_in, int step_out, int32_t* out, int* out_counts, int* padded_counts,
                       hipStream_t st);
// Pixel spacing of the occluder cache's levels, coarsest first; the last must be 1.  (A third,
// coarser level -- 16, 4, 1 -- was measured: it saves 4 % of the pair tests and loses more than
// that to its extra chain of small launches.)
#ifndef RT_SHADOW_NEIGHBOURS
#define RT_SHADOW_NEIGHBOURS 4      // samples per axis whose occluders a ray tries: the corners of its
                                    // cell and the ring around them (2: 241 ms of shadow sweeps on the
                                    // 1 M soup at 4K, 4: 227 ms, 6: 227 ms, 8: slower, the tries cost more)
#endif
constexpr int RT_SHADOW_LEVELS = 2;
constexpr int RT_SHADOW_STEPS[RT_SHADOW_LEVELS] = {4, 1};
// survivors of `sets` ray sets (stride entries apart in in/out; cnt of set s starts
// (s / cnt_div) * stride rays in): set s takes its n from in_counts[s] when `in` is given, else
// all n rays; writes out_counts[s]
int rt_launch_sort_triangles(rt_scene* sc, const double apex[3], int32_t* order, hipStream_t st);
int rt_launch_scatter_triangles(rt_scene* sc, int32_t* order, hipStream_t st);
int rt_launch_sort_rays(rt_scene* sc, const double* S, const double* E, const int32_t* ids_in,
                        int32_t* ids_out, const int* counts, const int* counts_dev, int sets,
                        long stride, hipStream_t st);
// slice_counts: RT_COMPACT_SLICES ints per set, scratch
constexpr int RT_COMPACT_SLICES = 2048;
int rt_launch_compact(const int32_t* cnt, const int32_t* in, const int* in_counts, int n, int sets,
                      long stride, int cnt_div, int32_t* out, int* out_counts, int* slice_counts,
                      hipStream_t st);

// rt_sweep_cells.hip: the any-hit sweep of binned shadow-ray lists over lattices of direction
// cells.  A ray set (light, dominant axis) is a set of points (u, v) in [-1, 1]^2 on the plane
// D_k = 1 through the light; the square is cut into 2^gu x 2^gv cells, a ray belongs to the cell
// its fp32 (u, v) falls into, and the sweep tests every (cell, triangle) pair -- the triangle's
// bounding square grown by half a cell against the cell's centre, separably per column and per
// row of a lane's lattice of cells like the pixel-grid sweep -- and behind that every ray of a
// passing cell against the square and the inside test on its own.
struct CellSet {
    int light, axis;        // of the launch's group of lights; dominant axis 0..2
    int gu_log2, gv_log2;   // the set's grid: 2^gu columns (u) by 2^gv rows (v)
    int cell_base;          // the set's first cell among the launch's cells (its first table entry: cell_base << 2 fine_log2)
    int fp_base;            // first footprint (the block of cells one wave owns) of the set
    int fp_cols_log2;       // footprints side by side in a row of the grid
    int ray_base, rays;     // the set's rays in the launch's sorted arrays (host's count)
    int list;               // which of the caller's binned lists the set is (light * 3 + axis)
};
struct CellArgs {
    CellSet set[RT_APEX_SETS];
    int sets;               // sets with rays (empty ones are left out)
    const float4* boxes;    // per set and plane entry: (cx, rx, cy, ry) in cell units, sets x num_tris_padded
    const float4* planes;   // the lights' plane sets (four apex planes + three planes of squares each)
    long plane_stride, light_plane_stride;
    int num_tris_padded;
    int tri_begin, tri_end; // tile-aligned
    int fine_log2;          // every cell is cut into 2^fine x 2^fine fine cells: the grid the rays are sorted and looked up by
    int tile_parts;         // 1, 2 or 4: a workgroup takes every tile of its chunk whole, or a half or a quarter of each wave's 32 triangles (blockIdx.y = chunk * tile_parts + part)
    const int* table;       // fine cells + 1 entries: rays in the fine cells before this one (a set's fine cells row by row
                            // over its whole grid); the rays of fine cell c are rays[table[c] .. table[c + 1])
    float2* rays;           // sorted by fine cell: (u, v); u = NaN once the ray has retired
    const float2* len_sig;  // same order: |light - S| and 1 / (unit direction)_k, for the rare path's ray parameter
    const int32_t* slot;    // the ray's entry in cnt / list / occ_id
    int32_t *cnt, *list, *occ_id;
    const int32_t* tri_order;   // entry k of light l's planes holds triangle tri_order[l * tri_order_stride + k] (nullptr: k)
    long tri_order_stride;      // num_tris_padded, or 0 when all lights share one list
    unsigned long long* ray_pairs;  // (ray, triangle) pairs tested one by one, accumulated (may be nullptr);
                                    // [2]: cell-table look-ups, [3]: rare-path entries of waves
    int debug_skip;                 // measurement only: leave the rare path out (wrong results)
};
struct CellPlan {
    CellArgs args;
    int cells_per_lane;     // 2048, 512, 128 or 32: the kernel shape
    int chunks;             // triangle chunks (blockIdx.y = chunk * args.tile_parts + part)
    long cells, footprints, rays;
    int fine_log2;          // = args.fine_log2
    const int* wide;        // per set: the triangles swept ray by ray instead (k_cell_boxes)
    int largest_set;        // rays of the largest set
};
constexpr int RT_CELLS_WIDE_CAP = 2048;     // wide triangles a set's list holds
// Per-lane cells of the lattice shapes and what their hot loop costs per lane and triangle
constexpr int rt_cells_lane_rows(int r) { return r >= 2048 ? 32 : (r >= 512 ? 16 : (r >= 128 ? 8 : (r >= 32 ? 4 : 2))); }
constexpr int RT_CELLS_SX = 8, RT_CELLS_SY = 8;     // a wave's 64 lanes are the 8 x 8 phases of the lattice
constexpr int rt_cells_wave_cols(int r) { return RT_CELLS_SX * (r / rt_cells_lane_rows(r)); }
constexpr int rt_cells_wave_rows(int r) { return RT_CELLS_SY * rt_cells_lane_rows(r); }
// (a footprint's columns and rows are dealt to the 64 lanes: cols / 64 and rows / 64 tests per lane, at least one each)
constexpr int rt_cells_instr(int r)
{
    return ((rt_cells_wave_cols(r) >= 64 ? rt_cells_wave_cols(r) / 64 : 1) + (rt_cells_wave_rows(r) >= 64 ? rt_cells_wave_rows(r) / 64 : 1)) * 3 / 2 + 2;
}
// Sorts the listed rays of every set into cells, scales the triangles' squares to each set's
// grid and fills `plan`; a.S/E/cnt/list/occ_id/tri_order/planes as for rt_launch_sweep_apex,
// lists = the binned ray numbers (set s at lists + s * a.light_ray_stride, counts_host[s] of them).
int rt_cells_prepare(rt_scene* sc, const SweepArgs& a, const int32_t* lists, const int* counts_host,
                     int sets, hipStream_t st, CellPlan* plan);
int rt_cells_launch(const CellPlan& plan, hipStream_t st);

// rt_exact.hip
struct ExactScene {
    const double *x, *y, *z;
    const int *A, *B, *C;
    int num_tris;
    // optional: the three vertices of triangle i as 9 consecutive doubles (Ax Ay Az Bx .. Cz),
    // the same values as x[A[i]] ..., gathered once per transform so that the exact test of a
    // candidate reads one or two cache lines instead of twelve
    const double* tv;
};
int rt_launch_resolve_primary(const ExactScene& es, int W, int H, double tan_half, const RowMap& rows,
                              const int32_t* cnt, const int32_t* list, long list_stride,
                              int32_t* hit, double* tuv, int32_t* ovf, int* counters,
                              hipStream_t st);
// gate (or nullptr): ray p is skipped altogether (no output written) when gate[p % gate_n] < 0:
// the shadow rays [light][pixel] of pixels whose primary ray hit nothing
int rt_launch_resolve_rays(const ExactScene& es, int n, const double* S, const double* E,
                           const int32_t* cnt, const int32_t* list, bool anyhit,
                           int32_t* hit, double* tuv, double* point, uint8_t* occ,
                           int32_t* ovf, int* counters, const int32_t* gate, long gate_n, hipStream_t st);
// `partials`: RT_OVERFLOW_PARTIAL_BYTES of device memory for the partial winners of rays whose
// sweep is cut into mesh slices
constexpr size_t RT_OVERFLOW_PARTIAL_BYTES = 32 * 2048;
int rt_launch_overflow_primary(const ExactScene& es, int W, int H, double tan_half, const RowMap& rows,
                               const int32_t* ovf, const int* counters,
                               int32_t* hit, double* tuv, void* partials, hipStream_t st);
int rt_launch_overflow_rays(const ExactScene& es, const double* S, const double* E,
                            const int32_t* ovf, const int* counters, bool anyhit,
                            int32_t* hit, double* tuv, double* point, uint8_t* occ,
                            void* partials, hipStream_t st);
// Occluder cache: every listed ray (sets = lights x 3, laid out like the apex sweep's lists)
// tries the triangles that occluded the sample rays around its pixel with the exact test and
// is marked occluded in cnt[] on success.
int rt_launch_try_cached(const ExactScene& es, const double* S, const double* E, const int32_t* lists,
                         const int* counts, int sets, int n_max, long stride, int W, int rows,
                         int row_block, int step, bool record, int32_t* cnt, int32_t* occ_id,
                         hipStream_t st);
int rt_launch_single_ray(const ExactScene& es, const double* S_E_dev, double* out_dev,
                         hipStream_t st);

// rt_shade.hip
struct ShadeArgs {
    ExactScene es;
    const int *An, *Bn, *Cn, *At, *Bt, *Ct, *mtl;
    const double *nx, *ny, *nz, *tu, *tv;
    const double* tn;         // 9 doubles per triangle: its three vertex normals (or nullptr: through An.. and nx..)
    const DevMaterial* mats;
    const DevTexture* tex;
    int num_mats;
    const DevFrame* frame;
    int num_lights;
    int W, H;
    RowMap rows;
    double tan_half;
    const int32_t* hit;       // per pixel (band-local)
    const double* tuv;
    double *shS, *shE;        // shadow rays: [light][pixel] 3 doubles each
    int8_t* sh_axis;          // [light][pixel]: dominant axis of the ray's direction, -1 = no ray (written with the rays; may be nullptr)
    const uint8_t* occ;       // [light][pixel]
    double* argb;
    uint8_t* rgb8;
};
int rt_launch_pack_normals(const double* nx, const double* ny, const double* nz, const int* An, const int* Bn,
                           const int* Cn, int num_tris, double* out, hipStream_t st);
int rt_launch_shadow_setup(const ShadeArgs& a, hipStream_t st);
int rt_launch_shade_final(const ShadeArgs& a, hipStream_t st);
int rt_launch_place_rows(const uint8_t* rgb8, int W, int H, const RowMap& rows, uint8_t* screen,
                         hipStream_t st);
