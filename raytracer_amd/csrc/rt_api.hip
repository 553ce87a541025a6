// rt_api.hip -- C ABI of the device path: library state, the device-resident scene, the
// batched trace / render entry points and the reference's two kernel.h functions.
#include "rt_internal.h"

#include <cstdarg>
#include <cmath>

// ---------------------------------------------------------------- errors / state
static thread_local char g_err[512] = "";

void rt_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

extern "C" const char* rt_last_error(void) { return g_err; }
extern "C" const char* rt_version(void) { return "rt_amd 0.1 (gfx950)"; }

RtGlobal& rt_g()
{
    static RtGlobal g;
    return g;
}

// ---------------------------------------------------------------- static scratch
// Workspace of rt_trace_rays_static, the entry point that allocates nothing: per-ray candidate
// counts and lists, the overflow queue, the counters and the partial winners of overflowed rays,
// for at most RT_STATIC_MAX_RAYS rays.  Every GPU the library runs on holds its own copy of the
// array at its own address until the process ends; the name below is only the host-side handle
// of the symbol, the address kernels are given comes from hipGetSymbolAddress (rt_init).
namespace {
constexpr size_t kStaticAlign = 256;
constexpr size_t static_round(size_t b) { return (b + kStaticAlign - 1) / kStaticAlign * kStaticAlign; }
constexpr size_t kStaticOffCounters = 0;
constexpr size_t kStaticOffCnt = static_round(64);
constexpr size_t kStaticOffList = kStaticOffCnt + static_round(sizeof(int32_t) * RT_STATIC_MAX_RAYS);
constexpr size_t kStaticOffOvf = kStaticOffList + static_round(sizeof(int32_t) * RT_LIST_K * (size_t)RT_STATIC_MAX_RAYS);
constexpr size_t kStaticOffPartials = kStaticOffOvf + static_round(sizeof(int32_t) * RT_STATIC_MAX_RAYS);
constexpr size_t kStaticBytes = kStaticOffPartials + static_round(RT_OVERFLOW_PARTIAL_BYTES);
constexpr int kMaxDevices = 64;
unsigned char* g_scratch_ptr[kMaxDevices] = {};     // device address of `scratch`, per device ordinal
}  // namespace
static __device__ __align__(256) unsigned char scratch[kStaticBytes];

extern "C" int rt_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int rt_init(int device_ordinal)
{
    RtGlobal& g = rt_g();
    int n = rt_device_count();
    if (n <= 0) {
        rt_set_error("rt_init: no HIP device visible (this library has no CPU path)");
        return RT_E_NODEVICE;
    }
    if (device_ordinal < 0 || device_ordinal >= n) {
        rt_set_error("rt_init: device %d out of range [0,%d)", device_ordinal, n);
        return RT_E_ARG;
    }
    RT_HIP(hipSetDevice(device_ordinal));
    hipDeviceProp_t prop;
    RT_HIP(hipGetDeviceProperties(&prop, device_ordinal));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        rt_set_error("rt_init: device %d is %s; this library is built for gfx950 only",
                     device_ordinal, prop.gcnArchName);
        return RT_E_NODEVICE;
    }
    if (!g.ev0) {
        RT_HIP(hipEventCreate(&g.ev0));
        RT_HIP(hipEventCreate(&g.ev1));
    }
    if (device_ordinal < kMaxDevices && !g_scratch_ptr[device_ordinal]) {
        // this device's copy of the static scratch array (the device is current here)
        unsigned char* ptr = nullptr;
        RT_HIP(hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch)));
        g_scratch_ptr[device_ordinal] = ptr;
    }
    (void)rt_device_cus();      // cached once, before any launcher can race for it
    g.device = device_ordinal;
    g.inited = true;
    return RT_OK;
}

int rt_ensure_init()
{
    if (rt_g().inited) return RT_OK;
    return rt_init(0);
}

extern "C" void rt_shutdown(void)
{
    RtGlobal& g = rt_g();
    rt_legacy_invalidate();
    if (g.ev0) { (void)hipEventDestroy(g.ev0); (void)hipEventDestroy(g.ev1); g.ev0 = g.ev1 = nullptr; }
    g.inited = false;
}

extern "C" int rt_set_stream(void* hip_stream)
{
    // (calls such as rt_scene_transform leave work on the stream; what follows on another stream
    // must not overtake it)
    RtGlobal& g = rt_g();
    hipStream_t next = static_cast<hipStream_t>(hip_stream);
    if (g.inited && next != g.stream) RT_HIP(hipStreamSynchronize(g.stream));
    g.stream = next;
    return RT_OK;
}

extern "C" int rt_sweep_config(int* form, int* tile_tris, int* rays_per_lane, int* flop_per_pair)
{
    if (form) *form = 2;        // moment form
    if (tile_tris) *tile_tris = RT_TILE;
    if (rays_per_lane) *rays_per_lane = RT_RPL;
    if (flop_per_pair) *flop_per_pair = RT_FLOP_PER_PAIR;
    return RT_OK;
}

extern "C" int rt_apex_sweep_config(int* tile_tris, int* primary_pixels_per_lane,
                                    double* primary_flop_per_pair, int* shadow_rays_per_lane,
                                    double* shadow_flop_per_pair, int* primary_chunks)
{
    if (tile_tris) *tile_tris = RT_TILE;
    // of the most recent primary sweep (the launcher shares less per lane on small images)
    const int last_rpl = rt_g().last_raster_rpl.load(), last_chunks = rt_g().last_raster_chunks.load();
    const int rpl = last_rpl > 0 ? last_rpl : RT_RASTER_RPL;
    if (primary_chunks) *primary_chunks = last_chunks > 0 ? last_chunks : 1;
    if (primary_pixels_per_lane) *primary_pixels_per_lane = rpl;
    if (primary_flop_per_pair) *primary_flop_per_pair = rt_raster_flop_per_pair(rpl);
    if (shadow_rays_per_lane) *shadow_rays_per_lane = RT_APEX_RPL;
    if (shadow_flop_per_pair) *shadow_flop_per_pair = RT_APEX_FLOP_PER_PAIR;
    return RT_OK;
}

extern "C" int rt_raster_sweep_mix(int* pixels_per_lane, int* fma, int* add, int* logic, int* compare)
{
    const int last_rpl = rt_g().last_raster_rpl.load();
    const int rpl = last_rpl > 0 ? last_rpl : RT_RASTER_RPL;
    const RasterMix m = rt_raster_mix(rpl);
    if (pixels_per_lane) *pixels_per_lane = rpl;
    if (fma) *fma = m.fma;
    if (add) *add = m.add;
    if (logic) *logic = m.logic;
    if (compare) *compare = m.compare;
    return RT_OK;
}

extern "C" int rt_raster_shape_query(int W, int rows, int triangles, int regular_rows, int* pixels_per_lane,
                                     int* chunks, int* waves_x, long* pixel_workgroups)
{
    RT_ARG(W > 0 && rows > 0 && triangles >= 0, "bad size");
    int r = 1, c = 1, wx = 0;
    const int tiles = (triangles + RT_TILE - 1) / RT_TILE;
    rt_raster_shape(W, rows, tiles < 1 ? 1 : tiles, regular_rows != 0, &r, &c, &wx);
    if (pixels_per_lane) *pixels_per_lane = r;
    if (chunks) *chunks = c;
    if (waves_x) *waves_x = wx;
    if (pixel_workgroups) *pixel_workgroups = rt_raster_tiles(W, rows, r, wx);
    return RT_OK;
}

extern "C" int rt_raster_sweep_form(int* regular_rows)
{
    if (regular_rows) *regular_rows = rt_g().last_raster_regular.load();
    return RT_OK;
}

extern "C" int rt_profile_enable(int on) { rt_g().profiling = on != 0; return RT_OK; }

bool rt_tile_culling()
{
    const int set = rt_g().tile_culling;
    if (set >= 0) return set != 0;
    const char* e = getenv("RT_AMD_CULL");      // read at every call: tests switch it
    return e && atoi(e) != 0;
}
extern "C" int rt_set_tile_culling(int on) { rt_g().tile_culling = on < 0 ? -1 : (on != 0); return RT_OK; }
extern "C" int rt_get_tile_culling(void) { return rt_tile_culling() ? 1 : 0; }
extern "C" int rt_profile_reset(void) { rt_g().prof = rt_profile{}; return RT_OK; }
extern "C" int rt_profile_get(rt_profile* out)
{
    RT_ARG(out, "null output");
    *out = rt_g().prof;
    return RT_OK;
}

// ---------------------------------------------------------------- DevBuf
int DevBuf::reserve(size_t bytes)
{
    if (bytes <= cap && p) return RT_OK;
    if (bytes == 0) bytes = 16;
    if (p) { RT_HIP(hipFree(p)); p = nullptr; cap = 0; }
    RT_HIP(hipMalloc(&p, bytes));
    cap = bytes;
    return RT_OK;
}

void DevBuf::release()
{
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
}

static int upload(DevBuf& b, const void* host, size_t bytes)
{
    RT_TRY(b.reserve(bytes));
    RT_HIP(hipMemcpyAsync(b.p, host, bytes, hipMemcpyHostToDevice, rt_g().stream));
    RT_HIP(hipStreamSynchronize(rt_g().stream));
    return RT_OK;
}

// ---------------------------------------------------------------- scene
extern "C" int rt_scene_create(rt_scene** out)
{
    RT_ARG(out, "null output");
    RT_TRY(rt_ensure_init());
    *out = new rt_scene();
    (void)hipGetDevice(&(*out)->device);
    return RT_OK;
}

extern "C" int rt_scene_destroy(rt_scene* sc)
{
    if (!sc) return RT_OK;
    if (sc->side_stream) {
        (void)hipStreamSynchronize(sc->side_stream);
        (void)hipStreamDestroy(sc->side_stream);
        (void)hipEventDestroy(sc->ev_fork);
        (void)hipEventDestroy(sc->ev_join);
    }
    DevBuf* bufs[] = {&sc->ox, &sc->oy, &sc->oz, &sc->ex, &sc->ey, &sc->ez, &sc->nx, &sc->ny, &sc->nz,
                      &sc->tu, &sc->tv, &sc->iA, &sc->iB, &sc->iC, &sc->iAn, &sc->iBn, &sc->iCn,
                      &sc->iAt, &sc->iBt, &sc->iCt, &sc->imtl, &sc->mats, &sc->tex_desc, &sc->planes, &sc->planes_apex, &sc->tri_verts, &sc->tri_normals,
                      &sc->scratch_max, &sc->mat_dev, &sc->ws_partials, &sc->ws_frame, &sc->planes_light, &sc->ws_sort_tri, &sc->ws_cnt, &sc->ws_list, &sc->ws_ovf, &sc->ws_besthi, &sc->ws_axis,
                      &sc->ws_counters, &sc->ws_hit, &sc->ws_tuv, &sc->ws_shS, &sc->ws_shE,
                      &sc->ws_occ, &sc->ws_act0, &sc->ws_act1, &sc->ws_act2, &sc->ws_tails, &sc->ws_live, &sc->ws_bin_pad, &sc->ws_occid, &sc->ws_sort, &sc->tri_order, &sc->tri_scatter,
                      &sc->ws_cell_rays, &sc->ws_cell_table, &sc->ws_cell_boxes, &sc->ws_cell_wide};
    for (DevBuf* b : bufs) b->release();
    for (DevBuf& b : sc->tex_data) b.release();
    if (sc->h_counts) (void)hipHostFree(sc->h_counts);
    for (hipEvent_t e : sc->count_events) (void)hipEventDestroy(e);
    delete sc;
    return RT_OK;
}

extern "C" int rt_scene_num_tris(const rt_scene* sc) { return sc ? sc->num_tris : 0; }

extern "C" int rt_scene_set_vertices(rt_scene* sc, int num_v, const double* x, const double* y,
                                     const double* z)
{
    RT_ARG(sc && x && y && z, "null pointer");
    RT_ARG(num_v >= 0, "negative vertex count");
    const size_t bytes = sizeof(double) * (size_t)(num_v + 1);
    RT_TRY(upload(sc->ox, x, bytes));
    RT_TRY(upload(sc->oy, y, bytes));
    RT_TRY(upload(sc->oz, z, bytes));
    RT_TRY(sc->ex.reserve(bytes));
    RT_TRY(sc->ey.reserve(bytes));
    RT_TRY(sc->ez.reserve(bytes));
    sc->num_v = num_v;
    sc->transformed = false;
    return RT_OK;
}

extern "C" int rt_scene_set_normals(rt_scene* sc, int num_vn, const double* xn, const double* yn,
                                    const double* zn)
{
    RT_ARG(sc && xn && yn && zn, "null pointer");
    RT_ARG(num_vn >= 0, "negative normal count");
    const size_t bytes = sizeof(double) * (size_t)(num_vn + 1);
    RT_TRY(upload(sc->nx, xn, bytes));
    RT_TRY(upload(sc->ny, yn, bytes));
    RT_TRY(upload(sc->nz, zn, bytes));
    sc->num_vn = num_vn;
    sc->tri_normals_ready = false;
    return RT_OK;
}

extern "C" int rt_scene_set_texcoords(rt_scene* sc, int num_vt, const double* u, const double* v)
{
    RT_ARG(sc && u && v, "null pointer");
    RT_ARG(num_vt >= 0, "negative texture-vertex count");
    const size_t bytes = sizeof(double) * (size_t)(num_vt + 1);
    RT_TRY(upload(sc->tu, u, bytes));
    RT_TRY(upload(sc->tv, v, bytes));
    sc->num_vt = num_vt;
    return RT_OK;
}

static bool in_range(const int* idx, int n, int lo, int hi)
{
    for (int i = 0; i < n; ++i)
        if (idx[i] < lo || idx[i] > hi) return false;
    return true;
}

extern "C" int rt_scene_set_triangles(rt_scene* sc, int num_tris, const int* A, const int* B,
                                      const int* C, const int* An, const int* Bn, const int* Cn,
                                      const int* At, const int* Bt, const int* Ct, const int* mtl)
{
    RT_ARG(sc && A && B && C, "null vertex index array");
    RT_ARG(num_tris >= 0, "negative triangle count");
    // 1-indexed like obj_reader.h:41-46: slot 0 of every array is never a real entry (it is left
    // out of the scene's extent, on which the sweeps' rounding bounds rest)
    RT_ARG(in_range(A, num_tris, 1, sc->num_v) && in_range(B, num_tris, 1, sc->num_v) &&
               in_range(C, num_tris, 1, sc->num_v),
           "vertex index outside [1,num_v] (set vertices first)");
    if (An || Bn || Cn) {
        RT_ARG(An && Bn && Cn, "normal index arrays must come together");
        RT_ARG(in_range(An, num_tris, 1, sc->num_vn) && in_range(Bn, num_tris, 1, sc->num_vn) &&
                   in_range(Cn, num_tris, 1, sc->num_vn),
               "normal index outside [1,num_vn] (set normals first)");
    }
    const bool tex = At && Bt && Ct;
    if (tex)
        RT_ARG(in_range(At, num_tris, 1, sc->num_vt) && in_range(Bt, num_tris, 1, sc->num_vt) &&
                   in_range(Ct, num_tris, 1, sc->num_vt),
               "texture index outside [1,num_vt] (set texcoords first)");
    int mtl_max = -1;
    if (mtl) {
        for (int i = 0; i < num_tris; ++i) {
            RT_ARG(mtl[i] >= -1, "material index below -1");
            mtl_max = mtl[i] > mtl_max ? mtl[i] : mtl_max;
        }
    }
    const size_t bytes = sizeof(int) * (size_t)(num_tris > 0 ? num_tris : 1);
    RT_TRY(upload(sc->iA, A, sizeof(int) * (size_t)num_tris));
    RT_TRY(upload(sc->iB, B, sizeof(int) * (size_t)num_tris));
    RT_TRY(upload(sc->iC, C, sizeof(int) * (size_t)num_tris));
    if (An) {
        RT_TRY(upload(sc->iAn, An, sizeof(int) * (size_t)num_tris));
        RT_TRY(upload(sc->iBn, Bn, sizeof(int) * (size_t)num_tris));
        RT_TRY(upload(sc->iCn, Cn, sizeof(int) * (size_t)num_tris));
    } else {
        sc->iAn.release(); sc->iBn.release(); sc->iCn.release();
    }
    if (tex) {
        RT_TRY(upload(sc->iAt, At, sizeof(int) * (size_t)num_tris));
        RT_TRY(upload(sc->iBt, Bt, sizeof(int) * (size_t)num_tris));
        RT_TRY(upload(sc->iCt, Ct, sizeof(int) * (size_t)num_tris));
    }
    sc->has_tex_idx = tex;
    if (mtl) {
        RT_TRY(upload(sc->imtl, mtl, sizeof(int) * (size_t)num_tris));
    } else {
        sc->imtl.release();
    }
    (void)bytes;
    sc->mtl_max = mtl_max;
    sc->num_tris = num_tris;
    sc->num_tris_padded = ((num_tris + RT_TILE - 1) / RT_TILE) * RT_TILE;
    if (sc->num_tris_padded == 0) sc->num_tris_padded = RT_TILE;
    const size_t pb = sizeof(float4) * (size_t)sc->num_tris_padded;
    RT_TRY(sc->planes.reserve(pb * RT_PLANES));
    RT_TRY(sc->planes_apex.reserve(pb * (RT_APEX_PLANES + 1)));     // + the pixel-grid bounding boxes
    RT_TRY(sc->tri_verts.reserve(sizeof(double) * 9 * (size_t)(num_tris > 0 ? num_tris : 1)));
    RT_TRY(sc->scratch_max.reserve(16));
    sc->tri_scatter_ready = false;
    sc->tri_normals_ready = false;
    sc->transformed = false;
    return RT_OK;
}

static int g_tri_stride = 0;
extern "C" void rt_set_triangle_stride(int bytes) { g_tri_stride = bytes; }

// Splits the reference's array-of-structures into index arrays (host side).
static void split_aos(const Triangle* tris, int n, std::vector<int> idx[9], std::vector<int>& mtl)
{
    const size_t stride = g_tri_stride > 0 ? (size_t)g_tri_stride : sizeof(Triangle);
    for (int k = 0; k < 9; ++k) idx[k].resize(n);
    mtl.resize(n);
    const char* base = reinterpret_cast<const char*>(tris);
    for (int i = 0; i < n; ++i) {
        const Triangle* t = reinterpret_cast<const Triangle*>(base + stride * (size_t)i);
        idx[0][i] = t->A;  idx[1][i] = t->B;  idx[2][i] = t->C;
        idx[3][i] = t->An; idx[4][i] = t->Bn; idx[5][i] = t->Cn;
        idx[6][i] = t->At; idx[7][i] = t->Bt; idx[8][i] = t->Ct;
        mtl[i] = t->mtl.index;
    }
}

extern "C" int rt_scene_set_triangles_aos(rt_scene* sc, const Triangle* tris, int num_tris)
{
    RT_ARG(sc && tris, "null pointer");
    RT_ARG(num_tris >= 0, "negative triangle count");
    std::vector<int> idx[9], mtl;
    split_aos(tris, num_tris, idx, mtl);
    // texture indices are uninitialised in the reference for faces without vt
    // (obj_reader.c:96-104,145-160): only pass them on when every one is in range.
    const bool tex = sc->num_vt > 0 && in_range(idx[6].data(), num_tris, 1, sc->num_vt) &&
                     in_range(idx[7].data(), num_tris, 1, sc->num_vt) &&
                     in_range(idx[8].data(), num_tris, 1, sc->num_vt);
    int max_m = -1;
    for (int i = 0; i < num_tris; ++i) max_m = mtl[i] > max_m ? mtl[i] : max_m;
    if (max_m >= 0) {
        // rebuild the material table from the per-triangle copies (obj_reader.c:84-90)
        const size_t stride = g_tri_stride > 0 ? (size_t)g_tri_stride : sizeof(Triangle);
        std::vector<double> Ka(3 * (max_m + 1), 1.0), Kd(3 * (max_m + 1), 1.0), Ks(3 * (max_m + 1), 1.0);
        const char* base = reinterpret_cast<const char*>(tris);
        for (int i = 0; i < num_tris; ++i) {
            if (mtl[i] < 0) continue;
            const Triangle* t = reinterpret_cast<const Triangle*>(base + stride * (size_t)i);
            for (int k = 0; k < 3; ++k) {
                Ka[3 * mtl[i] + k] = t->mtl.Ka[k];
                Kd[3 * mtl[i] + k] = t->mtl.Kd[k];
                Ks[3 * mtl[i] + k] = t->mtl.Ks[k];
            }
        }
        RT_TRY(rt_scene_set_materials(sc, max_m + 1, Ka.data(), Kd.data(), Ks.data(), nullptr,
                                      nullptr, nullptr));
    }
    return rt_scene_set_triangles(sc, num_tris, idx[0].data(), idx[1].data(), idx[2].data(),
                                  idx[3].data(), idx[4].data(), idx[5].data(),
                                  tex ? idx[6].data() : nullptr, tex ? idx[7].data() : nullptr,
                                  tex ? idx[8].data() : nullptr, mtl.data());
}

extern "C" int rt_scene_set_materials(rt_scene* sc, int num_mats, const double* Ka, const double* Kd,
                                      const double* Ks, const int* map_Ka, const int* map_Kd,
                                      const int* map_Ks)
{
    RT_ARG(sc, "null scene");
    RT_ARG(num_mats >= 0, "negative material count");
    RT_ARG(num_mats == 0 || (Ka && Kd && Ks), "null colour array");
    std::vector<DevMaterial> m((size_t)num_mats);
    int map_max = -1;
    for (int i = 0; i < num_mats; ++i) {
        for (int k = 0; k < 3; ++k) {
            m[i].Ka[k] = Ka[3 * i + k];
            m[i].Kd[k] = Kd[3 * i + k];
            m[i].Ks[k] = Ks[3 * i + k];
        }
        m[i].map_Ka = map_Ka ? map_Ka[i] : -1;
        m[i].map_Kd = map_Kd ? map_Kd[i] : -1;
        m[i].map_Ks = map_Ks ? map_Ks[i] : -1;
        m[i].pad = 0;
        RT_ARG(m[i].map_Ka >= -1 && m[i].map_Kd >= -1 && m[i].map_Ks >= -1, "texture index below -1");
        const int mm = m[i].map_Ka > m[i].map_Kd ? m[i].map_Ka : m[i].map_Kd;
        map_max = mm > map_max ? mm : map_max;
        map_max = m[i].map_Ks > map_max ? m[i].map_Ks : map_max;
    }
    if (num_mats > 0) RT_TRY(upload(sc->mats, m.data(), sizeof(DevMaterial) * (size_t)num_mats));
    sc->num_mats = num_mats;
    sc->map_max = map_max;
    return RT_OK;
}

extern "C" int rt_scene_set_textures(rt_scene* sc, int num_textures, const int* w, const int* h,
                                     const uint8_t* const* rgb)
{
    RT_ARG(sc, "null scene");
    RT_ARG(num_textures >= 0, "negative texture count");
    RT_ARG(num_textures == 0 || (w && h && rgb), "null pointer");
    for (DevBuf& b : sc->tex_data) b.release();
    sc->tex_data.assign((size_t)num_textures, DevBuf());
    std::vector<DevTexture> d((size_t)num_textures);
    for (int i = 0; i < num_textures; ++i) {
        RT_ARG(w[i] > 0 && h[i] > 0 && rgb[i], "bad texture");
        RT_TRY(upload(sc->tex_data[i], rgb[i], (size_t)w[i] * h[i] * 3));
        d[i].rgb = sc->tex_data[i].as<uint8_t>();
        d[i].w = w[i];
        d[i].h = h[i];
    }
    if (num_textures > 0) RT_TRY(upload(sc->tex_desc, d.data(), sizeof(DevTexture) * d.size()));
    sc->num_tex = num_textures;
    return RT_OK;
}

static int transform_into(rt_scene* sc, const double m[4][4], bool bake)
{
    RT_ARG(sc && m, "null pointer");
    if (!sc->ox.p) {
        rt_set_error("transform: no vertices set");
        return RT_E_STATE;
    }
    // Nothing here waits for the device: the matrix travels with the launch, and the one number
    // the host ever needs back (the extent of the eye-space vertices, for sweeps of rays in
    // general position) is fetched when such rays are traced.  Whatever follows on the library's
    // stream is ordered behind the transform; rt_set_stream drains the stream it leaves.
    hipStream_t st = rt_g().stream;
    RtMat12 mat;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 4; ++c) mat.v[4 * r + c] = m[r][c];
    const int n = sc->num_v + 1;
    if (bake) {
        RT_TRY(rt_launch_transform(mat, sc->ox.as<double>(), sc->oy.as<double>(), sc->oz.as<double>(),
                                   sc->ox.as<double>(), sc->oy.as<double>(), sc->oz.as<double>(), n, nullptr, st));
        sc->transformed = false;
        return RT_OK;
    }
    if (!sc->iA.p || !sc->scratch_max.p) {
        rt_set_error("transform: no triangles set");
        return RT_E_STATE;
    }
    RT_HIP(hipMemsetAsync(sc->scratch_max.p, 0, 4, st));
    sc->transformed = false;
    RT_TRY(rt_launch_transform(mat, sc->ox.as<double>(), sc->oy.as<double>(), sc->oz.as<double>(),
                               sc->ex.as<double>(), sc->ey.as<double>(), sc->ez.as<double>(), n,
                               sc->scratch_max.as<unsigned>(), st));
    RT_TRY(rt_launch_pack(sc, st, false, true));
    sc->planes_ready = false;
    sc->radius_ready = false;
    sc->transformed = true;
    return RT_OK;
}

extern "C" int rt_scene_transform(rt_scene* sc, const double m[4][4]) { return transform_into(sc, m, false); }
extern "C" int rt_scene_bake(rt_scene* sc, const double m[4][4]) { return transform_into(sc, m, true); }

extern "C" int rt_scene_get_eye_vertices(rt_scene* sc, double* x, double* y, double* z)
{
    RT_ARG(sc && x && y && z, "null pointer");
    if (!sc->transformed) {
        rt_set_error("get_eye_vertices: scene not transformed");
        return RT_E_STATE;
    }
    const size_t bytes = sizeof(double) * (size_t)(sc->num_v + 1);
    RT_HIP(hipStreamSynchronize(rt_g().stream));        // the transform may still be running
    RT_HIP(hipMemcpy(x, sc->ex.p, bytes, hipMemcpyDeviceToHost));
    RT_HIP(hipMemcpy(y, sc->ey.p, bytes, hipMemcpyDeviceToHost));
    RT_HIP(hipMemcpy(z, sc->ez.p, bytes, hipMemcpyDeviceToHost));
    return RT_OK;
}

// ---------------------------------------------------------------- tracing
static ExactScene exact_of(const rt_scene* sc)
{
    ExactScene es;
    es.x = sc->ex.as<double>(); es.y = sc->ey.as<double>(); es.z = sc->ez.as<double>();
    es.A = sc->iA.as<int>(); es.B = sc->iB.as<int>(); es.C = sc->iC.as<int>();
    es.num_tris = sc->num_tris;
    es.tv = sc->tri_verts.as<double>();
    return es;
}

static int require_ready(const rt_scene* sc)
{
    if (!sc) { rt_set_error("null scene"); return RT_E_ARG; }
    if (!sc->transformed) {
        rt_set_error("scene has no eye-space geometry: call rt_scene_transform first");
        return RT_E_STATE;
    }
    return RT_OK;
}

// Runs a sweep launch, optionally bracketed by HIP events on the launch stream.
template <class F>
static int timed_sweep(F&& launch, bool shadow, uint64_t tests_if_closest,
                       unsigned long long* dev_tests)
{
    RtGlobal& g = rt_g();
    if (!g.profiling) return launch();
    RT_HIP(hipEventRecord(g.ev0, g.stream));
    int rc = launch();
    if (rc) return rc;
    RT_HIP(hipEventRecord(g.ev1, g.stream));
    RT_HIP(hipEventSynchronize(g.ev1));
    float ms = 0;
    RT_HIP(hipEventElapsedTime(&ms, g.ev0, g.ev1));
    if (shadow) {
        unsigned long long t = 0;
        RT_HIP(hipMemcpyAsync(&t, dev_tests, sizeof t, hipMemcpyDeviceToHost, g.stream));
        RT_HIP(hipMemsetAsync(dev_tests, 0, sizeof t, g.stream));      // the counter restarts for the next range
        RT_HIP(hipStreamSynchronize(g.stream));
        g.prof.shadow_ms += ms; g.prof.shadow_launches++; g.prof.shadow_tests += t;
    } else {
        g.prof.sweep_ms += ms; g.prof.sweep_launches++; g.prof.sweep_tests += tests_if_closest;
    }
    return RT_OK;
}

static int workspaces(rt_scene* sc, size_t slots)
{
    RT_TRY(sc->ws_cnt.reserve(sizeof(int32_t) * slots));
    RT_TRY(sc->ws_list.reserve(sizeof(int32_t) * slots * RT_LIST_K));
    RT_TRY(sc->ws_ovf.reserve(sizeof(int32_t) * slots));
    RT_TRY(sc->ws_counters.reserve(64));
    RT_TRY(sc->ws_partials.reserve(RT_OVERFLOW_PARTIAL_BYTES));
    return RT_OK;
}

// Where a generic-ray trace keeps its per-ray state: the scene's grow-only buffers, or the
// static scratch array.
struct RayWorkspace {
    int32_t *cnt, *list, *ovf;
    int* counters;
    void* partials;
};
static RayWorkspace scene_workspace(rt_scene* sc)
{
    return RayWorkspace{sc->ws_cnt.as<int32_t>(), sc->ws_list.as<int32_t>(), sc->ws_ovf.as<int32_t>(),
                        sc->ws_counters.as<int>(), sc->ws_partials.p};
}

static int fetch_counters(const int* counters_dev, bool sync_always)
{
    // counters[0] = overflowed rays, [1] = fp64 candidates evaluated, [2..3] = pair counter,
    // [4] = live rays of a generic-ray launch
    RtGlobal& g = rt_g();
    if (!g.profiling && !sync_always) return RT_OK;
    int c[5] = {0, 0, 0, 0, 0};
    RT_HIP(hipMemcpyAsync(c, counters_dev, sizeof c, hipMemcpyDeviceToHost, g.stream));
    RT_HIP(hipStreamSynchronize(g.stream));
    g.prof.exact_overflow += (uint64_t)c[0];
    g.prof.exact_candidates += (uint64_t)c[1];
    g.prof.rays_live += (uint64_t)c[4];
    return RT_OK;
}

// ---- row sets (rt_rows in the public header -> RowMap)
static int rows_of_part(const rt_rows* r)
{
    if (!r || r->block <= 0 || r->parts <= 0 || r->part < 0 || r->part >= r->parts || r->y1 < r->y0)
        return -1;
    if ((long)r->block * r->parts > 0x7fffffffL) return -1;
    return RowMap::of(r->y0, r->y1 - r->y0, r->block, r->parts, r->part).rows;
}

extern "C" int rt_rows_count(const rt_rows* r)
{
    const int n = rows_of_part(r);
    if (n < 0) {
        rt_set_error("rt_rows_count: bad row set");
        return RT_E_ARG;
    }
    return n;
}

extern "C" int rt_rows_image_row(const rt_rows* r, int local_row)
{
    const int n = rows_of_part(r);
    if (n < 0 || local_row < 0 || local_row >= n) {
        rt_set_error("rt_rows_image_row: bad row set or row");
        return RT_E_ARG;
    }
    return RowMap::of(r->y0, r->y1 - r->y0, r->block, r->parts, r->part).image_row(local_row);
}

static int row_map(const rt_rows* r, int H, RowMap* out)
{
    const int n = rows_of_part(r);
    if (n < 0 || r->y0 < 0 || r->y1 > H) {
        rt_set_error("bad row set (need 0 <= y0 <= y1 <= H, block > 0, 0 <= part < parts)");
        return RT_E_ARG;
    }
    *out = RowMap::of(r->y0, r->y1 - r->y0, r->block, r->parts, r->part);
    return RT_OK;
}

static int lights_fork(rt_scene* sc, int lights);
static int lights_launch(rt_scene* sc, const rt_frame* f, int lights, long npix);

// aside_frame (or nullptr): the frame whose first aside_lights lights are sorted and packed on the
// scene's side stream beside the sweep (see lights_fork / lights_launch)
static int trace_primary_rows(rt_scene* sc, int W, int H, double tan_half, const RowMap& rows,
                              int32_t* hit, double* tuv, const rt_frame* aside_frame = nullptr,
                              int aside_lights = 0, bool wait = true)
{
    RtGlobal& g = rt_g();
    hipStream_t st = g.stream;
    const size_t npix = (size_t)W * rows.rows;
    if (aside_frame) RT_TRY(lights_fork(sc, aside_lights));
    int rpl = 1, chunks = 1, waves_x = 4;
    // RT_AMD_RASTER_REGULAR=0: the form with a coordinate per row even where the rows are regular
    // (read at every call: tests run both on the same inputs)
    const char* reg_env = getenv("RT_AMD_RASTER_REGULAR");
    const bool regular = rt_rows_regular(rows) && !rt_tile_culling() && !(reg_env && atoi(reg_env) == 0);
    rt_raster_shape(W, rows.rows, sc->num_tris_padded / RT_TILE, regular, &rpl, &chunks, &waves_x);
    // per pixel: a count, a bound and RT_RASTER_LIST_K list slots, shared by all triangle chunks
    static_assert(RT_RASTER_LIST_K % RT_LIST_K == 0, "the pixel lists are carved out of the ray lists' buffer");
    RT_TRY(workspaces(sc, npix * (size_t)(RT_RASTER_LIST_K / RT_LIST_K)));
    RT_TRY(sc->ws_besthi.reserve(sizeof(float) * npix));
    RT_HIP(hipMemsetAsync(sc->ws_counters.p, 0, 64, st));
    // the sweep keeps its per-pixel state in memory: no candidates yet, no certain hit yet (+inf)
    RT_HIP(hipMemsetAsync(sc->ws_cnt.p, 0, sizeof(int32_t) * npix, st));
    RT_HIP(hipMemsetD32Async((hipDeviceptr_t)sc->ws_besthi.p, 0x7F800000, npix, st));

    // planes for rays from the eye; the constant z of the pixel-grid directions is folded in
    const double ez = (double)(W / 2) / tan_half;       // driver.c:133
    const double eye[3] = {0, 0, 0};                    // driver.c:121-123
    const double hw = 0.5 * W + 1, hh = 0.5 * H + 1;
    RT_TRY(rt_launch_pack_apex(sc, eye, ez, sqrt(hw * hw + hh * hh + ez * ez), sc->planes_apex.as<float4>(), true,
                               nullptr, st, sc->planes_apex.as<float4>() + (size_t)RT_APEX_PLANES * sc->num_tris_padded));

    SweepArgs a{};
    a.planes = sc->planes_apex.as<float4>(); a.plane_stride = sc->num_tris_padded;
    a.num_tris = sc->num_tris; a.num_tris_padded = sc->num_tris_padded;
    a.W = W; a.H = H; a.rows = rows;
    a.ez = ez;
    a.cnt = sc->ws_cnt.as<int32_t>(); a.list = sc->ws_list.as<int32_t>();
    a.list_stride = (long)npix;         // slot-major lists: entry c of pixel p at list[c * npix + p]
    a.best_hi = sc->ws_besthi.as<float>();
    a.chunks = chunks;
    a.waves_x = waves_x;
    a.regular = regular ? 1 : 0;
    a.debug_skip = getenv("RT_AMD_RASTER_SKIP_RARE") ? atoi(getenv("RT_AMD_RASTER_SKIP_RARE")) : 0;
    a.cull = rt_tile_culling() ? 1 : 0;
    a.tests = nullptr;
#ifdef RT_RASTER_CHECK_INTERVAL
    a.tests = reinterpret_cast<unsigned long long*>(sc->ws_counters.as<char>() + 8);
#endif

    a.tri_begin = 0; a.tri_end = sc->num_tris_padded;
    // pairs of the band's own pixels (the lanes of a wave that reaches beyond the image's edge
    // execute their tests too; those are not counted)
    const uint64_t tests = (uint64_t)npix * (uint64_t)sc->num_tris_padded;
    g.last_raster_rpl.store(rpl, std::memory_order_relaxed);
    g.last_raster_chunks.store(chunks, std::memory_order_relaxed);
    g.last_raster_regular.store(regular ? 1 : 0, std::memory_order_relaxed);
    RT_TRY(timed_sweep([&] { return rt_launch_sweep_raster(a, st); }, false, tests, nullptr));
    // (queued by the host after the sweep, so that the sweep is not kept waiting for a dozen small
    // launches; on the device the side stream only waits for what preceded the sweep)
    if (aside_frame) RT_TRY(lights_launch(sc, aside_frame, aside_lights, (long)npix));
    if (g.profiling) g.prof.rays_primary += npix;
#ifdef RT_RASTER_CHECK_INTERVAL
    {
        unsigned long long t = 0;
        RT_HIP(hipMemcpy(&t, a.tests, sizeof t, hipMemcpyDeviceToHost));
        fprintf(stderr, "rt_amd: interval check: %llu passes outside the interval\n", t >> 48);
    }
#endif

    const ExactScene es = exact_of(sc);
    int* counters = sc->ws_counters.as<int>();
    RT_TRY(rt_launch_resolve_primary(es, W, H, tan_half, rows, a.cnt, a.list, a.list_stride, hit, tuv, sc->ws_ovf.as<int32_t>(), counters, st));
    RT_TRY(rt_launch_overflow_primary(es, W, H, tan_half, rows, sc->ws_ovf.as<int32_t>(), counters,
                                      hit, tuv, sc->ws_partials.p, st));
    RT_TRY(fetch_counters(counters, false));
    if (wait) RT_HIP(hipStreamSynchronize(st));     // (the pixel loop goes on queueing behind it instead)
    return RT_OK;
}

extern "C" int rt_trace_primary_part(rt_scene* sc, int W, int H, double tan_half, const rt_rows* rows,
                                     int32_t* hit, double* tuv)
{
    RT_TRY(require_ready(sc));
    RT_ARG(hit, "null hit buffer");
    RT_ARG(W > 0 && H > 0, "bad image size");
    RT_ARG(tan_half > 0, "tan_half must be positive");
    RowMap rm;
    RT_TRY(row_map(rows, H, &rm));
    if (rm.rows == 0) return RT_OK;
    return trace_primary_rows(sc, W, H, tan_half, rm, hit, tuv);
}

extern "C" int rt_trace_primary(rt_scene* sc, int W, int H, double tan_half, int y0, int y1,
                                int32_t* hit, double* tuv)
{
    RT_ARG(y0 < y1, "bad image size or row band");
    const rt_rows r = {y0, y1, y1 - y0, 1, 0};
    return rt_trace_primary_part(sc, W, H, tan_half, &r, hit, tuv);
}

// Any-hit sweep in triangle ranges of growing length; the survivors are compacted between
// ranges, so a ray's cost becomes ~ the index of its first certain occluder, not the mesh
// size.  Two users:
//   * rays in general position (rt_occluded): one set of n rays, the first range sweeps them
//     all, later ranges run indexed over the survivor list;
//   * shadow rays of the pixel loop: `sets` lists (light x dominant axis, binned beforehand
//     into ws_act0 with their sizes in ws_live), indexed from the first range on.
// launch(a) starts one range for all sets.
constexpr int kCountSlots = 48;     // ranges of one any-hit sweep whose counts can be in flight

// The sizes of `sets` binned ray lists (counts_dev, on the device) on the host: the one thing the
// host waits for before an any-hit phase.
static int binned_counts(rt_scene* sc, const int* counts_dev, int sets, int* out)
{
    hipStream_t st = rt_g().stream;
    if (!sc->h_counts) {
        RT_HIP(hipHostMalloc(reinterpret_cast<void**>(&sc->h_counts), sizeof(int) * RT_APEX_SETS * kCountSlots,
                             hipHostMallocDefault));
        sc->count_events.resize(kCountSlots);
        for (hipEvent_t& e : sc->count_events) RT_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    RT_HIP(hipMemcpyAsync(sc->h_counts, counts_dev, sizeof(int) * sets, hipMemcpyDeviceToHost, st));
    RT_HIP(hipEventRecord(sc->count_events[0], st));
    RT_HIP(hipEventSynchronize(sc->count_events[0]));
    for (int k = 0; k < sets; ++k) out[k] = sc->h_counts[k];
    return RT_OK;
}

// known (or nullptr): the sizes of the binned lists when the caller has fetched them already
template <class Launch>
static int anyhit_ranges(rt_scene* sc, SweepArgs a, int n, int sets, long stride, bool binned,
                         Launch&& launch, const int* known = nullptr)
{
    hipStream_t st = rt_g().stream;
    const int np = sc->num_tris_padded;
    RT_ARG(sets >= 1 && sets <= RT_APEX_SETS, "too many ray sets for one launch");
    const int cnt_div = binned ? 3 : 1;
    int32_t* act[2] = {sc->ws_act0.as<int32_t>(), sc->ws_act1.as<int32_t>()};
    int* live_dev[2] = {sc->ws_live.as<int>(), sc->ws_live.as<int>() + RT_APEX_SETS};
    static const double growth_env = getenv("RT_AMD_RANGE_GROWTH") ? atof(getenv("RT_AMD_RANGE_GROWTH")) : 0.0;
    static const int first_div = getenv("RT_AMD_RANGE_FIRST") ? atoi(getenv("RT_AMD_RANGE_FIRST")) : 64;
    int first = np / first_div;
    first = ((first + RT_TILE - 1) / RT_TILE) * RT_TILE;
    if (first < 16 * RT_TILE) first = 16 * RT_TILE;
    if (!sc->h_counts) {
        RT_HIP(hipHostMalloc(reinterpret_cast<void**>(&sc->h_counts), sizeof(int) * RT_APEX_SETS * kCountSlots,
                             hipHostMallocDefault));
        sc->count_events.resize(kCountSlots);
        for (hipEvent_t& e : sc->count_events) RT_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    // The host only needs the survivor counts to size the next launches (grid, rays per lane,
    // triangle chunks) and to stop early; the kernels read the counts themselves on the device.
    // So the counts of a compaction are copied to page-locked memory behind it on the stream,
    // and the host sizes range k + 1 from the counts that were valid BEFORE range k -- an upper
    // bound, since rays only ever retire -- which have long arrived while range k is still running:
    // the host never waits for the device between ranges, and a workgroup beyond the real
    // count returns at once.
    int live = n;
    long total = (long)n * sets;
    long est = total, prev_total = 0;       // estimate of the rays really alive (see SweepArgs::est_rays)
    auto post_counts = [&](const int* dev, int slot) -> int {
        RT_HIP(hipMemcpyAsync(sc->h_counts + (size_t)slot * RT_APEX_SETS, dev, sizeof(int) * sets,
                              hipMemcpyDeviceToHost, st));
        RT_HIP(hipEventRecord(sc->count_events[slot], st));
        return RT_OK;
    };
    auto take_counts = [&](int slot) -> int {
        RT_HIP(hipEventSynchronize(sc->count_events[slot]));
        const int* c = sc->h_counts + (size_t)slot * RT_APEX_SETS;
        prev_total = total;
        live = 0; total = 0;
        for (int k = 0; k < sets; ++k) { live = c[k] > live ? c[k] : live; total += c[k]; }
        est = total;
        return RT_OK;
    };
    int slot_next = 0;          // slot the next post goes to
    if (binned && known) {
        live = 0; total = 0;
        for (int k = 0; k < sets; ++k) { live = known[k] > live ? known[k] : live; total += known[k]; }
        est = total;
        // (the ray sort below reads the counts from the first slot of the host array)
        for (int k = 0; k < sets; ++k) sc->h_counts[k] = known[k];
        ++slot_next;
    } else if (binned) {        // the sizes of the binned lists: these the host does wait for
        RT_TRY(post_counts(live_dev[0], slot_next));
        RT_TRY(take_counts(slot_next));
        ++slot_next;
    }
    // ranges grow by 2x; by 3x for sweeps of fewer rays than RT_AMD_RANGE_GROWTH3_RAYS (their launches
    // are too short to fill the machine, and four of them cost less than six: an eighth of the
    // 1 M-triangle 4K frame takes 15.7 instead of 16.0 ms; 4x was measured too: most parts gain as
    // much, the slowest one and every quarter lose to the extra tests)
    static const long growth3_rays = getenv("RT_AMD_RANGE_GROWTH3_RAYS") ? atol(getenv("RT_AMD_RANGE_GROWTH3_RAYS")) : 400000;
    const double growth = growth_env > 1.0 ? growth_env : (total < growth3_rays ? 3.0 : 2.0);
    // a sweep of a few milliseconds is not worth the compaction round trips
    // (RT_AMD_RANGE_MIN_WORK, read at every call: tests set it to 0 to run ranges on small inputs)
    const char* min_work_env = getenv("RT_AMD_RANGE_MIN_WORK");
    const double min_work = min_work_env ? atof(min_work_env) : 1.5e10;
    if ((double)np * (double)total < min_work) first = np;
    int begin = 0, len = first, cur = 0;
    bool indexed = binned;
    // a long sweep of binned rays (they all end at their light): order every set by direction
    // as seen from the light first (rt_sort.hip)
    static const bool sort_rays = !(getenv("RT_AMD_SORT_RAYS") && atoi(getenv("RT_AMD_SORT_RAYS")) == 0);
    if (binned && sort_rays && first < np && total > 0) {
        RT_TRY(rt_launch_sort_rays(sc, a.S, a.E, act[0], act[1], sc->h_counts, live_dev[0], sets, stride, st));
        RT_HIP(hipMemcpyAsync(live_dev[1], live_dev[0], sizeof(int) * RT_APEX_SETS, hipMemcpyDeviceToDevice, st));
        cur = 1;
    }
    // with the profiling events on every launch is followed by a sync anyway: use exact counts
    static const bool lagged_env = !(getenv("RT_AMD_RANGE_SYNC") && atoi(getenv("RT_AMD_RANGE_SYNC")) != 0);
    int posted = -1;            // slot of the counts the most recent compaction produced
    while (begin < np && live > 0) {
        int end = begin + len;
        if (end > np || np - end < len / 2) end = np;
        a.tri_begin = begin; a.tri_end = end;
        a.resume = begin > 0;
        a.active = indexed ? act[cur] : nullptr;
        a.tails = sc->ws_tails.as<float4>();
        a.live_counts = indexed ? live_dev[cur] : nullptr;
        a.n_rays = indexed ? live : n;          // indexed: (a bound on) the largest set, sizes the grid
        a.total_rays = total;
        a.est_rays = est;
        a.lights = sets;
        a.light_ray_stride = stride;
        const double ms0 = rt_g().prof.shadow_ms;
        const uint64_t tests0 = rt_g().prof.shadow_tests;
        RT_TRY(timed_sweep([&] { return launch(a); }, true, 0, a.tests));
        if (getenv("RT_AMD_TRACE_PASSES")) {
            const double ms = rt_g().prof.shadow_ms - ms0;
            const double tests = (double)(rt_g().prof.shadow_tests - tests0);
            fprintf(stderr, "rt_amd: any-hit range [%d,%d) over %d set(s), at most %ld rays, largest set at most %d: %.3f ms, "
                            "%.3e tests, %.2e tests/s\n", begin, end, sets, total,
                    indexed ? live : n, ms, tests, ms > 0 ? tests / (ms * 1e-3) : 0.0);
        }
        begin = end;
        if (begin >= np) break;
        // the counts the compaction before this range produced have arrived by now: they bound
        // the survivors of this range too
        if (posted >= 0) RT_TRY(take_counts(posted));
        // compact the survivors (from the full set the first time in the unbinned case)
        RT_HIP(hipMemsetAsync(live_dev[cur ^ 1], 0, sizeof(int) * RT_APEX_SETS, st));
        RT_TRY(rt_launch_compact(a.cnt, indexed ? act[cur] : nullptr, indexed ? live_dev[cur] : nullptr,
                                 indexed ? live : n, sets, stride, cnt_div, act[cur ^ 1], live_dev[cur ^ 1],
                                 sc->ws_live.as<int>() + 3 * RT_APEX_SETS, st));
        posted = slot_next % kCountSlots;
        ++slot_next;
        RT_TRY(post_counts(live_dev[cur ^ 1], posted));
        if (!lagged_env || rt_g().profiling) {
            RT_TRY(take_counts(posted));      // exact counts, host waits
        } else if (prev_total > 0) {
            // `total` counts the rays that went into the range just launched; expect the range to
            // thin them out like the one before it did
            double keep = (double)total / (double)prev_total;
            keep = keep < 0.25 ? 0.25 : (keep > 1.0 ? 1.0 : keep);
            est = (long)(total * keep);
        }
        cur ^= 1;
        indexed = true;
        len = (int)(len * growth + RT_TILE - 1) / RT_TILE * RT_TILE;
    }
    return RT_OK;
}

// Whether the any-hit phase of `total` binned shadow rays runs as a sweep over lattices of
// direction cells (rt_sweep_cells.hip) or as the ray-by-ray sweep in triangle ranges
// (rt_sweep_apex.hip).  RT_AMD_SHADOW_LATTICE = 0 / 1 forces one or the other (read at every call:
// tests run both on the same inputs); otherwise the lattice takes every phase large enough to
// repay its ray sort.  Tile culling keeps the ray-by-ray sweep, which it is defined on.
static bool use_cell_lattice(long total, int np)
{
    if (rt_tile_culling()) return false;
    if (const char* e = getenv("RT_AMD_SHADOW_LATTICE")) return atoi(e) != 0;
    static const double min_pairs = getenv("RT_AMD_LATTICE_MIN_PAIRS") ? atof(getenv("RT_AMD_LATTICE_MIN_PAIRS")) : 2e10;
    return (double)total * (double)np >= min_pairs;
}

// One any-hit phase over the binned lists (lists + s * stride, counts[s] rays) as a cell-lattice
// sweep: no triangle ranges, no compaction, one sweep launch.
static int anyhit_cells(rt_scene* sc, SweepArgs a, int sets, long stride, const int32_t* lists, const int* counts)
{
    RtGlobal& g = rt_g();
    hipStream_t st = g.stream;
    a.light_ray_stride = stride;
    CellPlan plan{};
    RT_TRY(rt_cells_prepare(sc, a, lists, counts, sets, st, &plan));
    if (plan.rays == 0) return RT_OK;
    if (getenv("RT_AMD_TRACE_PASSES"))
        fprintf(stderr, "rt_amd: cell sweep: %ld rays in %d set(s), %ld cells (%d per lane, %ld footprints) of %d fine cells each, %d chunks\n",
                plan.rays, plan.args.sets, plan.cells, plan.cells_per_lane, plan.footprints, 1 << (2 * plan.fine_log2), plan.chunks);
    if (!g.profiling) return rt_cells_launch(plan, st);
    plan.args.ray_pairs = a.tests;      // (the counter block's bytes 8..15 and 24..31)
    RT_HIP(hipMemsetAsync(a.tests, 0, 7 * sizeof(unsigned long long), st));
    RT_HIP(hipEventRecord(g.ev0, st));
    RT_TRY(rt_cells_launch(plan, st));
    RT_HIP(hipEventRecord(g.ev1, st));
    RT_HIP(hipEventSynchronize(g.ev1));
    float ms = 0;
    RT_HIP(hipEventElapsedTime(&ms, g.ev0, g.ev1));
    unsigned long long tt[7] = {0, 0, 0, 0, 0, 0, 0};
    RT_HIP(hipMemcpyAsync(tt, a.tests, sizeof tt, hipMemcpyDeviceToHost, st));
    RT_HIP(hipMemsetAsync(a.tests, 0, sizeof tt, st));
    RT_HIP(hipStreamSynchronize(st));
    const unsigned long long t = tt[0];
    g.prof.cell_lookups += tt[2];
    g.prof.cell_entries += tt[3];
    g.prof.cell_fine = 1ull << (2 * plan.fine_log2);
    g.prof.shadow_ms += ms;
    g.prof.shadow_launches++;
    // pairs decided: every listed ray against every triangle; what the hot loop executes is one
    // test per (cell, triangle) pair, and the rare path one per (ray, triangle) pair it visits
    g.prof.shadow_tests += (uint64_t)plan.rays * (uint64_t)a.num_tris_padded;
    g.prof.cell_ms += ms;
    g.prof.cell_launches++;
    g.prof.cell_pairs += (uint64_t)plan.cells * (uint64_t)a.num_tris_padded;
    g.prof.cell_ray_pairs += t;
    g.prof.cell_rays += (uint64_t)plan.rays;
    g.prof.cells_per_lane = (uint64_t)plan.cells_per_lane;
    if (getenv("RT_AMD_TRACE_PASSES"))
        fprintf(stderr, "rt_amd: cell sweep: %.3f ms, %.3e cell-triangle pairs (%.2e /s), %.3e ray-triangle pairs one by one, %.3e cell look-ups, %.3e wave entries\n",
                ms, (double)plan.cells * a.num_tris_padded, (double)plan.cells * a.num_tris_padded / (ms * 1e-3), (double)t, (double)tt[2], (double)tt[3]);
    return RT_OK;
}

static int anyhit_buffers(rt_scene* sc, size_t entries)
{
    RT_TRY(sc->ws_act0.reserve(sizeof(int32_t) * entries));
    RT_TRY(sc->ws_act1.reserve(sizeof(int32_t) * entries));
    RT_TRY(sc->ws_tails.reserve(sizeof(float4) * entries));
    RT_TRY(sc->ws_live.reserve(sizeof(int) * (3 + RT_COMPACT_SLICES) * RT_APEX_SETS));   // counts + the compaction's slice sums
    RT_TRY(sc->ws_bin_pad.reserve(sizeof(int) * RT_BIN_COUNT_STRIDE * RT_APEX_SETS));
    return RT_OK;
}

// fp64 confirmation of the candidate lists of n rays + the overflow fallback + statistics
static int resolve_rays(rt_scene* sc, const RayWorkspace& ws, int n, const double* S, const double* E,
                        bool anyhit, int32_t* hit, double* tuv, double* point, uint8_t* occ,
                        const int32_t* gate = nullptr, long gate_n = 0)
{
    RtGlobal& g = rt_g();
    const ExactScene es = exact_of(sc);
    RT_TRY(rt_launch_resolve_rays(es, n, S, E, ws.cnt, ws.list, anyhit, hit, tuv, point, occ, ws.ovf,
                                  ws.counters, gate, gate_n, g.stream));
    RT_TRY(rt_launch_overflow_rays(es, S, E, ws.ovf, ws.counters, anyhit, hit, tuv, point, occ,
                                   ws.partials, g.stream));
    if (!g.profiling) return RT_OK;        // (the record below is only kept while profiling)
    const uint64_t before = g.prof.rays_live;
    RT_TRY(fetch_counters(ws.counters, false));
    if (anyhit) g.prof.rays_shadow += g.prof.rays_live - before;
    else g.prof.rays_primary += g.prof.rays_live - before;
    return RT_OK;
}

// Rays in general position (rt_sweep.hip).
static int trace_rays_impl(rt_scene* sc, const RayWorkspace* static_ws, int n, const double* S,
                           const double* E, bool anyhit, int32_t* hit, double* tuv, double* point,
                           uint8_t* occ)
{
    hipStream_t st = rt_g().stream;
    if (n == 0) return RT_OK;
    if (!static_ws) RT_TRY(workspaces(sc, (size_t)n));
    const RayWorkspace ws = static_ws ? *static_ws : scene_workspace(sc);
    RT_HIP(hipMemsetAsync(ws.counters, 0, 64, st));
    if (!sc->planes_ready) {        // first rays in general position since the transform
        RT_TRY(rt_launch_pack(sc, st, true, false));
        sc->planes_ready = true;
    }
    if (!sc->radius_ready) {
        float radius = 0;
        RT_HIP(hipMemcpyAsync(&radius, sc->scratch_max.p, sizeof radius, hipMemcpyDeviceToHost, st));
        RT_HIP(hipStreamSynchronize(st));
        sc->radius = (double)radius;
        sc->radius_ready = true;
    }
    SweepArgs a{};
    a.planes = sc->planes.as<float4>(); a.plane_stride = sc->num_tris_padded;
    a.num_tris = sc->num_tris; a.num_tris_padded = sc->num_tris_padded;
    a.scene_radius = sc->radius;
    a.n_rays = n; a.S = S; a.E = E;
    a.cnt = ws.cnt; a.list = ws.list;
    a.tests = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(ws.counters) + 8);
    const int np = sc->num_tris_padded;
    if (!anyhit) {
        a.tri_begin = 0; a.tri_end = np;
        const uint64_t wgs = (uint64_t)((n + RT_RAYS_PER_WG - 1) / RT_RAYS_PER_WG);
        RT_TRY(timed_sweep([&] { return rt_launch_sweep_rays(a, false, st); }, false,
                           wgs * RT_RAYS_PER_WG * (uint64_t)np, a.tests));
    } else {
        if (static_ws) {
            rt_set_error("any-hit sweeps have no static workspace");
            return RT_E_ARG;
        }
        RT_TRY(anyhit_buffers(sc, (size_t)n));
        RT_TRY(anyhit_ranges(sc, a, n, 1, n, false,
                             [&](const SweepArgs& r) { return rt_launch_sweep_rays(r, true, st); }));
    }
    return resolve_rays(sc, ws, n, S, E, anyhit, hit, tuv, point, occ);
}

// The order in which a light's plane set holds the triangles (rt_scene::lights_order):
//   1  by the solid angle they subtend at the light, largest first (rt_sort.hip), a sort per light
//      and frame: occluded rays meet an occluder early, which the ray-by-ray sweep in triangle
//      ranges gains from (rays that retire in the first ranges drop out of the later ones);
//   2  one fixed pseudo-random order for all lights, computed once per mesh: the cell-lattice
//      sweep meets every triangle in one launch and only needs the tiles it deals to its
//      workgroups to be alike -- in a mesh file's own order the triangles of a tile lie side by
//      side and send their workgroup to the same few rays.  Measured at 3840x2160 with the lattice
//      sweep (frame, ms): 100 000-triangle torus 9.6 in file order, 4.2 by solid angle, 3.7
//      scattered; 69 000-triangle sphere 14.4, 5.7, 4.8; a soup of a million triangles (random
//      order as generated) 10.4, 10.7, 10.7 -- and an eighth of that frame spends no 0.8 ms on
//      two sorts it does not need;
//   0  as the mesh lists them.
// Order 2 is taken when even the first, coarsest level of the shadow pass (one pixel in sixteen)
// is large enough for the lattice sweep, order 1 otherwise.  RT_AMD_SORT_TRIS = 0 / 1 / 2 forces
// one (read at every call).
static int light_triangle_order(const rt_scene* sc, long npix, int lights)
{
    if (const char* e = getenv("RT_AMD_SORT_TRIS")) {
        const int v = atoi(e);
        return v < 0 || v > 2 ? 1 : v;
    }
    constexpr int coarsest = RT_SHADOW_STEPS[0] * RT_SHADOW_STEPS[0];
    return use_cell_lattice(npix * lights / coarsest, sc->num_tris_padded) ? 2 : 1;
}

// Plane sets of lights [l0, l0 + lights) for one any-hit launch, on stream st: every light's
// triangles in the order light_triangle_order chooses, packed relative to the light.
// |D'| <= sqrt(3) for a direction scaled to a unit dominant component.  npix: pixels of the
// call (every one of them may send a ray to every light).
static int prepare_lights(rt_scene* sc, const rt_frame* f, int l0, int lights, long npix, hipStream_t st)
{
    const int order = light_triangle_order(sc, npix, lights);
    sc->lights_order = order;
    if (order == 2 && !sc->tri_scatter_ready) {
        RT_TRY(sc->tri_scatter.reserve(sizeof(int32_t) * (size_t)sc->num_tris_padded));
        RT_TRY(rt_launch_scatter_triangles(sc, sc->tri_scatter.as<int32_t>(), st));
        sc->tri_scatter_ready = true;
    }
    const size_t set_planes = (size_t)RT_LIGHT_PLANES * sc->num_tris_padded;
    for (int k = 0; k < lights; ++k) {
        int32_t* ord = order == 1 ? sc->tri_order.as<int32_t>() + (size_t)k * sc->num_tris_padded
                                  : (order == 2 ? sc->tri_scatter.as<int32_t>() : nullptr);
        if (order == 1) RT_TRY(rt_launch_sort_triangles(sc, f->light_in_eye_space[l0 + k], ord, st));
        float4* dst = sc->planes_light.as<float4>() + k * set_planes;
        RT_TRY(rt_launch_pack_apex(sc, f->light_in_eye_space[l0 + k], 1.0, 1.75, dst, false, ord, st,
                                   dst + (size_t)RT_APEX_PLANES * sc->num_tris_padded));
    }
    return RT_OK;
}

static int reserve_light_buffers(rt_scene* sc, int lights)
{
    RT_TRY(sc->planes_light.reserve(sizeof(float4) * (size_t)sc->num_tris_padded * RT_LIGHT_PLANES * lights));
    RT_TRY(sc->tri_order.reserve(sizeof(int32_t) * (size_t)sc->num_tris_padded * lights));
    return RT_OK;
}

// prepare_lights for the first group of lights on the scene's side stream, beside the primary
// sweep: the sorts and packs (bandwidth-bound, well under a millisecond) then run while the
// instruction-bound sweep does, instead of after it.  lights_fork marks, on the launch stream,
// the point the side stream has to wait for (everything queued before the sweep: the transform,
// an earlier frame's use of the buffers); lights_launch queues the work; the any-hit phase waits
// for ev_join.
static int lights_fork(rt_scene* sc, int lights)
{
    if (!sc->side_stream) {
        RT_HIP(hipStreamCreateWithFlags(&sc->side_stream, hipStreamNonBlocking));
        RT_HIP(hipEventCreateWithFlags(&sc->ev_fork, hipEventDisableTiming));
        RT_HIP(hipEventCreateWithFlags(&sc->ev_join, hipEventDisableTiming));
    }
    RT_TRY(reserve_light_buffers(sc, lights));
    RT_HIP(hipEventRecord(sc->ev_fork, rt_g().stream));
    return RT_OK;
}

static int lights_launch(rt_scene* sc, const rt_frame* f, int lights, long npix)
{
    RT_HIP(hipStreamWaitEvent(sc->side_stream, sc->ev_fork, 0));
    const int rc = prepare_lights(sc, f, 0, lights, npix, sc->side_stream);
    RT_HIP(hipEventRecord(sc->ev_join, sc->side_stream));      // also after a failed launch: the join must not hang
    return rc;
}

// Shadow rays of the pixel loop: [light][pixel] rays that end at their light (k_shadow_setup).
// Apex sweeps over planes packed relative to each light (rt_sweep_apex.hip), RT_APEX_LIGHTS
// lights per launch; the rays of a light are binned by the dominant axis of their direction.
static int trace_shadow_rays(rt_scene* sc, const rt_frame* f, int W, int nrows, int row_block,
                             const double* S, const double* E, const int8_t* axis, const int32_t* hit, uint8_t* occ,
                             bool first_group_aside)
{
    hipStream_t st = rt_g().stream;
    const size_t npix = (size_t)W * nrows;
    const int L = f->num_lights;
    RT_TRY(workspaces(sc, npix * (size_t)L));
    RT_HIP(hipMemsetAsync(sc->ws_counters.p, 0, 64, st));
    // the apex sweep keeps its per-ray counts in cnt[] from the start (atomic updates)
    RT_HIP(hipMemsetAsync(sc->ws_cnt.p, 0, sizeof(int32_t) * npix * (size_t)L, st));
    static const bool use_cache = !(getenv("RT_AMD_SHADOW_CACHE") && atoi(getenv("RT_AMD_SHADOW_CACHE")) == 0);
    if (use_cache) {
        RT_TRY(sc->ws_occid.reserve(sizeof(int32_t) * npix * (size_t)L));
        RT_HIP(hipMemsetAsync(sc->ws_occid.p, 0xFF, sizeof(int32_t) * npix * (size_t)L, st));     // -1 = none
    }
    for (int l0 = 0; l0 < L; l0 += RT_APEX_LIGHTS) {
        const int lights = L - l0 < RT_APEX_LIGHTS ? L - l0 : RT_APEX_LIGHTS;
        RT_TRY(anyhit_buffers(sc, npix * 3 * (size_t)lights));
        if (l0 == 0 && first_group_aside) {
            RT_HIP(hipStreamWaitEvent(st, sc->ev_join, 0));     // prepare_lights_aside
        } else {
            RT_TRY(reserve_light_buffers(sc, lights));
            RT_TRY(prepare_lights(sc, f, l0, lights, (long)npix, st));
        }
        int32_t* order = sc->lights_order == 1 ? sc->tri_order.as<int32_t>()
                                               : (sc->lights_order == 2 ? sc->tri_scatter.as<int32_t>() : nullptr);
        SweepArgs a{};
        a.cull = rt_tile_culling() ? 1 : 0;
        a.tri_order = order;
        a.tri_order_stride = sc->lights_order == 1 ? sc->num_tris_padded : 0;
        a.planes = sc->planes_light.as<float4>(); a.plane_stride = sc->num_tris_padded;
        a.light_plane_stride = (long)RT_LIGHT_PLANES * sc->num_tris_padded;
        a.num_tris = sc->num_tris; a.num_tris_padded = sc->num_tris_padded;
        a.S = S + 3 * npix * (size_t)l0; a.E = E + 3 * npix * (size_t)l0;
        const int8_t* ax = axis ? axis + npix * (size_t)l0 : nullptr;   // the rays' dominant axes, written with them
        a.cnt = sc->ws_cnt.as<int32_t>() + npix * (size_t)l0;
        a.list = sc->ws_list.as<int32_t>() + npix * (size_t)l0 * RT_LIST_K;
        a.tests = reinterpret_cast<unsigned long long*>(sc->ws_counters.as<char>() + 8);
        const int sets = 3 * lights;
        int32_t* act0 = sc->ws_act0.as<int32_t>();
        int* live0 = sc->ws_live.as<int>();
        auto sweep = [&](const SweepArgs& r) { return rt_launch_sweep_apex(r, st); };
        // one any-hit phase over the lists binned into act0 / live0
        auto phase = [&]() -> int {
            int counts[RT_APEX_SETS];
            RT_TRY(binned_counts(sc, live0, sets, counts));
            long total = 0;
            for (int k = 0; k < sets; ++k) total += counts[k];
            if (use_cell_lattice(total, sc->num_tris_padded)) return anyhit_cells(sc, a, sets, (long)npix, act0, counts);
            return anyhit_ranges(sc, a, (int)npix, sets, (long)npix, true, sweep, counts);
        };
        RT_HIP(hipMemsetAsync(sc->ws_live.p, 0, sizeof(int) * 3 * RT_APEX_SETS, st));
        if (!use_cache) {
            RT_TRY(rt_launch_bin_rays(a.S, a.E, ax, hit, W, nrows, lights, (long)npix, 1, 0, act0, live0, sc->ws_bin_pad.as<int>(), st));
            RT_TRY(phase());
            continue;
        }
        // Occluder cache (k_try_cached), coarse to fine: the rays of every 4th pixel in x and y
        // are swept first; then all the others try the occluders found around them and what is
        // left of them is swept.
        RT_TRY(sc->ws_act2.reserve(sizeof(int32_t) * npix * (size_t)sets));
        int32_t* act2 = sc->ws_act2.as<int32_t>();
        int* live2 = live0 + 2 * RT_APEX_SETS;
        int32_t* occ_id = sc->ws_occid.as<int32_t>() + npix * (size_t)l0;
        for (int level = 0; level < RT_SHADOW_LEVELS; ++level) {
            const int step = RT_SHADOW_STEPS[level];
            const bool record = level + 1 < RT_SHADOW_LEVELS;       // the next level reads what this one finds
            a.occ_id = record ? occ_id : nullptr;
            RT_HIP(hipMemsetAsync(live0, 0, sizeof(int) * 3 * RT_APEX_SETS, st));
            if (level == 0) {
                RT_TRY(rt_launch_bin_rays(a.S, a.E, ax, hit, W, nrows, lights, (long)npix, step, 0, act0, live0, sc->ws_bin_pad.as<int>(), st));
            } else {
                const int coarser = RT_SHADOW_STEPS[level - 1];
                RT_TRY(rt_launch_bin_rays(a.S, a.E, ax, hit, W, nrows, lights, (long)npix, step, coarser, act2, live2, sc->ws_bin_pad.as<int>(), st));
                RT_TRY(rt_launch_try_cached(exact_of(sc), a.S, a.E, act2, live2, sets, (int)npix, (long)npix,
                                            W, nrows, row_block, coarser, record, a.cnt, occ_id, st));
                RT_TRY(rt_launch_compact(a.cnt, act2, live2, (int)npix, sets, (long)npix, 3, act0, live0,
                                         sc->ws_live.as<int>() + 3 * RT_APEX_SETS, st));
            }
            RT_TRY(phase());
        }
    }
    return resolve_rays(sc, scene_workspace(sc), (int)(npix * (size_t)L), S, E, true, nullptr, nullptr, nullptr, occ,
                        hit, (long)npix);
}

extern "C" int rt_trace_rays(rt_scene* sc, int n, const double* S, const double* E, int32_t* hit,
                             double* tuv, double* point)
{
    RT_TRY(require_ready(sc));
    RT_ARG(n >= 0, "negative ray count");
    RT_ARG(n == 0 || (S && E && hit), "null pointer");
    RT_TRY(trace_rays_impl(sc, nullptr, n, S, E, false, hit, tuv, point, nullptr));
    RT_HIP(hipStreamSynchronize(rt_g().stream));
    return RT_OK;
}

// rt_trace_rays on the static scratch array: no hipMalloc / hipFree, no grow-only buffer, no
// profiling read-backs.  The scene's own buffers (geometry, planes) were allocated when it was
// set up and are only read here.
extern "C" int rt_trace_rays_static(rt_scene* sc, int n, const double* S, const double* E, int32_t* hit,
                                    double* tuv, double* point)
{
    RT_TRY(require_ready(sc));
    RT_ARG(n >= 0, "negative ray count");
    RT_ARG(n == 0 || (S && E && hit), "null pointer");
    if (n > RT_STATIC_MAX_RAYS) {
        rt_set_error("rt_trace_rays_static: %d rays exceed the static workspace's %d; split the batch or "
                     "use rt_trace_rays", n, RT_STATIC_MAX_RAYS);
        return RT_E_LIMIT;
    }
    RtGlobal& g = rt_g();
    // the scene's own device: its copy of the array, and that device must be the current one
    // (rt_init may since have been run for another GPU of the process)
    int current = -1;
    RT_HIP(hipGetDevice(&current));
    if (current != sc->device) {
        rt_set_error("rt_trace_rays_static: the scene lives on device %d but device %d is current", sc->device, current);
        return RT_E_STATE;
    }
    unsigned char* base = sc->device >= 0 && sc->device < kMaxDevices ? g_scratch_ptr[sc->device] : nullptr;
    if (!base) {
        rt_set_error("rt_trace_rays_static: no static workspace on device %d (rt_init not run on it)", sc->device);
        return RT_E_STATE;
    }
    const RayWorkspace ws{reinterpret_cast<int32_t*>(base + kStaticOffCnt),
                          reinterpret_cast<int32_t*>(base + kStaticOffList),
                          reinterpret_cast<int32_t*>(base + kStaticOffOvf),
                          reinterpret_cast<int*>(base + kStaticOffCounters), base + kStaticOffPartials};
    const bool was_profiling = g.profiling;
    g.profiling = false;            // the profiling path reads counters back through staging copies
    const int rc = trace_rays_impl(sc, &ws, n, S, E, false, hit, tuv, point, nullptr);
    g.profiling = was_profiling;
    RT_TRY(rc);
    RT_HIP(hipStreamSynchronize(g.stream));
    return RT_OK;
}

extern "C" int rt_occluded(rt_scene* sc, int n, const double* S, const double* E, uint8_t* occ)
{
    RT_TRY(require_ready(sc));
    RT_ARG(n >= 0, "negative ray count");
    RT_ARG(n == 0 || (S && E && occ), "null pointer");
    RT_TRY(trace_rays_impl(sc, nullptr, n, S, E, true, nullptr, nullptr, nullptr, occ));
    RT_HIP(hipStreamSynchronize(rt_g().stream));
    return RT_OK;
}

// ---------------------------------------------------------------- whole pixel loop
static int render_rows_impl(rt_scene* sc, const rt_frame* f, int W, int H, double tan_half,
                            const RowMap& rows, int32_t* hit, double* tuv, double* argb, uint8_t* rgb8)
{
    RtGlobal& g = rt_g();
    hipStream_t st = g.stream;
    const size_t npix = (size_t)W * rows.rows;
    const int L = f->num_lights;

    if (!hit) { RT_TRY(sc->ws_hit.reserve(sizeof(int32_t) * npix)); hit = sc->ws_hit.as<int32_t>(); }
    if (!tuv) { RT_TRY(sc->ws_tuv.reserve(sizeof(double) * 3 * npix)); tuv = sc->ws_tuv.as<double>(); }
    // the first group of lights is sorted and packed on a side stream beside the primary sweep
    static const bool aside_env = !(getenv("RT_AMD_LIGHTS_ASIDE") && atoi(getenv("RT_AMD_LIGHTS_ASIDE")) == 0);
    const bool aside = L > 0 && aside_env;
    // Work queued on the side stream reads the eye-space vertices and writes the lights' planes.  The
    // any-hit phase makes the launch stream wait for it; when the call fails before that join
    // (a reserve, a launch), the side stream is drained here, so that whatever the caller does
    // next on the launch stream cannot race with it.
    struct SideJoin {
        rt_scene* sc;
        bool armed;
        ~SideJoin() { if (armed && sc->side_stream) (void)hipStreamSynchronize(sc->side_stream); }
    } side_join{sc, aside};
    // frame constants: uploaded ahead of the sweep, so that nothing between the sweep and the
    // shading kernels has to wait for the host
    DevBuf& frame_dev = sc->ws_frame;
    RT_TRY(frame_dev.reserve(sizeof(DevFrame)));
    sc->frame_host.resize(sizeof(DevFrame));        // kept with the scene: outlives the async copy below
    DevFrame& hf = *reinterpret_cast<DevFrame*>(sc->frame_host.data());
    hf.num_lights = L;
    hf.pad = 0;
    memcpy(hf.obinv, f->obinv, sizeof hf.obinv);
    memcpy(hf.light_pos, f->light_in_eye_space, sizeof hf.light_pos);
    memcpy(hf.light_rgb, f->light_rgb, sizeof hf.light_rgb);
    memcpy(hf.light_att, f->light_attenuation, sizeof hf.light_att);
    memcpy(hf.light_max, f->light_max_dist, sizeof hf.light_max);
    RT_HIP(hipMemcpyAsync(frame_dev.p, &hf, sizeof hf, hipMemcpyHostToDevice, st));

    RT_TRY(trace_primary_rows(sc, W, H, tan_half, rows, hit, tuv, aside ? f : nullptr,
                              L < RT_APEX_LIGHTS ? L : RT_APEX_LIGHTS, false));

    ShadeArgs a{};
    a.es = exact_of(sc);
    a.An = sc->iAn.as<int>(); a.Bn = sc->iBn.as<int>(); a.Cn = sc->iCn.as<int>();
    a.At = sc->has_tex_idx ? sc->iAt.as<int>() : nullptr;
    a.Bt = sc->has_tex_idx ? sc->iBt.as<int>() : nullptr;
    a.Ct = sc->has_tex_idx ? sc->iCt.as<int>() : nullptr;
    a.mtl = sc->imtl.p ? sc->imtl.as<int>() : nullptr;
    a.nx = sc->nx.as<double>(); a.ny = sc->ny.as<double>(); a.nz = sc->nz.as<double>();
    if (!sc->tri_normals_ready) {       // once per mesh: the normals do not move with the transform
        RT_TRY(sc->tri_normals.reserve(sizeof(double) * 9 * (size_t)(sc->num_tris > 0 ? sc->num_tris : 1)));
        RT_TRY(rt_launch_pack_normals(a.nx, a.ny, a.nz, a.An, a.Bn, a.Cn, sc->num_tris, sc->tri_normals.as<double>(), st));
        sc->tri_normals_ready = true;
    }
    a.tn = sc->tri_normals.as<double>();
    a.tu = sc->tu.as<double>(); a.tv = sc->tv.as<double>();
    a.mats = sc->mats.as<DevMaterial>(); a.tex = sc->tex_desc.as<DevTexture>();
    a.num_mats = sc->num_mats;
    a.frame = frame_dev.as<DevFrame>();
    a.num_lights = L;
    a.W = W; a.H = H; a.rows = rows; a.tan_half = tan_half;
    a.hit = hit; a.tuv = tuv;
    a.argb = argb; a.rgb8 = rgb8;
    if (L > 0) {
        const size_t nsh = npix * (size_t)L;
        RT_ARG(nsh < (size_t)1 << 31, "band * lights exceeds 2^31 shadow rays; render fewer rows per call");
        RT_TRY(sc->ws_shS.reserve(sizeof(double) * 3 * nsh));
        RT_TRY(sc->ws_shE.reserve(sizeof(double) * 3 * nsh));
        RT_TRY(sc->ws_occ.reserve(nsh));
        RT_TRY(sc->ws_axis.reserve(nsh));
        a.shS = sc->ws_shS.as<double>(); a.shE = sc->ws_shE.as<double>();
        a.sh_axis = sc->ws_axis.as<int8_t>();
        a.occ = sc->ws_occ.as<uint8_t>();
        RT_TRY(rt_launch_shadow_setup(a, st));
        RT_TRY(trace_shadow_rays(sc, f, W, rows.rows, rows.parts > 1 ? rows.block : rows.rows, a.shS, a.shE, a.sh_axis,
                                 hit, sc->ws_occ.as<uint8_t>(), aside));
    }
    RT_TRY(rt_launch_shade_final(a, st));
    RT_HIP(hipStreamSynchronize(st));
    side_join.armed = false;        // joined by the any-hit phase, and everything has completed
    return RT_OK;
}

extern "C" int rt_render_rows_part(rt_scene* sc, const rt_frame* f, int W, int H, double tan_half,
                                   const rt_rows* rows, int32_t* hit, double* tuv, double* argb,
                                   uint8_t* rgb8)
{
    RT_TRY(require_ready(sc));
    RT_ARG(f, "null frame");
    RT_ARG(f->num_lights >= 0 && f->num_lights <= RT_MAX_LIGHTS, "num_lights outside [0,100]");
    RT_ARG(W > 0 && H > 0, "bad image size");
    RT_ARG(tan_half > 0, "tan_half must be positive");
    RT_ARG(sc->iAn.p && sc->nx.p, "scene has no normals (needed for shading)");
    // the shading kernel indexes these tables without checks of its own
    if (sc->mtl_max >= sc->num_mats) {
        rt_set_error("a triangle uses material %d but only %d are set", sc->mtl_max, sc->num_mats);
        return RT_E_STATE;
    }
    if (sc->mtl_max >= 0 && sc->map_max >= 0) {
        if (sc->map_max >= sc->num_tex) {
            rt_set_error("a material uses texture %d but only %d are set", sc->map_max, sc->num_tex);
            return RT_E_STATE;
        }
        if (!sc->has_tex_idx || !sc->tu.p || !sc->tv.p) {
            rt_set_error("a textured material needs texture coordinates and At/Bt/Ct on every triangle");
            return RT_E_STATE;
        }
    }
    RowMap rm;
    RT_TRY(row_map(rows, H, &rm));
    if (rm.rows == 0) return RT_OK;
    return render_rows_impl(sc, f, W, H, tan_half, rm, hit, tuv, argb, rgb8);
}

extern "C" int rt_render_rows(rt_scene* sc, const rt_frame* f, int W, int H, double tan_half,
                              int y0, int y1, int32_t* hit, double* tuv, double* argb,
                              uint8_t* rgb8)
{
    RT_ARG(y0 < y1, "bad image size or row band");
    const rt_rows r = {y0, y1, y1 - y0, 1, 0};
    return rt_render_rows_part(sc, f, W, H, tan_half, &r, hit, tuv, argb, rgb8);
}

extern "C" int rt_place_rows_part(const uint8_t* rgb8, int W, int H, const rt_rows* rows, uint8_t* screen)
{
    RT_ARG(W > 0 && H > 0, "bad image size");
    RowMap rm;
    RT_TRY(row_map(rows, H, &rm));
    if (rm.rows == 0) return RT_OK;         // a part may be empty (fewer rows than parts)
    RT_ARG(rgb8 && screen, "null pointer");
    RT_TRY(rt_ensure_init());
    RT_TRY(rt_launch_place_rows(rgb8, W, H, rm, screen, rt_g().stream));
    RT_HIP(hipStreamSynchronize(rt_g().stream));
    return RT_OK;
}

extern "C" int rt_place_rows(const uint8_t* rgb8, int W, int H, int y0, int y1, uint8_t* screen)
{
    RT_ARG(y0 <= y1, "bad image size or row band");
    const rt_rows r = {y0, y1, y1 - y0 > 0 ? y1 - y0 : 1, 1, 0};
    return rt_place_rows_part(rgb8, W, H, &r, screen);
}

// ---------------------------------------------------------------- the reference's kernel.h API
namespace {

struct LegacyState {
    DevBuf m, x, y, z;                 // mat_mult_device staging
    // cached mesh of intersect_all_triangles_device
    const void* tris_host = nullptr;
    const double *xh = nullptr, *yh = nullptr, *zh = nullptr;
    int num_tris = -1, num_pts = 0;
    bool valid = false;
    DevBuf iA, iB, iC, mx, my, mz, io;   // io: S,E in / result + partials out
};
LegacyState& legacy()
{
    static LegacyState s;
    return s;
}

bool overlaps(const double* a, size_t na, const double* b, size_t nb)
{
    if (!a || !b) return false;
    const char* a0 = reinterpret_cast<const char*>(a);
    const char* b0 = reinterpret_cast<const char*>(b);
    return a0 < b0 + nb * sizeof(double) && b0 < a0 + na * sizeof(double);
}

void legacy_fail(const char* what)
{
    // The reference API has no error channel (kernel.h:10-17).  A device failure here must
    // not pass silently as a plausible result, so it is reported and the call is fatal.
    fprintf(stderr, "rt_amd: %s: %s\n", what, rt_last_error());
    abort();
}

}  // namespace

extern "C" void rt_legacy_invalidate(void) { legacy().valid = false; }

extern "C" void mat_mult_device(double* X, double* Y, double* Z, double m[4][4], double* x,
                                double* y, double* z, int numpoints)
{
    if (numpoints <= 0) return;
    if (rt_ensure_init() != RT_OK) legacy_fail("mat_mult_device");
    LegacyState& s = legacy();
    hipStream_t st = rt_g().stream;
    const size_t bytes = sizeof(double) * (size_t)numpoints;
    auto ok = [&](hipError_t e) {
        if (e != hipSuccess) {
            rt_set_error("%s", hipGetErrorString(e));
            legacy_fail("mat_mult_device");
        }
    };
    if (s.m.reserve(128) || s.x.reserve(bytes) || s.y.reserve(bytes) || s.z.reserve(bytes))
        legacy_fail("mat_mult_device");
    ok(hipMemcpyAsync(s.m.p, m, 128, hipMemcpyHostToDevice, st));
    ok(hipMemcpyAsync(s.x.p, x, bytes, hipMemcpyHostToDevice, st));
    ok(hipMemcpyAsync(s.y.p, y, bytes, hipMemcpyHostToDevice, st));
    ok(hipMemcpyAsync(s.z.p, z, bytes, hipMemcpyHostToDevice, st));
    if (rt_launch_transform(s.m.as<double>(), s.x.as<double>(), s.y.as<double>(), s.z.as<double>(),
                            s.x.as<double>(), s.y.as<double>(), s.z.as<double>(), numpoints, nullptr,
                            st) != RT_OK)
        legacy_fail("mat_mult_device");
    ok(hipMemcpyAsync(X, s.x.p, bytes, hipMemcpyDeviceToHost, st));
    ok(hipMemcpyAsync(Y, s.y.p, bytes, hipMemcpyDeviceToHost, st));
    ok(hipMemcpyAsync(Z, s.z.p, bytes, hipMemcpyDeviceToHost, st));
    ok(hipStreamSynchronize(st));
    // the vertex arrays behind the cached mesh were rewritten (driver.c:110,202)
    if (s.valid && (overlaps(X, numpoints, s.xh, s.num_pts) || overlaps(Y, numpoints, s.yh, s.num_pts) ||
                    overlaps(Z, numpoints, s.zh, s.num_pts)))
        s.valid = false;
}

extern "C" int intersect_all_triangles_device(double S[3], double E[3], double uv[2],
                                              double point[3], Triangle* tris, double* x, double* y,
                                              double* z, int num_tris)
{
    if (num_tris <= 0) return -1;
    if (rt_ensure_init() != RT_OK) legacy_fail("intersect_all_triangles_device");
    LegacyState& s = legacy();
    hipStream_t st = rt_g().stream;
    auto ok = [&](hipError_t e) {
        if (e != hipSuccess) {
            rt_set_error("%s", hipGetErrorString(e));
            legacy_fail("intersect_all_triangles_device");
        }
    };
    if (!s.valid || s.tris_host != tris || s.xh != x || s.yh != y || s.zh != z || s.num_tris != num_tris) {
        // (re)build the device copy.  The API carries no vertex count (kernel.h:16-17); the
        // arrays must reach at least the largest index any triangle uses.
        std::vector<int> idx[9], mtl;
        split_aos(tris, num_tris, idx, mtl);
        int maxi = 0;
        for (int k = 0; k < 3; ++k)
            for (int i = 0; i < num_tris; ++i) {
                if (idx[k][i] < 0) {
                    rt_set_error("negative vertex index in triangle %d", i);
                    legacy_fail("intersect_all_triangles_device");
                }
                maxi = idx[k][i] > maxi ? idx[k][i] : maxi;
            }
        const size_t ib = sizeof(int) * (size_t)num_tris, vb = sizeof(double) * (size_t)(maxi + 1);
        if (s.iA.reserve(ib) || s.iB.reserve(ib) || s.iC.reserve(ib) || s.mx.reserve(vb) ||
            s.my.reserve(vb) || s.mz.reserve(vb) || s.io.reserve(64 + 64 + 32 * 256))
            legacy_fail("intersect_all_triangles_device");
        ok(hipMemcpyAsync(s.iA.p, idx[0].data(), ib, hipMemcpyHostToDevice, st));
        ok(hipMemcpyAsync(s.iB.p, idx[1].data(), ib, hipMemcpyHostToDevice, st));
        ok(hipMemcpyAsync(s.iC.p, idx[2].data(), ib, hipMemcpyHostToDevice, st));
        ok(hipMemcpyAsync(s.mx.p, x, vb, hipMemcpyHostToDevice, st));
        ok(hipMemcpyAsync(s.my.p, y, vb, hipMemcpyHostToDevice, st));
        ok(hipMemcpyAsync(s.mz.p, z, vb, hipMemcpyHostToDevice, st));
        ok(hipStreamSynchronize(st));      // idx[] vectors die at the end of this block
        s.tris_host = tris; s.xh = x; s.yh = y; s.zh = z;
        s.num_tris = num_tris; s.num_pts = maxi + 1;
        s.valid = true;
    }
    double se[6] = {S[0], S[1], S[2], E[0], E[1], E[2]};
    double* io = s.io.as<double>();
    ok(hipMemcpyAsync(io, se, sizeof se, hipMemcpyHostToDevice, st));
    ExactScene es;
    es.x = s.mx.as<double>(); es.y = s.my.as<double>(); es.z = s.mz.as<double>();
    es.A = s.iA.as<int>(); es.B = s.iB.as<int>(); es.C = s.iC.as<int>();
    es.num_tris = num_tris;
    es.tv = nullptr;
    if (rt_launch_single_ray(es, io, io + 8, st) != RT_OK) legacy_fail("intersect_all_triangles_device");
    double out[7];
    ok(hipMemcpyAsync(out, io + 8, sizeof out, hipMemcpyDeviceToHost, st));
    ok(hipStreamSynchronize(st));
    const int closest = (int)out[0];
    if (closest != -1) {
        uv[0] = out[2]; uv[1] = out[3];
        point[0] = out[4]; point[1] = out[5]; point[2] = out[6];
    }
    return closest;
}
