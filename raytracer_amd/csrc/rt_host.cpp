// rt_host.cpp -- host-side pieces of the path that need no GPU: the per-frame camera / light
// set-up of the reference driver, the 4x4 helpers it uses, row-band arithmetic for multi-GPU
// runs, and a BMP writer compatible with the reference's screenshots.
//
// The matrix routines reproduce M3d_matrix_tools.c operation for operation (same products,
// same summation order, sums started from 0.0), so their outputs are bit-identical to the
// reference's when both are compiled without FMA contraction; tests check this against the
// reference's own file compiled unmodified.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "rt_amd.h"

void rt_set_error(const char* fmt, ...);

namespace {

struct Mat4 {
    double a[4][4];
    static Mat4 identity()                                     // M3d_matrix_tools.c:57-69
    {
        Mat4 m;
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) m.a[r][c] = r == c ? 1.0 : 0.0;
        return m;
    }
    static Mat4 translation(double dx, double dy, double dz)   // :75-80
    {
        Mat4 m = identity();
        m.a[0][3] = dx; m.a[1][3] = dy; m.a[2][3] = dz;
        return m;
    }
    static Mat4 scaling(double sx, double sy, double sz)       // :86-91
    {
        Mat4 m = identity();
        m.a[0][0] = sx; m.a[1][1] = sy; m.a[2][2] = sz;
        return m;
    }
    // rotation about axis 0/1/2 (x/y/z) from a known cosine and sine, :93-127
    static Mat4 rotation(int axis, double cs, double sn)
    {
        Mat4 m = identity();
        const int i = (axis + 1) % 3, j = (axis + 2) % 3;     // x: (1,2)  y: (2,0)  z: (0,1)
        m.a[i][i] = cs; m.a[i][j] = -sn;
        m.a[j][i] = sn; m.a[j][j] = cs;
        return m;
    }
};

// :133-157 -- each entry is 0.0 + a[r][0]*b[0][c] + ... + a[r][3]*b[3][c], in that order
Mat4 operator*(const Mat4& a, const Mat4& b)
{
    Mat4 res;
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
            double sum = 0.0;
            for (int k = 0; k < 4; ++k) sum = sum + a.a[r][k] * b.a[k][c];
            res.a[r][c] = sum;
        }
    return res;
}

void apply(const Mat4& m, const double q[3], double p[3])      // :163-178
{
    const double u = m.a[0][0] * q[0] + m.a[0][1] * q[1] + m.a[0][2] * q[2] + m.a[0][3];
    const double v = m.a[1][0] * q[0] + m.a[1][1] * q[1] + m.a[1][2] * q[2] + m.a[1][3];
    const double t = m.a[2][0] * q[0] + m.a[2][1] * q[1] + m.a[2][2] * q[2] + m.a[2][3];
    p[0] = u; p[1] = v; p[2] = t;
}

bool movement_sequence(Mat4& mat, Mat4& inv, int n, const int* types, const double* params)
{
    mat = Mat4::identity();
    inv = Mat4::identity();
    for (int i = 0; i < n; ++i) {
        const int code = types[i];
        const double p = params[i];
        Mat4 fwd, bwd;
        if (code >= 0 && code < 3) {                            // scale, :354-376
            double s[3] = {1.0, 1.0, 1.0}, si[3] = {1.0, 1.0, 1.0};
            s[code] = p;
            si[code] = 1.0 / p;
            fwd = Mat4::scaling(s[0], s[1], s[2]);
            bwd = Mat4::scaling(si[0], si[1], si[2]);
        } else if (code < 6) {                                  // rotate (degrees), :381-403
            const double rad = p * M_PI / 180, nrad = -p * M_PI / 180;
            fwd = Mat4::rotation(code - 3, std::cos(rad), std::sin(rad));
            bwd = Mat4::rotation(code - 3, std::cos(nrad), std::sin(nrad));
        } else if (code < 9) {                                  // translate, :408-430
            double d[3] = {0.0, 0.0, 0.0}, di[3] = {0.0, 0.0, 0.0};
            d[code - 6] = p;
            di[code - 6] = -p;
            fwd = Mat4::translation(d[0], d[1], d[2]);
            bwd = Mat4::translation(di[0], di[1], di[2]);
        } else if (code < 12) {                                 // reflect, :436-455
            double s[3] = {1.0, 1.0, 1.0};
            s[code - 9] = -1.0;
            fwd = bwd = Mat4::scaling(s[0], s[1], s[2]);
        } else {
            return false;
        }
        mat = fwd * mat;
        inv = inv * bwd;
    }
    return true;
}

// M3d_view (:591-628) with its two variants (:484-527 when |b| < p, :536-581 otherwise).
bool view(Mat4& V, Mat4& Vi, const double eye[3], const double coi[3], const double up[3])
{
    const double a = coi[0] - eye[0], b = coi[1] - eye[1], c = coi[2] - eye[2];
    double p = std::sqrt(a * a + c * c);
    const double r = std::sqrt(a * a + b * b + c * c);
    const bool yaw_first = std::fabs(b) < p;
    if (!yaw_first) p = std::sqrt(b * b + c * c);
    if (p == 0 || r == 0) return false;

    const Mat4 T = Mat4::translation(-eye[0], -eye[1], -eye[2]);
    const Mat4 Ti = Mat4::translation(eye[0], eye[1], eye[2]);
    Mat4 R1, R2, R1i, R2i;
    if (yaw_first) {
        R1 = Mat4::rotation(1, c / p, -a / p);  R1i = Mat4::rotation(1, c / p, a / p);
        R2 = Mat4::rotation(0, p / r, b / r);   R2i = Mat4::rotation(0, p / r, -b / r);
    } else {
        R1 = Mat4::rotation(0, c / p, b / p);   R1i = Mat4::rotation(0, c / p, -b / p);
        R2 = Mat4::rotation(1, p / r, -a / r);  R2i = Mat4::rotation(1, p / r, a / r);
    }
    V = T * Mat4::identity();
    V = R1 * V;
    V = R2 * V;
    double Up[3];
    apply(V, up, Up);
    const double Uplen = std::sqrt(Up[0] * Up[0] + Up[1] * Up[1]);
    if (Uplen == 0) return false;
    V = Mat4::rotation(2, Up[1] / Uplen, Up[0] / Uplen) * V;

    Vi = Mat4::rotation(2, Up[1] / Uplen, -Up[0] / Uplen) * Mat4::identity();
    Vi = R2i * Vi;
    Vi = R1i * Vi;
    Vi = Ti * Vi;
    return true;
}

}  // namespace

extern "C" int rt_mat_mult(double res[4][4], const double a[4][4], const double b[4][4])
{
    if (!res || !a || !b) { rt_set_error("rt_mat_mult: null pointer"); return RT_E_ARG; }
    Mat4 A, B;
    memcpy(A.a, a, sizeof A.a);
    memcpy(B.a, b, sizeof B.a);
    const Mat4 R = A * B;
    memcpy(res, R.a, sizeof R.a);
    return RT_OK;
}

extern "C" int rt_mat_mult_pt(double P[3], const double m[4][4], const double Q[3])
{
    if (!P || !m || !Q) { rt_set_error("rt_mat_mult_pt: null pointer"); return RT_E_ARG; }
    Mat4 M;
    memcpy(M.a, m, sizeof M.a);
    const double q[3] = {Q[0], Q[1], Q[2]};
    apply(M, q, P);
    return RT_OK;
}

extern "C" int rt_movement_sequence(double mat[4][4], double inv[4][4], int n, const int* types,
                                    const double* params)
{
    if (!mat || !inv || n < 0 || (n > 0 && (!types || !params))) {
        rt_set_error("rt_movement_sequence: bad argument");
        return RT_E_ARG;
    }
    Mat4 M, I;
    if (!movement_sequence(M, I, n, types, params)) {
        rt_set_error("rt_movement_sequence: unrecognised movement code");
        return RT_E_ARG;
    }
    memcpy(mat, M.a, sizeof M.a);
    memcpy(inv, I.a, sizeof I.a);
    return RT_OK;
}

extern "C" int rt_view(double view_m[4][4], double view_inverse[4][4], const double eye[3],
                       const double coi[3], const double up[3])
{
    if (!view_m || !view_inverse || !eye || !coi || !up) {
        rt_set_error("rt_view: null pointer");
        return RT_E_ARG;
    }
    Mat4 V, Vi;
    if (!view(V, Vi, eye, coi, up)) {
        rt_set_error("rt_view: degenerate eye / centre of interest / up");
        return RT_E_ARG;
    }
    memcpy(view_m, V.a, sizeof V.a);
    memcpy(view_inverse, Vi.a, sizeof Vi.a);
    return RT_OK;
}

// driver.c:34-107
extern "C" int rt_driver_frame_setup(int frame_number, int frames_per_turn, rt_frame* out)
{
    if (!out || frames_per_turn <= 0) {
        rt_set_error("rt_driver_frame_setup: bad argument");
        return RT_E_ARG;
    }
    memset(out, 0, sizeof *out);
    const double eangle = 2 * M_PI * frame_number / frames_per_turn;
    const double eye[3] = {15.0 * std::cos(eangle), 10.0, 15.0 * std::sin(eangle)};
    const double coi[3] = {0, 0, 0};
    const double up[3] = {eye[0], eye[1] + 1, eye[2]};
    Mat4 vm, vi;
    if (!view(vm, vi, eye, coi, up)) {
        rt_set_error("rt_driver_frame_setup: view matrix is degenerate");
        return RT_E_ARG;
    }
    // light 0: white, rising with the frame number (driver.c:63-77)
    const double l0[3] = {0.0, 15.0 + frame_number, 0.0};
    out->light_rgb[0][0] = 1.0; out->light_rgb[0][1] = 1.0; out->light_rgb[0][2] = 1.0;
    out->light_attenuation[0][0] = 0.045; out->light_attenuation[0][1] = 0.0075;
    out->light_max_dist[0] = 100;
    apply(vm, l0, out->light_in_eye_space[0]);
    // light 1: violet, fixed (driver.c:79-92)
    const double l1[3] = {-10, 10, 0};
    out->light_rgb[1][0] = 0.1; out->light_rgb[1][1] = 0.0; out->light_rgb[1][2] = 1.0;
    out->light_attenuation[1][0] = 0.09; out->light_attenuation[1][1] = 0.032;
    out->light_max_dist[1] = 50;
    apply(vm, l1, out->light_in_eye_space[1]);
    out->num_lights = 2;

    const int types[1] = {3};          // RX, M3d_matrix_tools.h:64
    const double params[1] = {270};
    Mat4 m, mi;
    movement_sequence(m, mi, 1, types, params);
    const Mat4 obmat = vm * m, obinv = mi * vi;                 // driver.c:106-107
    memcpy(out->obmat, obmat.a, sizeof obmat.a);
    memcpy(out->obinv, obinv.a, sizeof obinv.a);
    return RT_OK;
}

// Contiguous bands; the first H % n ranks get one extra row.
extern "C" int rt_band(int H, int nranks, int rank, int* y0, int* y1)
{
    if (H < 0 || nranks <= 0 || rank < 0 || rank >= nranks || !y0 || !y1) {
        rt_set_error("rt_band: bad argument");
        return RT_E_ARG;
    }
    const int base = H / nranks, extra = H % nranks;
    *y0 = rank * base + (rank < extra ? rank : extra);
    *y1 = *y0 + base + (rank < extra ? 1 : 0);
    return RT_OK;
}

// 32 bits per pixel, BI_BITFIELDS with the masks of graphic_tools.c:19-23, bottom-up rows.
extern "C" int rt_save_bmp(const char* path, const uint8_t* screen_rgb, int W, int H)
{
    if (!path || !screen_rgb || W <= 0 || H <= 0) {
        rt_set_error("rt_save_bmp: bad argument");
        return RT_E_ARG;
    }
    FILE* f = fopen(path, "wb");
    if (!f) {
        rt_set_error("rt_save_bmp: cannot open %s", path);
        return RT_E_IO;
    }
    const uint32_t header = 14 + 108, image = (uint32_t)W * (uint32_t)H * 4;
    std::vector<uint8_t> h(header, 0);
    auto put16 = [&](size_t o, uint32_t v) { h[o] = v & 255; h[o + 1] = (v >> 8) & 255; };
    auto put32 = [&](size_t o, uint32_t v) { put16(o, v & 0xffff); put16(o + 2, v >> 16); };
    h[0] = 'B'; h[1] = 'M';
    put32(2, header + image);
    put32(10, header);
    put32(14, 108);                 // BITMAPV4HEADER
    put32(18, (uint32_t)W);
    put32(22, (uint32_t)H);
    put16(26, 1);
    put16(28, 32);
    put32(30, 3);                   // BI_BITFIELDS
    put32(34, image);
    put32(38, 2835); put32(42, 2835);
    put32(54, 0x00FF0000u); put32(58, 0x0000FF00u); put32(62, 0x000000FFu); put32(66, 0xFF000000u);
    h[70] = 'B'; h[71] = 'G'; h[72] = 'R'; h[73] = 's';      // LCS_sRGB
    bool ok = fwrite(h.data(), 1, h.size(), f) == h.size();
    std::vector<uint8_t> row((size_t)W * 4);
    for (int y = H - 1; y >= 0 && ok; --y) {
        const uint8_t* src = screen_rgb + (size_t)y * W * 3;
        for (int x = 0; x < W; ++x) {
            row[4 * x + 0] = src[3 * x + 2];
            row[4 * x + 1] = src[3 * x + 1];
            row[4 * x + 2] = src[3 * x + 0];
            row[4 * x + 3] = 0xFF;
        }
        ok = fwrite(row.data(), 1, row.size(), f) == row.size();
    }
    ok = (fclose(f) == 0) && ok;
    if (!ok) {
        rt_set_error("rt_save_bmp: short write to %s", path);
        return RT_E_IO;
    }
    return RT_OK;
}
