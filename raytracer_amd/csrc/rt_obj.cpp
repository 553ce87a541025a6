// rt_obj.cpp -- Wavefront .obj / .mtl loader with the reference's semantics, emitting
// structure-of-arrays buffers (page-locked when a GPU is present) instead of the reference's
// 488-byte array-of-structures with a Material copy per triangle.
//
// Behaviour follows obj_reader.c:
//   * tokens are whitespace-delimited and read like fscanf("%s") does, across line ends (:664);
//   * v / vt / vn append to 1-indexed arrays (:666-680; slot 0 exists and is zero here,
//     uninitialised in the reference);
//   * the face format is sniffed from the first vertex (:27-48): 1 = v, 2 = v/vt,
//     3 = v/vt/vn, 4 = v//vn;
//   * a fourth vertex makes a quad, split (a,b,c) + (a,c,d) (:167-231); a fifth is an error
//     (:500-505; the reference exits the process, this loader returns RT_E_IO);
//   * formats without vn get ONE un-normalised face normal AB x AC, computed from the
//     coordinates as read (before centring/scaling) and shared by both halves of a quad
//     (:51-82,180,196);
//   * `usemtl` selects the material copied into following triangles (:84-90,708-716), here an
//     index into the material table; -1 until the first successful usemtl (:18);
//   * after parsing, vertices are centred on their mean and scaled so the largest bounding
//     box extent is 15 (:239-289), slot 0 included (:288).
// One evident defect is not reproduced: for format 1 the reference stores the ADDRESS of its
// scratch array as the vertex index of the first triangle of each face (:98-100); the
// index itself is stored here.
//
// Image maps: the reference decodes them with SDL_image (:352-381) and keeps a NULL map when
// decoding fails.  This loader decodes PNG (every colour type and bit depth, interlaced or not,
// through zlib; what the texture fetch of graphic_tools.c:161-197 sees is the pixel's RGB as
// SDL_GetRGB reports it: palette entries resolved, grey replicated, alpha ignored, 16-bit
// samples cut to their high byte), baseline JPEG (libjpeg's default decoding stages, see
// JpegDecoder), binary PPM (P6) and uncompressed 24/32-bit BMP.  Any other format (progressive
// JPEG among them) is treated like a failed IMG_Load: the material keeps its constant colour.
#include <hip/hip_runtime_api.h>
#include <zlib.h>

#include <cctype>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "rt_amd.h"

void rt_set_error(const char* fmt, ...);

namespace {

// ---------------------------------------------------------------- fscanf-like cursor
struct Cursor {
    const char* p;
    const char* end;
    bool eof() const { return p >= end; }
    void skip_ws() { while (p < end && isspace((unsigned char)*p)) ++p; }
    bool token(std::string& out)                     // "%s"
    {
        skip_ws();
        if (eof()) return false;
        const char* s = p;
        while (p < end && !isspace((unsigned char)*p)) ++p;
        out.assign(s, p);
        return true;
    }
    bool integer(int& v)                             // "%d": on failure nothing past the
    {                                                // leading white space is consumed
        skip_ws();
        const char* s = p;
        if (s < end && (*s == '+' || *s == '-')) ++s;
        if (s >= end || !isdigit((unsigned char)*s)) return false;
        char* e = nullptr;
        const long val = strtol(p, &e, 10);
        if (e == p) return false;
        v = (int)val;
        p = e;
        return true;
    }
    bool real(double& v)                             // "%lf"
    {
        skip_ws();
        if (eof()) return false;
        const char c = *p;
        if (!(isdigit((unsigned char)c) || c == '+' || c == '-' || c == '.')) return false;
        char* e = nullptr;
        const double val = strtod(p, &e);
        if (e == p) return false;
        v = val;
        p = e;
        return true;
    }
    bool literal(char c)                             // a literal in a scanf format: no skipping
    {
        if (p < end && *p == c) { ++p; return true; }
        return false;
    }
    void skip_line()                                 // fgets to the end of the line
    {
        while (p < end && *p != '\n') ++p;
        if (p < end) ++p;
    }
};

bool read_file(const std::string& path, std::string& out)
{
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    char buf[1 << 16];
    size_t n;
    out.clear();
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) out.append(buf, n);
    fclose(f);
    return true;
}

// ---------------------------------------------------------------- images
struct Image {
    int w = 0, h = 0;
    std::vector<uint8_t> rgb;
};
// A texture file is untrusted input: its header may declare any size.  No decoder allocates for
// more pixels than this, nor for more than the bytes present could possibly encode.
constexpr size_t kMaxTexturePixels = (size_t)64 << 20;

bool load_ppm(const std::string& data, Image& im)
{
    Cursor c{data.data(), data.data() + data.size()};
    std::string tok;
    if (!c.token(tok) || tok != "P6") return false;
    int vals[3], got = 0;
    while (got < 3) {
        c.skip_ws();
        if (!c.eof() && *c.p == '#') { c.skip_line(); continue; }
        if (!c.integer(vals[got])) return false;
        ++got;
    }
    if (vals[0] <= 0 || vals[1] <= 0 || vals[2] != 255) return false;
    if ((size_t)vals[0] * (size_t)vals[1] > kMaxTexturePixels) return false;
    ++c.p;      // the single white-space byte after maxval
    const size_t need = (size_t)vals[0] * vals[1] * 3;
    if ((size_t)(c.end - c.p) < need) return false;
    im.w = vals[0]; im.h = vals[1];
    im.rgb.assign((const uint8_t*)c.p, (const uint8_t*)c.p + need);
    return true;
}

bool load_bmp(const std::string& d, Image& im)
{
    if (d.size() < 54 || d[0] != 'B' || d[1] != 'M') return false;
    auto u32 = [&](size_t o) {
        return (uint32_t)(uint8_t)d[o] | (uint32_t)(uint8_t)d[o + 1] << 8 |
               (uint32_t)(uint8_t)d[o + 2] << 16 | (uint32_t)(uint8_t)d[o + 3] << 24;
    };
    const uint32_t off = u32(10);
    const int32_t w = (int32_t)u32(18), hs = (int32_t)u32(22);
    const int bpp = (uint8_t)d[28] | (uint8_t)d[29] << 8;
    const uint32_t comp = u32(30);
    if (w <= 0 || hs == 0 || (bpp != 24 && bpp != 32) || (comp != 0 && comp != 3)) return false;
    if (hs == INT32_MIN) return false;
    const int h = hs < 0 ? -hs : hs;
    if ((size_t)w * (size_t)h > kMaxTexturePixels) return false;
    const size_t stride = ((size_t)w * (bpp / 8) + 3) & ~(size_t)3;
    if (d.size() < off + stride * (size_t)h) return false;
    im.w = w; im.h = h;
    im.rgb.resize((size_t)w * h * 3);
    for (int y = 0; y < h; ++y) {
        const int src_row = hs < 0 ? y : h - 1 - y;          // top row first in `im`
        const uint8_t* s = (const uint8_t*)d.data() + off + stride * (size_t)src_row;
        for (int x = 0; x < w; ++x) {
            const uint8_t* px = s + (size_t)x * (bpp / 8);
            uint8_t* o = &im.rgb[((size_t)y * w + x) * 3];
            o[0] = px[2]; o[1] = px[1]; o[2] = px[0];
        }
    }
    return true;
}

// PNG (ISO/IEC 15948): signature, IHDR, optional PLTE, IDAT stream (zlib), IEND.  Scanlines are
// filtered per byte (None, Sub, Up, Average, Paeth); Adam7 interlacing stores seven reduced
// images one after the other.  Chunk CRCs are not verified (zlib checks the pixel stream's
// Adler-32).
bool load_png(const std::string& d, Image& im)
{
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (d.size() < 8 + 25 || memcmp(d.data(), sig, 8) != 0) return false;
    const uint8_t* p = (const uint8_t*)d.data();
    auto be32 = [&](size_t o) {
        return (uint32_t)p[o] << 24 | (uint32_t)p[o + 1] << 16 | (uint32_t)p[o + 2] << 8 | (uint32_t)p[o + 3];
    };
    uint32_t W = 0, H = 0;
    int depth = 0, ctype = -1, interlace = 0;
    std::vector<uint8_t> plte, idat;
    bool have_ihdr = false, done = false;
    for (size_t o = 8; !done && o + 12 <= d.size();) {
        const uint32_t len = be32(o);
        if (len > d.size() - o - 12) return false;
        const char* type = d.data() + o + 4;
        const uint8_t* body = p + o + 8;
        if (!memcmp(type, "IHDR", 4)) {
            if (len != 13 || have_ihdr) return false;
            W = be32(o + 8); H = be32(o + 12);
            depth = body[8]; ctype = body[9]; interlace = body[12];
            if (body[10] != 0 || body[11] != 0 || interlace > 1) return false;
            have_ihdr = true;
        } else if (!memcmp(type, "PLTE", 4)) {
            plte.assign(body, body + len);
        } else if (!memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), body, body + len);
        } else if (!memcmp(type, "IEND", 4)) {
            done = true;
        }
        o += 12 + (size_t)len;
    }
    if (!have_ihdr || W == 0 || H == 0 || W > 32768 || H > 32768 || idat.empty()) return false;
    if ((size_t)W * (size_t)H > kMaxTexturePixels) return false;
    int channels;
    switch (ctype) {
    case 0: channels = 1; break;        // grey
    case 2: channels = 3; break;        // RGB
    case 3: channels = 1; break;        // palette index
    case 4: channels = 2; break;        // grey + alpha
    case 6: channels = 4; break;        // RGB + alpha
    default: return false;
    }
    const bool depth_ok = (ctype == 0 && (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) ||
                          (ctype == 3 && (depth == 1 || depth == 2 || depth == 4 || depth == 8)) ||
                          ((ctype == 2 || ctype == 4 || ctype == 6) && (depth == 8 || depth == 16));
    if (!depth_ok || (ctype == 3 && (plte.empty() || plte.size() % 3 != 0))) return false;
    const int bits_pp = channels * depth;                   // bits per pixel
    const size_t bpp = (size_t)(bits_pp + 7) / 8;           // filter distance in bytes (>= 1)
    auto row_bytes = [&](uint32_t w) { return ((size_t)w * bits_pp + 7) / 8; };
    // the passes: (x0, y0, dx, dy); one pass when not interlaced
    static const int adam7[7][4] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
    static const int whole[1][4] = {{0, 0, 1, 1}};
    const int (*pass)[4] = interlace ? adam7 : whole;
    const int npass = interlace ? 7 : 1;
    size_t raw_size = 0;
    for (int k = 0; k < npass; ++k) {
        const uint32_t pw = (W - pass[k][0] + pass[k][2] - 1) / pass[k][2], ph = (H - pass[k][1] + pass[k][3] - 1) / pass[k][3];
        if (W <= (uint32_t)pass[k][0] || H <= (uint32_t)pass[k][1] || pw == 0 || ph == 0) continue;
        raw_size += (row_bytes(pw) + 1) * ph;
    }
    // deflate cannot expand its input by more than a factor of about 1032: a header that asks for
    // more than the IDAT bytes could hold is refused before anything is allocated for it
    if (raw_size > idat.size() * 1032 + 4096) return false;
    std::vector<uint8_t> raw(raw_size);
    uLongf got = (uLongf)raw_size;
    if (uncompress(raw.data(), &got, idat.data(), (uLong)idat.size()) != Z_OK || got != raw_size) return false;
    im.w = (int)W; im.h = (int)H;
    im.rgb.assign((size_t)W * H * 3, 0);
    size_t at = 0;
    std::vector<uint8_t> prev, cur;
    for (int k = 0; k < npass; ++k) {
        if (W <= (uint32_t)pass[k][0] || H <= (uint32_t)pass[k][1]) continue;
        const uint32_t pw = (W - pass[k][0] + pass[k][2] - 1) / pass[k][2], ph = (H - pass[k][1] + pass[k][3] - 1) / pass[k][3];
        if (pw == 0 || ph == 0) continue;
        const size_t rb = row_bytes(pw);
        prev.assign(rb, 0);
        cur.resize(rb);
        for (uint32_t y = 0; y < ph; ++y) {
            const int filter = raw[at++];
            const uint8_t* src = &raw[at];
            at += rb;
            for (size_t i = 0; i < rb; ++i) {
                const int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
                int pred;
                switch (filter) {
                case 0: pred = 0; break;
                case 1: pred = a; break;
                case 2: pred = b; break;
                case 3: pred = (a + b) / 2; break;
                case 4: {
                    const int pa = abs(b - c), pb = abs(a - c), pc = abs(a + b - 2 * c);
                    pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
                    break;
                }
                default: return false;
                }
                cur[i] = (uint8_t)(src[i] + pred);
            }
            // samples of the row -> 8-bit RGB
            auto sample = [&](uint32_t x, int ch) -> int {          // raw sample value
                const size_t bit = ((size_t)x * channels + ch) * depth;
                if (depth == 8) return cur[bit / 8];
                if (depth == 16) return cur[bit / 8];               // high byte (SDL_image strips 16 -> 8)
                return (cur[bit / 8] >> (8 - depth - (bit % 8))) & ((1 << depth) - 1);
            };
            const uint32_t oy = pass[k][1] + y * pass[k][3];
            for (uint32_t x = 0; x < pw; ++x) {
                uint8_t* o = &im.rgb[((size_t)oy * W + pass[k][0] + (size_t)x * pass[k][2]) * 3];
                if (ctype == 3) {
                    const size_t e = (size_t)sample(x, 0) * 3;
                    if (e + 2 < plte.size()) { o[0] = plte[e]; o[1] = plte[e + 1]; o[2] = plte[e + 2]; }
                } else if (ctype == 0 || ctype == 4) {
                    int g = sample(x, 0);
                    if (depth < 8) g = g * 255 / ((1 << depth) - 1);        // full-range scaling of 1/2/4-bit grey
                    o[0] = o[1] = o[2] = (uint8_t)g;
                } else {
                    o[0] = (uint8_t)sample(x, 0); o[1] = (uint8_t)sample(x, 1); o[2] = (uint8_t)sample(x, 2);
                }
            }
            prev.swap(cur);
            cur.resize(rb);
        }
    }
    return true;
}

// Baseline / extended-sequential JPEG (ITU T.81), 8 bits per sample, one or three components in
// one interleaved scan, chroma subsampled 1x1, 2x1 or 2x2, restart intervals.  SDL_image decodes
// JPEG through libjpeg with its defaults, so the same three stages are used here so that the
// texels agree with what the reference's texture fetch would see: the accurate integer inverse
// DCT (13-bit constants, two passes), "fancy" triangle-filter chroma upsampling, and the 16-bit
// fixed-point YCbCr -> RGB tables.  Progressive, arithmetic-coded, 12-bit, CMYK and other
// subsamplings are refused (the map then behaves like a failed IMG_Load).
struct JpegDecoder {
    const uint8_t* d;
    size_t n, pos = 0;
    struct Huff {
        bool present = false;
        uint8_t vals[256];
        int mincode[17], maxcode[18], valptr[17];
    } dc[4], ac[4];
    uint16_t qt[4][64];
    bool qt_present[4] = {false, false, false, false};
    struct Comp {
        int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0, pred = 0;
        int stride = 0, rows = 0;       // padded plane
        int dw = 0, dh = 0;             // real (downsampled) size
        std::vector<uint8_t> plane;
    } comp[3];
    int W = 0, H = 0, ncomp = 0, hmax = 1, vmax = 1, restart = 0;
    bool adobe_rgb = false;
    uint32_t bitbuf = 0;
    int bitcnt = 0;
    bool bad = false;

    int u8() { return pos < n ? d[pos++] : (bad = true, 0); }
    int u16() { const int a = u8(); return a << 8 | u8(); }

    void fill()
    {
        while (bitcnt <= 24) {
            int b = 0;
            if (pos >= n) bad = true;           // the data ends inside the scan (no EOI marker): refuse
            if (pos < n) {
                b = d[pos];
                if (b == 0xFF) {
                    const int m = pos + 1 < n ? d[pos + 1] : 0xD9;
                    if (m == 0) pos += 2;           // stuffed zero
                    else b = 0;                     // a marker: feed zeros, leave it in place
                } else {
                    ++pos;
                }
            }
            bitbuf |= (uint32_t)b << (24 - bitcnt);
            bitcnt += 8;
        }
    }
    int bits(int k)
    {
        if (k == 0) return 0;
        fill();
        const int v = (int)(bitbuf >> (32 - k));
        bitbuf <<= k;
        bitcnt -= k;
        return v;
    }
    int decode(const Huff& h)
    {
        int code = 0;
        for (int len = 1; len <= 16; ++len) {
            code = code << 1 | bits(1);
            if (h.maxcode[len] >= 0 && code <= h.maxcode[len] && code >= h.mincode[len])
                return h.vals[h.valptr[len] + code - h.mincode[len]];
        }
        bad = true;
        return 0;
    }
    static int extend(int v, int s) { return s && v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

    bool build(Huff& h, const uint8_t counts[17], const uint8_t* vals, int nvals)
    {
        int code = 0, k = 0;
        for (int len = 1; len <= 16; ++len) {
            h.valptr[len] = k;
            h.mincode[len] = code;
            code += counts[len];
            k += counts[len];
            h.maxcode[len] = counts[len] ? code - 1 : -1;
            code <<= 1;
        }
        if (k != nvals || k > 256) return false;
        memcpy(h.vals, vals, (size_t)nvals);
        h.present = true;
        return true;
    }

    // jidctint.c (accurate integer IDCT): CONST_BITS 13, PASS1_BITS 2
    static void idct(const int in[64], uint8_t* out, int stride)
    {
        constexpr long F0298 = 2446, F0390 = 3196, F0541 = 4433, F0765 = 6270, F0899 = 7373, F1175 = 9633,
                       F1501 = 12299, F1847 = 15137, F1961 = 16069, F2053 = 16819, F2562 = 20995, F3072 = 25172;
        long ws[64];
        auto descale = [](long x, int nbits) { return (x + (1L << (nbits - 1))) >> nbits; };
        auto pass = [&](const long* src, int sstep, long* dst, int dstep, int shift) {
            long z2 = src[2 * sstep], z3 = src[6 * sstep];
            long z1 = (z2 + z3) * F0541;
            long tmp2 = z1 + z3 * (-F1847), tmp3 = z1 + z2 * F0765;
            z2 = src[0]; z3 = src[4 * sstep];
            long tmp0 = (z2 + z3) << 13, tmp1 = (z2 - z3) << 13;
            const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
            tmp0 = src[7 * sstep]; tmp1 = src[5 * sstep]; tmp2 = src[3 * sstep]; tmp3 = src[1 * sstep];
            z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
            long z4 = tmp1 + tmp3;
            const long z5 = (z3 + z4) * F1175;
            tmp0 *= F0298; tmp1 *= F2053; tmp2 *= F3072; tmp3 *= F1501;
            z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
            z3 += z5; z4 += z5;
            tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
            dst[0 * dstep] = descale(tmp10 + tmp3, shift); dst[7 * dstep] = descale(tmp10 - tmp3, shift);
            dst[1 * dstep] = descale(tmp11 + tmp2, shift); dst[6 * dstep] = descale(tmp11 - tmp2, shift);
            dst[2 * dstep] = descale(tmp12 + tmp1, shift); dst[5 * dstep] = descale(tmp12 - tmp1, shift);
            dst[3 * dstep] = descale(tmp13 + tmp0, shift); dst[4 * dstep] = descale(tmp13 - tmp0, shift);
        };
        long col[64];
        for (int i = 0; i < 64; ++i) col[i] = in[i];
        for (int c = 0; c < 8; ++c) pass(col + c, 8, ws + c, 8, 13 - 2);            // columns
        for (int r = 0; r < 8; ++r) {
            long row[8];
            pass(ws + 8 * r, 1, row, 1, 13 + 2 + 3);                                // rows
            for (int c = 0; c < 8; ++c) {
                long v = row[c] + 128;
                out[(size_t)r * stride + c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
            }
        }
    }

    bool decode_scan()
    {
        static const uint8_t zigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48,
                                           41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22,
                                           15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55,
                                           62, 63};
        const int mcux = (W + 8 * hmax - 1) / (8 * hmax), mcuy = (H + 8 * vmax - 1) / (8 * vmax);
        // every block of every MCU takes at least two bits of entropy-coded data (a DC and an
        // end-of-block code): a frame header that asks for more MCUs than the bytes left in the
        // file could hold is refused before its planes are allocated
        {
            size_t blocks_per_mcu = 0;
            for (int c = 0; c < ncomp; ++c) blocks_per_mcu += (size_t)comp[c].h * comp[c].v;
            const size_t left = pos < n ? n - pos : 0;
            if ((size_t)W * (size_t)H > kMaxTexturePixels || (size_t)mcux * mcuy * blocks_per_mcu > left * 4 + 64) return false;
        }
        for (int c = 0; c < ncomp; ++c) {
            Comp& k = comp[c];
            k.stride = mcux * k.h * 8;
            k.rows = mcuy * k.v * 8;
            k.dw = (W * k.h + hmax - 1) / hmax;
            k.dh = (H * k.v + vmax - 1) / vmax;
            k.plane.assign((size_t)k.stride * k.rows, 0);
            k.pred = 0;
            if (!dc[k.td].present || !ac[k.ta].present || !qt_present[k.tq]) return false;
        }
        int until_restart = restart, next_rst = 0;
        for (int my = 0; my < mcuy; ++my)
            for (int mx = 0; mx < mcux; ++mx) {
                if (restart && until_restart == 0) {
                    // byte-align, expect RSTn
                    bitbuf = 0; bitcnt = 0;
                    while (pos + 1 < n && !(d[pos] == 0xFF && d[pos + 1] >= 0xD0 && d[pos + 1] <= 0xD7)) ++pos;
                    if (pos + 1 >= n || d[pos + 1] != 0xD0 + next_rst) return false;
                    pos += 2;
                    next_rst = (next_rst + 1) & 7;
                    until_restart = restart;
                    for (int c = 0; c < ncomp; ++c) comp[c].pred = 0;
                }
                for (int c = 0; c < ncomp; ++c) {
                    Comp& k = comp[c];
                    for (int by = 0; by < k.v; ++by)
                        for (int bx = 0; bx < k.h; ++bx) {
                            int coef[64] = {0};
                            const int s = decode(dc[k.td]);
                            if (s > 11) return false;
                            k.pred += extend(bits(s), s);
                            // (a corrupt stream can drive the prediction anywhere: keep the products in range)
                            if (k.pred > 32767 || k.pred < -32768) { bad = true; k.pred = 0; }
                            coef[0] = (int)((long)k.pred * qt[k.tq][0]);
                            for (int i = 1; i < 64;) {
                                const int rs = decode(ac[k.ta]);
                                const int r = rs >> 4, sz = rs & 15;
                                if (sz == 0) {
                                    if (r != 15) break;         // end of block
                                    i += 16;
                                    continue;
                                }
                                i += r;
                                if (i > 63) return false;
                                coef[zigzag[i]] = (int)((long)extend(bits(sz), sz) * qt[k.tq][i]);
                                ++i;
                            }
                            if (bad) return false;
                            idct(coef, &k.plane[((size_t)(my * k.v + by) * 8) * k.stride + (size_t)(mx * k.h + bx) * 8],
                                 k.stride);
                        }
                }
                if (restart) --until_restart;
            }
        return !bad;
    }

    bool parse()
    {
        if (n < 4 || d[0] != 0xFF || d[1] != 0xD8) return false;
        pos = 2;
        bool have_sof = false;
        while (pos + 4 <= n) {
            if (u8() != 0xFF) return false;
            int m = u8();
            while (m == 0xFF) m = u8();
            if (m == 0xD9) return false;                    // EOI before any scan
            const int len = u16();
            if (len < 2 || pos + (size_t)(len - 2) > n) return false;
            const size_t end = pos + (size_t)(len - 2);
            if (m == 0xDB) {                                // DQT
                while (pos < end) {
                    const int pq = u8(), t = pq & 15;
                    if (t > 3) return false;
                    for (int i = 0; i < 64; ++i) qt[t][i] = (uint16_t)((pq >> 4) ? u16() : u8());     // zigzag order
                    qt_present[t] = true;
                }
            } else if (m == 0xC0 || m == 0xC1) {            // SOF0 / SOF1 (Huffman, sequential)
                if (u8() != 8) return false;                // sample precision
                H = u16(); W = u16(); ncomp = u8();
                if (W <= 0 || H <= 0 || W > 32768 || H > 32768 || (ncomp != 1 && ncomp != 3)) return false;
                for (int c = 0; c < ncomp; ++c) {
                    comp[c].id = u8();
                    const int hv = u8();
                    comp[c].h = hv >> 4; comp[c].v = hv & 15; comp[c].tq = u8();
                    if (comp[c].tq > 3) return false;
                }
                have_sof = true;
            } else if (m >= 0xC2 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
                return false;                               // progressive, lossless, arithmetic
            } else if (m == 0xC4) {                         // DHT
                while (pos < end) {
                    const int tc = u8(), cls = tc >> 4, t = tc & 15;
                    if (cls > 1 || t > 3) return false;
                    uint8_t counts[17] = {0};
                    int total = 0;
                    for (int i = 1; i <= 16; ++i) { counts[i] = (uint8_t)u8(); total += counts[i]; }
                    if (total > 256 || pos + (size_t)total > end) return false;
                    if (!build(cls ? ac[t] : dc[t], counts, d + pos, total)) return false;
                    pos += (size_t)total;
                }
            } else if (m == 0xDD) {                         // DRI
                restart = u16();
            } else if (m == 0xEE && len >= 14 && !memcmp(d + pos, "Adobe", 5)) {
                adobe_rgb = d[pos + 11] == 0;               // transform 0 with three components: RGB
            } else if (m == 0xDA) {                         // SOS
                if (!have_sof) return false;
                const int ns = u8();
                if (ns != ncomp) return false;              // one interleaved scan only
                for (int i = 0; i < ns; ++i) {
                    const int id = u8(), tt = u8();
                    if (id != comp[i].id) return false;
                    comp[i].td = tt >> 4; comp[i].ta = tt & 15;
                    if (comp[i].td > 3 || comp[i].ta > 3) return false;
                }
                pos = end;
                if (ncomp == 1) { comp[0].h = comp[0].v = 1; }      // a single component is never interleaved
                hmax = vmax = 1;
                for (int c = 0; c < ncomp; ++c) {
                    if (comp[c].h < 1 || comp[c].v < 1) return false;
                    hmax = comp[c].h > hmax ? comp[c].h : hmax;
                    vmax = comp[c].v > vmax ? comp[c].v : vmax;
                }
                if (ncomp == 3) {
                    const bool chroma_ok = comp[1].h == 1 && comp[1].v == 1 && comp[2].h == 1 && comp[2].v == 1;
                    const bool luma_ok = (comp[0].h == 1 || comp[0].h == 2) && (comp[0].v == 1 || comp[0].v == 2) &&
                                         !(comp[0].h == 1 && comp[0].v == 2);
                    if (!chroma_ok || !luma_ok) return false;
                }
                return decode_scan();
            }
            pos = end;
            if (bad) return false;
        }
        return false;
    }

    // jdsample.c: chroma plane to full resolution, "fancy" (triangle filter) upsampling
    std::vector<uint8_t> upsample(const Comp& k) const
    {
        std::vector<uint8_t> out((size_t)W * H);
        const int hs = hmax / k.h, vs = vmax / k.v;
        for (int y = 0; y < H; ++y) {
            uint8_t* o = &out[(size_t)y * W];
            if (hs == 1 && vs == 1) {
                memcpy(o, &k.plane[(size_t)y * k.stride], (size_t)W);
            } else if (k.dw <= 2) {                         // too narrow for the filter: plain replication
                const uint8_t* in = &k.plane[(size_t)(y / vs) * k.stride];
                for (int x = 0; x < W; ++x) o[x] = in[x >> 1];
            } else if (hs == 2 && vs == 1) {                // h2v1_fancy_upsample
                const uint8_t* in = &k.plane[(size_t)y * k.stride];
                for (int x = 0; x < W; ++x) {
                    const int c = x >> 1;
                    int v;
                    if ((x & 1) == 0) v = c == 0 ? in[0] : (in[c] * 3 + in[c - 1] + 1) >> 2;
                    else v = c == k.dw - 1 ? in[c] : (in[c] * 3 + in[c + 1] + 2) >> 2;
                    o[x] = (uint8_t)v;
                }
            } else {                                        // h2v2_fancy_upsample
                const int r0 = y >> 1;
                int r1 = (y & 1) ? r0 + 1 : r0 - 1;         // the further row: below for odd rows, above for even
                r1 = r1 < 0 ? 0 : (r1 > k.dh - 1 ? k.dh - 1 : r1);
                const uint8_t* in0 = &k.plane[(size_t)r0 * k.stride];
                const uint8_t* in1 = &k.plane[(size_t)r1 * k.stride];
                auto colsum = [&](int c) { return in0[c] * 3 + in1[c]; };
                for (int x = 0; x < W; ++x) {
                    const int c = x >> 1, cur = colsum(c);
                    int v;
                    if ((x & 1) == 0) v = c == 0 ? (cur * 4 + 8) >> 4 : (cur * 3 + colsum(c - 1) + 8) >> 4;
                    else v = c == k.dw - 1 ? (cur * 4 + 7) >> 4 : (cur * 3 + colsum(c + 1) + 7) >> 4;
                    o[x] = (uint8_t)v;
                }
            }
        }
        return out;
    }
};

bool load_jpeg(const std::string& data, Image& im)
{
    JpegDecoder j{(const uint8_t*)data.data(), data.size()};
    if (!j.parse()) return false;
    im.w = j.W; im.h = j.H;
    im.rgb.resize((size_t)j.W * j.H * 3);
    if (j.ncomp == 1) {
        for (int y = 0; y < j.H; ++y)
            for (int x = 0; x < j.W; ++x) {
                const uint8_t g = j.comp[0].plane[(size_t)y * j.comp[0].stride + x];
                uint8_t* o = &im.rgb[((size_t)y * j.W + x) * 3];
                o[0] = o[1] = o[2] = g;
            }
        return true;
    }
    const std::vector<uint8_t> Y = j.upsample(j.comp[0]), Cb = j.upsample(j.comp[1]), Cr = j.upsample(j.comp[2]);
    // jdcolor.c: 16-bit fixed-point tables, FIX(x) = (int)(x * 65536 + 0.5)
    auto clamp = [](int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); };
    for (size_t i = 0; i < (size_t)j.W * j.H; ++i) {
        uint8_t* o = &im.rgb[i * 3];
        if (j.adobe_rgb) { o[0] = Y[i]; o[1] = Cb[i]; o[2] = Cr[i]; continue; }
        const int y = Y[i], cb = Cb[i] - 128, cr = Cr[i] - 128;
        const int r = y + (int)((91881L * cr + 32768) >> 16);
        const int g = y + (int)((-22554L * cb - 46802L * cr + 32768) >> 16);
        const int b = y + (int)((116130L * cb + 32768) >> 16);
        o[0] = clamp(r); o[1] = clamp(g); o[2] = clamp(b);
    }
    return true;
}

// ---------------------------------------------------------------- parsed object
struct MaterialRec {
    std::string name;
    double Ka[3] = {1.0, 1.0, 1.0}, Kd[3] = {1.0, 1.0, 1.0}, Ks[3] = {1.0, 1.0, 1.0};   // :296-298
    int map_Ka = -1, map_Kd = -1, map_Ks = -1;
};

}  // namespace

struct rt_obj {
    int num_v = 0, num_vn = 0, num_vt = 0, num_tris = 0, num_mats = 0;
    bool pinned = false;
    // single allocation carved into the arrays of rt_obj_view
    void* block = nullptr;
    rt_obj_view view{};
    std::vector<Image> textures;
    std::vector<std::string> texture_names;
};

namespace {

struct Builder {
    std::vector<double> x{0.0}, y{0.0}, z{0.0}, xn{0.0}, yn{0.0}, zn{0.0}, u{0.0}, v{0.0};
    std::vector<int> idx[9];    // A B C  An Bn Cn  At Bt Ct
    std::vector<int> mtl;
    std::vector<MaterialRec> mats;
    std::vector<Image> textures;
    std::vector<std::string> texture_names;
    int current_mat = -1;
    std::string dir;
    std::string err;

    int num_v() const { return (int)x.size() - 1; }
    int num_vn() const { return (int)xn.size() - 1; }
    int num_vt() const { return (int)u.size() - 1; }

    void push_tri(const int v[3], const int vt[3], const int vn[3])
    {
        for (int k = 0; k < 3; ++k) {
            idx[0 + k].push_back(v[k]);
            idx[3 + k].push_back(vn[k]);
            idx[6 + k].push_back(vt[k]);
        }
        mtl.push_back(current_mat);                          // set_mtl, :84-90
    }

    // find_and_add_normal, :51-82 with M3d_x_product (M3d_matrix_tools.c:220-222)
    int add_face_normal(int A, int B, int C)
    {
        const double AB[3] = {x[B] - x[A], y[B] - y[A], z[B] - z[A]};
        const double AC[3] = {x[C] - x[A], y[C] - y[A], z[C] - z[A]};
        xn.push_back(AB[1] * AC[2] - AC[1] * AB[2]);
        yn.push_back(AC[0] * AB[2] - AB[0] * AC[2]);
        zn.push_back(AB[0] * AC[1] - AC[0] * AB[1]);
        return num_vn();
    }

    // get_face_format, :27-48
    static int face_format(Cursor c)
    {
        int a, n = 0;
        if (!c.integer(a)) return 0;
        n = 1;
        if (c.literal('/')) {
            if (c.integer(a)) {
                n = 2;
                if (c.literal('/') && c.integer(a)) n = 3;
            }
        }
        if (n == 1) return c.literal('/') ? 4 : 1;
        return n;
    }

    // one vertex reference in the given format; false = not there (cursor untouched)
    static bool face_vertex(Cursor& c, int format, int& v, int& vt, int& vn)
    {
        Cursor save = c;
        vt = vn = 0;
        bool ok = c.integer(v);
        if (ok && format == 2) ok = c.literal('/') && c.integer(vt);
        if (ok && format == 3) ok = c.literal('/') && c.integer(vt) && c.literal('/') && c.integer(vn);
        if (ok && format == 4) ok = c.literal('/') && c.literal('/') && c.integer(vn);
        if (!ok) c = save;
        return ok;
    }

    bool check_index(int i, int n, const char* what)
    {
        if (i >= 1 && i <= n) return true;
        char buf[160];
        snprintf(buf, sizeof buf, "%s index %d outside [1,%d] (relative indices are not supported)",
                 what, i, n);
        err = buf;
        return false;
    }

    // scan_face, :233-237
    bool face(Cursor& c)
    {
        const int format = face_format(c);
        if (format == 0) { err = "face without vertices"; return false; }
        int v[5], vt[5], vn[5], n = 0;
        while (n < 5 && face_vertex(c, format, v[n], vt[n], vn[n])) ++n;
        if (n < 3) { err = "face with fewer than 3 vertices"; return false; }
        if (n > 4) { err = "The object contains a face with more than 4 vertices"; return false; }
        for (int k = 0; k < n; ++k) {
            if (!check_index(v[k], num_v(), "vertex")) return false;
            if ((format == 2 || format == 3) && !check_index(vt[k], num_vt(), "texture")) return false;
            if ((format == 3 || format == 4) && !check_index(vn[k], num_vn(), "normal")) return false;
        }
        if (format == 1 || format == 2) {
            const int fn = add_face_normal(v[0], v[1], v[2]);
            for (int k = 0; k < n; ++k) vn[k] = fn;
        }
        const int t0v[3] = {v[0], v[1], v[2]}, t0t[3] = {vt[0], vt[1], vt[2]}, t0n[3] = {vn[0], vn[1], vn[2]};
        push_tri(t0v, t0t, t0n);
        if (n == 4) {                                        // :167-231: (a, c, d)
            const int t1v[3] = {v[0], v[2], v[3]}, t1t[3] = {vt[0], vt[2], vt[3]}, t1n[3] = {vn[0], vn[2], vn[3]};
            push_tri(t1v, t1t, t1n);
        }
        return true;
    }

    int load_texture(const std::string& name)
    {
        std::string data;
        Image im;
        bool decoded = false;
        try {
            decoded = read_file(dir + name, data) &&
                      (load_png(data, im) || load_jpeg(data, im) || load_ppm(data, im) || load_bmp(data, im));
        } catch (const std::exception&) {       // (out of memory inside a decoder: the map stays unset)
            decoded = false;
        }
        if (!decoded) {
            fprintf(stderr, "rt_obj: cannot decode image map %s%s (kept as a constant colour)\n",
                    dir.c_str(), name.c_str());
            return -1;                                       // IMG_Load failed, :360-363
        }
        textures.push_back(std::move(im));
        texture_names.push_back(name);
        return (int)textures.size() - 1;
    }

    // read_mtl_file, :383-453
    bool read_mtl(const std::string& path)
    {
        std::string data;
        if (!read_file(path, data)) { err = "can't open file, " + path; return false; }
        Cursor c{data.data(), data.data() + data.size()};
        std::string key, name;
        auto three = [&](double d[3]) { c.real(d[0]) && c.real(d[1]) && c.real(d[2]); };
        double dump[3];
        while (c.token(key)) {
            MaterialRec* cur = mats.empty() ? nullptr : &mats.back();
            if (key == "newmtl") {
                MaterialRec m;
                c.token(m.name);
                mats.push_back(m);
            } else if (key == "Ka" && cur) three(cur->Ka);
            else if (key == "Kd" && cur) three(cur->Kd);
            else if (key == "Ks" && cur) three(cur->Ks);
            else if (key == "Ke" || key == "Tf") three(dump);
            else if (key == "map_Ka" && cur) { c.token(name); cur->map_Ka = load_texture(name); }
            else if (key == "map_Kd" && cur) { c.token(name); cur->map_Kd = load_texture(name); }
            else if (key == "map_Ks" && cur) { c.token(name); cur->map_Ks = load_texture(name); }
            else if (key == "Tr" || key == "Ns" || key == "Ni" || key == "d") c.real(dump[0]);
            else if (key == "illum") { int i; c.integer(i); }
            else c.skip_line();                              // map_bump, '#', anything else
        }
        return true;
    }

    bool parse(const std::string& path)
    {
        std::string data;
        if (!read_file(path, data)) { err = "can't open file, " + path; return false; }
        const size_t slash = path.find_last_of("/\\");
        dir = slash == std::string::npos ? "" : path.substr(0, slash + 1);
        Cursor c{data.data(), data.data() + data.size()};
        std::string key, name;
        while (c.token(key)) {
            if (key == "v") {                                // :666-670 (optional w is swallowed)
                double p[4] = {0, 0, 0, 0};
                c.real(p[0]) && c.real(p[1]) && c.real(p[2]) && c.real(p[3]);
                x.push_back(p[0]); y.push_back(p[1]); z.push_back(p[2]);
            } else if (key == "vt") {                        // :671-675
                double p[2] = {0, 0};
                c.real(p[0]) && c.real(p[1]);
                u.push_back(p[0]); v.push_back(p[1]);
            } else if (key == "vn") {                        // :676-680
                double p[3] = {0, 0, 0};
                c.real(p[0]) && c.real(p[1]) && c.real(p[2]);
                xn.push_back(p[0]); yn.push_back(p[1]); zn.push_back(p[2]);
            } else if (key == "g" || key == "s") {           // :684-691
                c.token(name);
            } else if (key == "f") {                         // :692-697
                if (!face(c)) return false;
            } else if (key == "mtllib") {                    // :698-707
                c.token(name);
                if (!read_mtl(dir + name)) return false;
            } else if (key == "usemtl") {                    // :708-716
                c.token(name);
                for (size_t i = 0; i < mats.size(); ++i)
                    if (mats[i].name == name) { current_mat = (int)i; break; }
            } else if (key[0] == '#') {                      // :717-719
                c.skip_line();
            }
            // any other keyword ("vp", "o", ...): ignored, its arguments are read as further
            // unknown keywords, as in the reference's second pass
        }
        if (num_v() < 1) { err = "object has no vertices"; return false; }
        return true;
    }

    // center_and_scale_object, :239-289; m = S*T applied like mat_mult.cu:20-22
    void center_and_scale()
    {
        const int n = num_v();
        double xc = 0, yc = 0, zc = 0;
        double lo[3] = {x[1], y[1], z[1]}, hi[3] = {x[1], y[1], z[1]};
        for (int i = 1; i <= n; ++i) {
            xc += x[i]; yc += y[i]; zc += z[i];
            if (x[i] < lo[0]) lo[0] = x[i];
            if (y[i] < lo[1]) lo[1] = y[i];
            if (z[i] < lo[2]) lo[2] = z[i];
            if (x[i] > hi[0]) hi[0] = x[i];
            if (y[i] > hi[1]) hi[1] = y[i];
            if (z[i] > hi[2]) hi[2] = z[i];
        }
        xc /= n; yc /= n; zc /= n;
        const double xd = hi[0] - lo[0], yd = hi[1] - lo[1], zd = hi[2] - lo[2];
        double delta;
        if ((xd > yd) && (xd > zd)) delta = xd;
        else if (yd > zd) delta = yd;
        else delta = zd;
        const double sf = 15 / delta;
        const double t[4][4] = {{1, 0, 0, -xc}, {0, 1, 0, -yc}, {0, 0, 1, -zc}, {0, 0, 0, 1}};
        const double s[4][4] = {{sf, 0, 0, 0}, {0, sf, 0, 0}, {0, 0, sf, 0}, {0, 0, 0, 1}};
        double m[4][4];
        rt_mat_mult(m, s, t);                                // M3d_mat_mult(m, s, t), :286
        for (int i = 0; i <= n; ++i) {
            const double px = x[i], py = y[i], pz = z[i];
            x[i] = m[0][0] * px + m[0][1] * py + m[0][2] * pz + m[0][3];
            y[i] = m[1][0] * px + m[1][1] * py + m[1][2] * pz + m[1][3];
            z[i] = m[2][0] * px + m[2][1] * py + m[2][2] * pz + m[2][3];
        }
    }
};

template <class T>
T* carve(char*& cur, size_t count)
{
    T* p = reinterpret_cast<T*>(cur);
    cur += ((count * sizeof(T) + 63) / 64) * 64;
    return p;
}

}  // namespace

static int obj_load_impl(const char* path, rt_obj** out);

extern "C" int rt_obj_load(const char* path, rt_obj** out)
{
    if (!path || !out) {
        rt_set_error("rt_obj_load: null pointer");
        return RT_E_ARG;
    }
    // nothing may leave an extern "C" function by exception
    try {
        return obj_load_impl(path, out);
    } catch (const std::bad_alloc&) {
        rt_set_error("rt_obj_load: out of memory");
        return RT_E_IO;
    } catch (const std::exception& e) {
        rt_set_error("rt_obj_load: %s", e.what());
        return RT_E_IO;
    }
}

static int obj_load_impl(const char* path, rt_obj** out)
{
    Builder b;
    if (!b.parse(path)) {
        rt_set_error("rt_obj_load: %s", b.err.c_str());
        return RT_E_IO;
    }
    b.center_and_scale();

    rt_obj* o = new rt_obj();
    o->num_v = b.num_v(); o->num_vn = b.num_vn(); o->num_vt = b.num_vt();
    o->num_tris = (int)b.mtl.size(); o->num_mats = (int)b.mats.size();
    const size_t nv = o->num_v + 1, nn = o->num_vn + 1, nt = o->num_vt + 1, ntri = o->num_tris,
                 nm = o->num_mats;
    const size_t bytes = 64 * 32 + sizeof(double) * (3 * nv + 3 * nn + 2 * nt + 9 * nm) +
                         sizeof(int) * (10 * ntri + 3 * nm);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0 &&
        hipHostMalloc(&o->block, bytes, hipHostMallocDefault) == hipSuccess) {
        o->pinned = true;
    } else {
        (void)hipGetLastError();
        o->block = aligned_alloc(64, (bytes + 63) / 64 * 64);
        o->pinned = false;
    }
    if (!o->block) {
        delete o;
        rt_set_error("rt_obj_load: out of memory");
        return RT_E_IO;
    }
    char* cur = static_cast<char*>(o->block);
    rt_obj_view& v = o->view;
    auto put_d = [&](const std::vector<double>& src) {
        double* p = carve<double>(cur, src.size());
        memcpy(p, src.data(), src.size() * sizeof(double));
        return p;
    };
    auto put_i = [&](const std::vector<int>& src) {
        int* p = carve<int>(cur, src.size());
        if (!src.empty()) memcpy(p, src.data(), src.size() * sizeof(int));
        return p;
    };
    v.num_v = o->num_v; v.num_vn = o->num_vn; v.num_vt = o->num_vt;
    v.num_tris = o->num_tris; v.num_mats = o->num_mats;
    v.x = put_d(b.x); v.y = put_d(b.y); v.z = put_d(b.z);
    v.xn = put_d(b.xn); v.yn = put_d(b.yn); v.zn = put_d(b.zn);
    v.u = put_d(b.u); v.v = put_d(b.v);
    v.A = put_i(b.idx[0]); v.B = put_i(b.idx[1]); v.C = put_i(b.idx[2]);
    v.An = put_i(b.idx[3]); v.Bn = put_i(b.idx[4]); v.Cn = put_i(b.idx[5]);
    v.At = put_i(b.idx[6]); v.Bt = put_i(b.idx[7]); v.Ct = put_i(b.idx[8]);
    v.mtl = put_i(b.mtl);
    std::vector<double> Ka(3 * nm), Kd(3 * nm), Ks(3 * nm);
    std::vector<int> mKa(nm), mKd(nm), mKs(nm);
    for (size_t i = 0; i < nm; ++i) {
        for (int k = 0; k < 3; ++k) {
            Ka[3 * i + k] = b.mats[i].Ka[k]; Kd[3 * i + k] = b.mats[i].Kd[k]; Ks[3 * i + k] = b.mats[i].Ks[k];
        }
        mKa[i] = b.mats[i].map_Ka; mKd[i] = b.mats[i].map_Kd; mKs[i] = b.mats[i].map_Ks;
    }
    v.Ka = put_d(Ka); v.Kd = put_d(Kd); v.Ks = put_d(Ks);
    v.map_Ka = put_i(mKa); v.map_Kd = put_i(mKd); v.map_Ks = put_i(mKs);
    v.pinned = o->pinned ? 1 : 0;
    o->textures = std::move(b.textures);
    o->texture_names = std::move(b.texture_names);
    *out = o;
    return RT_OK;
}

extern "C" int rt_obj_free(rt_obj* o)
{
    if (!o) return RT_OK;
    if (o->block) {
        if (o->pinned) (void)hipHostFree(o->block);
        else free(o->block);
    }
    delete o;
    return RT_OK;
}

extern "C" int rt_obj_get(const rt_obj* o, rt_obj_view* out)
{
    if (!o || !out) {
        rt_set_error("rt_obj_get: null pointer");
        return RT_E_ARG;
    }
    *out = o->view;
    return RT_OK;
}

extern "C" int rt_obj_num_textures(const rt_obj* o) { return o ? (int)o->textures.size() : 0; }

extern "C" int rt_obj_texture(const rt_obj* o, int index, int* w, int* h, const uint8_t** rgb)
{
    if (!o || index < 0 || index >= (int)o->textures.size()) {
        rt_set_error("rt_obj_texture: no texture %d", index);
        return RT_E_ARG;
    }
    const Image& im = o->textures[(size_t)index];
    if (w) *w = im.w;
    if (h) *h = im.h;
    if (rgb) *rgb = im.rgb.data();
    return RT_OK;
}

extern "C" int rt_scene_set_obj(rt_scene* sc, const rt_obj* o)
{
    if (!sc || !o) {
        rt_set_error("rt_scene_set_obj: null pointer");
        return RT_E_ARG;
    }
    const rt_obj_view& v = o->view;
    int rc = rt_scene_set_vertices(sc, v.num_v, v.x, v.y, v.z);
    if (rc) return rc;
    rc = rt_scene_set_normals(sc, v.num_vn, v.xn, v.yn, v.zn);
    if (rc) return rc;
    rc = rt_scene_set_texcoords(sc, v.num_vt, v.u, v.v);
    if (rc) return rc;
    if (!o->textures.empty()) {
        std::vector<int> w, h;
        std::vector<const uint8_t*> px;
        for (const Image& im : o->textures) { w.push_back(im.w); h.push_back(im.h); px.push_back(im.rgb.data()); }
        rc = rt_scene_set_textures(sc, (int)w.size(), w.data(), h.data(), px.data());
        if (rc) return rc;
    }
    const bool tex = v.num_vt > 0;
    bool tex_ok = tex;
    for (int i = 0; tex_ok && i < v.num_tris; ++i)
        tex_ok = v.At[i] >= 1 && v.Bt[i] >= 1 && v.Ct[i] >= 1;
    // Faces without vt leave At/Bt/Ct uninitialised in the reference (obj_reader.c:96-104,145-160)
    // and its texture lookup then reads through them, which is undefined; here a mesh with any
    // such face is shaded with its materials' constant colours, like a failed IMG_Load.
    rc = rt_scene_set_materials(sc, v.num_mats, v.Ka, v.Kd, v.Ks, tex_ok ? v.map_Ka : nullptr,
                                tex_ok ? v.map_Kd : nullptr, tex_ok ? v.map_Ks : nullptr);
    if (rc) return rc;
    return rt_scene_set_triangles(sc, v.num_tris, v.A, v.B, v.C, v.An, v.Bn, v.Cn,
                                  tex_ok ? v.At : nullptr, tex_ok ? v.Bt : nullptr,
                                  tex_ok ? v.Ct : nullptr, v.mtl);
}
