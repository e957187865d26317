// host_helpers.cpp — rtxh_* entry points: the caller side of the seam, restated so that a host
// program (or the Rust shim in INTEGRATION.md) can assemble an RtxSceneDesc the way the
// reference's main() assembles its Scene.  Pure host code; citations are path:line in the
// reference repository.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <array>
#include <fstream>
#include <string>
#include <new>
#include <vector>

#include "../../include/rtx.h"
#include "scene_prep.h"

namespace {

// --- PNG (RGB8, stored deflate blocks: no compression library needed) ---------------
uint32_t crc_table[256];
bool crc_ready = false;

void crc_init()
{
    for (uint32_t n = 0; n < 256; ++n) {
        uint32_t c = n;
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
        crc_table[n] = c;
    }
    crc_ready = true;
}

uint32_t crc_update(uint32_t crc, const uint8_t *p, size_t n)
{
    for (size_t i = 0; i < n; ++i) crc = crc_table[(crc ^ p[i]) & 0xFFu] ^ (crc >> 8);
    return crc;
}

void put_be32(std::vector<uint8_t> &v, uint32_t x)
{
    v.push_back(uint8_t(x >> 24)); v.push_back(uint8_t(x >> 16)); v.push_back(uint8_t(x >> 8)); v.push_back(uint8_t(x));
}

void put_chunk(std::vector<uint8_t> &png, const char type[4], const std::vector<uint8_t> &data)
{
    put_be32(png, static_cast<uint32_t>(data.size()));
    const size_t start = png.size();
    png.insert(png.end(), type, type + 4);
    png.insert(png.end(), data.begin(), data.end());
    const uint32_t crc = crc_update(0xFFFFFFFFu, png.data() + start, png.size() - start) ^ 0xFFFFFFFFu;
    put_be32(png, crc);
}

std::vector<std::string> split_on_space(const std::string &line)
{
    // Rust's str::split(" "): every single space separates, empty tokens are kept
    std::vector<std::string> tok;
    size_t start = 0;
    for (;;) {
        const size_t sp = line.find(' ', start);
        if (sp == std::string::npos) { tok.push_back(line.substr(start)); break; }
        tok.push_back(line.substr(start, sp - start));
        start = sp + 1;
    }
    return tok;
}

bool parse_f32(const std::string &s, float &out)
{
    if (s.empty()) return false;
    char *end = nullptr;
    out = std::strtof(s.c_str(), &end);
    return end && *end == '\0';
}

bool parse_index(const std::string &s, size_t &out)
{
    if (s.empty() || s[0] == '-' || s[0] == '+') return false;
    char *end = nullptr;
    const unsigned long long v = std::strtoull(s.c_str(), &end, 10);
    out = static_cast<size_t>(v);
    return end && *end == '\0';
}

}  // namespace

extern "C" {

void rtxh_camera_new(const float eye[3], const float look_at[3], const float up[3],
                     float u[3], float v[3], float w[3])
{
    rtx::camera_new(eye, look_at, up, u, v, w);
}

// import_obj — src/main.rs:114-149.  Only "v x y z" and "f i j k" lines mean anything; face
// indices are 1-based into the vertices read so far.  Where the reference would panic
// (unparsable number, index out of range, fewer than four tokens) this returns RTX_ERR_IO.
static int import_obj_impl(const char *path, float **v0v1v2)
{
    if (!path || !v0v1v2) return RTX_ERR_BAD_ARG;
    *v0v1v2 = nullptr;
    std::ifstream in(path);
    if (!in) return RTX_ERR_IO;
    std::vector<float> verts, tris;
    std::string line;
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();   // BufRead::lines strips "\r\n" too
        const std::vector<std::string> tok = split_on_space(line);
        if (tok[0] == "v") {                                         // :130-136
            if (tok.size() < 4) return RTX_ERR_IO;
            float xyz[3];
            for (int k = 0; k < 3; ++k)
                if (!parse_f32(tok[1 + k], xyz[k])) return RTX_ERR_IO;
            verts.insert(verts.end(), xyz, xyz + 3);
        } else if (tok[0] == "f") {                                  // :137-145
            if (tok.size() < 4) return RTX_ERR_IO;
            for (int k = 0; k < 3; ++k) {
                size_t id;
                if (!parse_index(tok[1 + k], id) || id < 1 || id > verts.size() / 3) return RTX_ERR_IO;
                tris.insert(tris.end(), verts.begin() + 3 * (id - 1), verts.begin() + 3 * id);
            }
        }
    }
    const size_t n = tris.size() / 9;
    if (n > 0x3FFFFFFFu) return RTX_ERR_UNSUPPORTED;
    float *out = static_cast<float *>(std::malloc(n ? tris.size() * sizeof(float) : sizeof(float)));
    if (!out) return RTX_ERR_OOM;
    if (n) std::memcpy(out, tris.data(), tris.size() * sizeof(float));
    *v0v1v2 = out;
    return static_cast<int>(n);
}

}  // extern "C"

namespace {

std::vector<std::string> split_on_blanks(const std::string &line)
{
    std::vector<std::string> tok;
    size_t i = 0;
    while (i < line.size()) {
        while (i < line.size() && (line[i] == ' ' || line[i] == '\t')) ++i;
        size_t j = i;
        while (j < line.size() && line[j] != ' ' && line[j] != '\t') ++j;
        if (j > i) tok.push_back(line.substr(i, j - i));
        i = j;
    }
    return tok;
}

// "newmtl name" ... "Kd r g b" -> name -> colour; anything else is ignored
bool read_mtl(const std::string &path, std::vector<std::pair<std::string, std::array<float, 3>>> &out)
{
    std::ifstream in(path);
    if (!in) return false;
    std::string line, current;
    bool have = false;
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        const std::vector<std::string> tok = split_on_blanks(line);
        if (tok.empty()) continue;
        if (tok[0] == "newmtl" && tok.size() >= 2) {
            current = tok[1];
            have = true;
            out.push_back({current, {1.0f, 1.0f, 1.0f}});
        } else if (tok[0] == "Kd" && tok.size() >= 4 && have) {
            std::array<float, 3> c;
            if (parse_f32(tok[1], c[0]) && parse_f32(tok[2], c[1]) && parse_f32(tok[3], c[2])) out.back().second = c;
        }
    }
    return true;
}

}  // namespace

extern "C" {

static int import_obj_ex_impl(const char *path, uint32_t flags, float **v0v1v2, float **rgb)
{
    if (!path || !v0v1v2 || (flags & ~RTXH_OBJ_ALL)) return RTX_ERR_BAD_ARG;
    *v0v1v2 = nullptr;
    if (rgb) *rgb = nullptr;
    std::vector<float> tris, cols;
    if (flags == 0u) {   // import_obj itself, every triangle Color::new(1,1,1) (src/main.rs:146)
        float *t = nullptr;
        const int n = rtxh_import_obj(path, &t);
        if (n < 0) return n;
        *v0v1v2 = t;
        if (rgb) {
            float *c = static_cast<float *>(std::malloc(n ? static_cast<size_t>(n) * 3 * sizeof(float) : sizeof(float)));
            if (!c) { std::free(t); *v0v1v2 = nullptr; return RTX_ERR_OOM; }
            for (size_t i = 0; i < static_cast<size_t>(n) * 3; ++i) c[i] = 1.0f;
            *rgb = c;
        }
        return n;
    }
    std::ifstream in(path);
    if (!in) return RTX_ERR_IO;
    const std::string spath(path);
    const size_t slash = spath.find_last_of("/\\");
    const std::string dir = slash == std::string::npos ? std::string() : spath.substr(0, slash + 1);
    std::vector<float> verts;
    std::vector<std::pair<std::string, std::array<float, 3>>> materials;
    std::array<float, 3> colour = {1.0f, 1.0f, 1.0f};
    std::string line;
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        const std::vector<std::string> tok = split_on_blanks(line);
        if (tok.empty() || tok[0][0] == '#') continue;
        if (tok[0] == "v") {
            if (tok.size() < 4) return RTX_ERR_IO;
            float xyz[3];
            for (int k = 0; k < 3; ++k)
                if (!parse_f32(tok[1 + k], xyz[k])) return RTX_ERR_IO;
            verts.insert(verts.end(), xyz, xyz + 3);
        } else if (tok[0] == "f") {
            if (tok.size() < 4) return RTX_ERR_IO;
            const size_t n_corners = (flags & RTXH_OBJ_POLYGONS) ? tok.size() - 1 : 3;
            std::vector<size_t> ids(n_corners);
            for (size_t k = 0; k < n_corners; ++k) {
                std::string t = tok[1 + k];
                const size_t sl = t.find('/');
                if (sl != std::string::npos) {
                    if (!(flags & RTXH_OBJ_SLASHES)) return RTX_ERR_IO;
                    t = t.substr(0, sl);
                }
                const size_t n_verts = verts.size() / 3;
                size_t id = 0;
                if (!t.empty() && t[0] == '-') {
                    size_t back = 0;
                    if (!(flags & RTXH_OBJ_RELATIVE) || !parse_index(t.substr(1), back) || back < 1 || back > n_verts)
                        return RTX_ERR_IO;
                    id = n_verts - back + 1;
                } else if (!parse_index(t, id) || id < 1 || id > n_verts) {
                    return RTX_ERR_IO;
                }
                ids[k] = id;
            }
            for (size_t k = 2; k < n_corners; ++k) {                 // fan around the first corner
                for (size_t id : {ids[0], ids[k - 1], ids[k]})
                    tris.insert(tris.end(), verts.begin() + 3 * (id - 1), verts.begin() + 3 * id);
                cols.insert(cols.end(), colour.begin(), colour.end());
            }
        } else if ((flags & RTXH_OBJ_MATERIALS) && tok[0] == "mtllib" && tok.size() >= 2) {
            (void)read_mtl(dir + tok[1], materials);
        } else if ((flags & RTXH_OBJ_MATERIALS) && tok[0] == "usemtl" && tok.size() >= 2) {
            colour = {1.0f, 1.0f, 1.0f};
            for (const auto &m : materials)
                if (m.first == tok[1]) colour = m.second;           // the last definition wins
        }
    }
    const size_t n = tris.size() / 9;
    if (n > 0x3FFFFFFFu) return RTX_ERR_UNSUPPORTED;
    float *t = static_cast<float *>(std::malloc(n ? tris.size() * sizeof(float) : sizeof(float)));
    float *c = rgb ? static_cast<float *>(std::malloc(n ? cols.size() * sizeof(float) : sizeof(float))) : nullptr;
    if (!t || (rgb && !c)) { std::free(t); std::free(c); return RTX_ERR_OOM; }
    if (n) std::memcpy(t, tris.data(), tris.size() * sizeof(float));
    if (n && c) std::memcpy(c, cols.data(), cols.size() * sizeof(float));
    *v0v1v2 = t;
    if (rgb) *rgb = c;
    return static_cast<int>(n);
}

void rtxh_free(void *p) { std::free(p); }

int rtxh_ref_leaf_rank(uint32_t n_tris, const float *v0v1v2, uint32_t *out_rank)
{
    try {
        return rtx::ref_leaf_rank(n_tris, v0v1v2, out_rank);
    } catch (...) {
        return RTX_ERR_OOM;
    }
}

// Deterministic stand-in for the table the reference fills from thread_rng (src/main.rs:260-265):
// splitmix64; each draw keeps the top 24 bits of the high word, f = bits * 2^-24 in [0,1)
// (the 24-bit construction rand 0.3 uses for f32); entry = (s.0, s.1) in draw order.
void rtxh_gen_samples(uint64_t seed, uint32_t n_pairs, float *out)
{
    uint64_t state = seed;
    const uint64_t total = 2ull * n_pairs;
    for (uint64_t i = 0; i < total; ++i) {
        state += 0x9E3779B97F4A7C15ull;
        uint64_t z = state;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        out[i] = static_cast<float>(static_cast<uint32_t>(z >> 40)) * (1.0f / 16777216.0f);
    }
}

// Synthetic soup of BASELINE.json configs[4] (see include/rtx.h).  f32 arithmetic, one rounding per operation:
// c = lo + f*(hi-lo) per axis, vertex = c + (2f-1).
int rtxh_synthetic_mesh(uint64_t seed, uint32_t n_tris, float *out)
{
    if (!out) return RTX_ERR_BAD_ARG;
    static const float lo[3] = {-92.4f, 32.7f, -60.5f}, hi[3] = {59.7f, 183.4f, 57.6f};
    uint64_t state = seed;
    auto draw = [&state]() {
        state += 0x9E3779B97F4A7C15ull;
        uint64_t z = state;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        return static_cast<float>(static_cast<uint32_t>(z >> 40)) * (1.0f / 16777216.0f);
    };
    for (uint32_t i = 0; i < n_tris;) {
        float c[3], v[9];
        for (int k = 0; k < 3; ++k) {
            const float span = hi[k] - lo[k];
            c[k] = lo[k] + draw() * span;
        }
        for (int k = 0; k < 9; ++k) {
            const float off = 2.0f * draw() - 1.0f;
            v[k] = c[k % 3] + off;
        }
        const float e1[3] = {v[3] - v[0], v[4] - v[1], v[5] - v[2]}, e2[3] = {v[6] - v[0], v[7] - v[1], v[8] - v[2]};
        const float nx = e1[1] * e2[2] - e1[2] * e2[1], ny = e1[2] * e2[0] - e1[0] * e2[2], nz = e1[0] * e2[1] - e1[1] * e2[0];
        if (nx == 0.0f && ny == 0.0f && nz == 0.0f) continue;   // zero area: draw this triangle again
        std::memcpy(out + 9 * static_cast<size_t>(i), v, sizeof v);
        ++i;
    }
    return RTX_OK;
}

int rtxh_write_png(const char *path, uint32_t width, uint32_t height, const uint8_t *rgb)
{
    if (!path || !rgb || !width || !height) return RTX_ERR_BAD_ARG;
    if (!crc_ready) crc_init();
    try {
        std::vector<uint8_t> png = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
        std::vector<uint8_t> ihdr;
        put_be32(ihdr, width);
        put_be32(ihdr, height);
        const uint8_t tail[5] = {8, 2, 0, 0, 0};   // 8-bit, truecolour, deflate, no filter, no interlace
        ihdr.insert(ihdr.end(), tail, tail + 5);
        put_chunk(png, "IHDR", ihdr);

        // raw scanlines: filter byte 0 + row
        const size_t row = static_cast<size_t>(width) * 3u;
        std::vector<uint8_t> raw;
        raw.reserve((row + 1) * height);
        for (uint32_t y = 0; y < height; ++y) {
            raw.push_back(0);
            raw.insert(raw.end(), rgb + y * row, rgb + (y + 1) * row);
        }
        std::vector<uint8_t> z = {0x78, 0x01};
        uint32_t a = 1, b = 0;   // adler32
        size_t pos = 0;
        while (pos < raw.size()) {
            const size_t n = std::min<size_t>(65535, raw.size() - pos);
            z.push_back(pos + n == raw.size() ? 1 : 0);
            z.push_back(uint8_t(n)); z.push_back(uint8_t(n >> 8));
            z.push_back(uint8_t(~n)); z.push_back(uint8_t((~n) >> 8));
            z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
            for (size_t i = 0; i < n; ++i) { a = (a + raw[pos + i]) % 65521u; b = (b + a) % 65521u; }
            pos += n;
        }
        put_be32(z, (b << 16) | a);
        put_chunk(png, "IDAT", z);
        put_chunk(png, "IEND", {});
        FILE *f = std::fopen(path, "wb");
        if (!f) return RTX_ERR_IO;
        const bool ok = std::fwrite(png.data(), 1, png.size(), f) == png.size();
        return (std::fclose(f) == 0 && ok) ? RTX_OK : RTX_ERR_IO;
    } catch (...) {
        return RTX_ERR_OOM;
    }
}

// Nothing may unwind through the C ABI into a Rust or C caller: a large or hostile OBJ can make a vector or a string
// throw (bad_alloc, length_error).
int rtxh_import_obj(const char *path, float **v0v1v2)
{
    try {
        return import_obj_impl(path, v0v1v2);
    } catch (const std::bad_alloc &) {
        if (v0v1v2) *v0v1v2 = nullptr;
        return RTX_ERR_OOM;
    } catch (...) {
        if (v0v1v2) *v0v1v2 = nullptr;
        return RTX_ERR_INTERNAL;
    }
}

int rtxh_import_obj_ex(const char *path, uint32_t flags, float **v0v1v2, float **rgb)
{
    try {
        return import_obj_ex_impl(path, flags, v0v1v2, rgb);
    } catch (const std::bad_alloc &) {
        if (v0v1v2) *v0v1v2 = nullptr;
        if (rgb) *rgb = nullptr;
        return RTX_ERR_OOM;
    } catch (...) {
        if (v0v1v2) *v0v1v2 = nullptr;
        if (rgb) *rgb = nullptr;
        return RTX_ERR_INTERNAL;
    }
}

int rtxh_scatter_tiles(uint8_t *frame, uint32_t height, uint32_t width, const uint8_t *packed, uint32_t first_tile,
                       uint32_t tile_stride, uint32_t tile_rows)
{
    if (!frame || !packed || !tile_rows || !tile_stride) return RTX_ERR_BAD_ARG;
    const size_t row_bytes = static_cast<size_t>(width) * 3u;
    size_t ly = 0;
    for (uint64_t t = first_tile; t * tile_rows < height; t += tile_stride) {
        const size_t r0 = static_cast<size_t>(t * tile_rows);
        const size_t n = (height - r0 < tile_rows) ? height - r0 : tile_rows;
        std::memcpy(frame + r0 * row_bytes, packed + ly * row_bytes, n * row_bytes);
        ly += n;
    }
    return RTX_OK;
}

}  // extern "C"
