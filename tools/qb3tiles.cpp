// tools/qb3tiles.cpp -- tile batcher over qb3x_encode_tiles / qb3x_decode_tiles (include/qb3x.h): the caller the
// reference does not ship but its users write.  GDAL's MRF driver keeps a raster as a grid of independently coded QB3
// tiles, one qb3_encode call per tile (reference README.md:30-31; the reference's own nearest pattern is cqb3's folder
// mode, cqb3.cpp:614-641: one file, one call).  Here a whole raster is cut into tiles, every tile is an independent
// QB3 container -- exactly what qb3_encode would write for it -- and all tiles of a call go through ONE set of kernel
// launches.
//
//   qb3tiles -e [-v] [-b|-f] [-t N] input.pnm out.qts          cut into N x N tiles (default 512), code, store
//   qb3tiles -d [-v] in.qts output.pnm                          decode every tile, reassemble the raster
//   qb3tiles -x in.qts K out.qb3                                extract tile K as a stand-alone .qb3 file
//
// Tile-set file (.qts): "QTS1", u32 raster width, height, bands, dtype, tile edge, tiles per row, tiles per column, then
// one (u64 offset, u64 size) pair per tile in row-major order, then the containers back to back.  Edge tiles are padded
// by repeating the raster's last column / row (every tile of a call has one geometry) and cropped on decode.
// Input is binary PNM (P5 / P6, 8 or 16 bit) like tools/cqb3x.cpp.
#include "QB3.h"
#include "qb3x.h"
#include <hip/hip_runtime_api.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

int fail(const std::string &msg) {
    fprintf(stderr, "qb3tiles: %s\n", msg.c_str());
    return 1;
}

bool read_file(const std::string &name, std::vector<uint8_t> &v) {
    FILE *f = fopen(name.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    v.resize(n > 0 ? (size_t)n : 0);
    bool ok = v.empty() || fread(v.data(), 1, v.size(), f) == v.size();
    fclose(f);
    return ok;
}

bool write_file(const std::string &name, const void *a, size_t na, const void *b = nullptr, size_t nb = 0) {
    FILE *f = fopen(name.c_str(), "wb");
    if (!f) return false;
    bool ok = (!na || fwrite(a, 1, na, f) == na) && (!nb || fwrite(b, 1, nb, f) == nb);
    return fclose(f) == 0 && ok;
}

bool pnm_number(const std::vector<uint8_t> &v, size_t &pos, size_t &val) {
    for (;;) {
        while (pos < v.size() && (v[pos] == ' ' || v[pos] == '\t' || v[pos] == '\n' || v[pos] == '\r')) pos++;
        if (pos < v.size() && v[pos] == '#') { while (pos < v.size() && v[pos] != '\n') pos++; continue; }
        break;
    }
    if (pos >= v.size() || v[pos] < '0' || v[pos] > '9') return false;
    val = 0;
    while (pos < v.size() && v[pos] >= '0' && v[pos] <= '9') val = val * 10 + (v[pos++] - '0');
    return true;
}

struct Raster { size_t w = 0, h = 0, bands = 0; int type = QB3_U8; std::vector<uint8_t> px; };

bool read_pnm(const std::string &name, Raster &r) {
    std::vector<uint8_t> v;
    if (!read_file(name, v) || v.size() < 8 || v[0] != 'P' || (v[1] != '5' && v[1] != '6')) return false;
    r.bands = v[1] == '5' ? 1 : 3;
    size_t pos = 2, maxval = 0;
    if (!pnm_number(v, pos, r.w) || !pnm_number(v, pos, r.h) || !pnm_number(v, pos, maxval) || pos >= v.size()) return false;
    pos++;                                              // the single whitespace after maxval
    r.type = maxval < 256 ? QB3_U8 : QB3_U16;
    const size_t tsz = r.type == QB3_U8 ? 1 : 2, n = r.w * r.h * r.bands * tsz;
    if (v.size() - pos < n) return false;
    r.px.assign(v.begin() + pos, v.begin() + pos + n);
    if (tsz == 2) for (size_t i = 0; i + 1 < n; i += 2) std::swap(r.px[i], r.px[i + 1]);      // PNM samples are big endian
    return true;
}

bool write_pnm(const std::string &name, Raster &r) {
    char hdr[64];
    snprintf(hdr, sizeof(hdr), "P%c\n%zu %zu\n%d\n", r.bands == 1 ? '5' : '6', r.w, r.h, r.type == QB3_U8 ? 255 : 65535);
    if (r.type != QB3_U8) for (size_t i = 0; i + 1 < r.px.size(); i += 2) std::swap(r.px[i], r.px[i + 1]);
    return write_file(name, hdr, strlen(hdr), r.px.data(), r.px.size());
}

struct DevMem {
    void *p = nullptr;
    explicit DevMem(size_t n) { if (hipMalloc(&p, n ? n : 4) != hipSuccess) p = nullptr; }
    ~DevMem() { if (p) (void)hipFree(p); }
};

#pragma pack(push, 1)
struct QtsHeader { char sig[4]; uint32_t w, h, bands, dtype, tile, tx, ty; };
struct QtsEntry { uint64_t off, size; };
#pragma pack(pop)

double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int encode(const std::string &in, const std::string &out, size_t T, int mode, bool verbose, int tables) {
    Raster r;
    if (!read_pnm(in, r)) return fail("cannot read " + in + " as binary PNM");
    if (T < 4 || T > 65536) return fail("tile edge out of range");
    const size_t tsz = r.type == QB3_U8 ? 1 : 2, pix = r.bands * tsz;
    const size_t tx = (r.w + T - 1) / T, ty = (r.h + T - 1) / T, n = tx * ty, raw = T * T * pix;
    // cut: tile (i, j) at tile index j * tx + i, padded by edge replication
    std::vector<uint8_t> tiles(n * raw);
    for (size_t j = 0; j < ty; j++)
        for (size_t i = 0; i < tx; i++) {
            uint8_t *t = tiles.data() + (j * tx + i) * raw;
            for (size_t y = 0; y < T; y++) {
                const size_t sy = std::min(j * T + y, r.h - 1);
                const uint8_t *row = r.px.data() + sy * r.w * pix;
                const size_t x0 = i * T, nx = x0 + T <= r.w ? T : r.w - x0;
                memcpy(t + y * T * pix, row + x0 * pix, nx * pix);
                for (size_t x = nx; x < T; x++) memcpy(t + (y * T + x) * pix, row + (r.w - 1) * pix, pix);
            }
        }
    encsp p = qb3_create_encoder(T, T, r.bands, (qb3_dtype)r.type);
    if (!p) return fail("qb3_create_encoder refused the tile geometry");
    qb3_set_encoder_mode(p, (qb3_mode)mode);
    if (tables) qb3x_set_encoder_index_chunk(p, tables);  // every tile carries its restart table (2: with block lengths): the set decodes in parallel inside every tile
    const size_t pitch = (qb3_max_encoded_size(p) + 3) / 4 * 4;
    DevMem d_src(n * raw), d_dst(n * pitch);
    if (!d_src.p || !d_dst.p) { qb3_destroy_encoder(p); return fail("out of device memory"); }
    std::vector<size_t> sizes(n);
    const double t0 = now();
    if (hipMemcpy(d_src.p, tiles.data(), tiles.size(), hipMemcpyHostToDevice) != hipSuccess) { qb3_destroy_encoder(p); return fail("upload failed"); }
    const double t1 = now();
    const size_t done = qb3x_encode_tiles(p, d_src.p, n, raw, d_dst.p, pitch, nullptr, sizes.data(), nullptr);
    const double t2 = now();
    qb3_destroy_encoder(p);
    if (done != n) return fail(std::string("qb3x_encode_tiles: ") + qb3x_last_error());
    // gather the containers: header, table, data
    QtsHeader h;
    memcpy(h.sig, "QTS1", 4);
    h.w = (uint32_t)r.w; h.h = (uint32_t)r.h; h.bands = (uint32_t)r.bands; h.dtype = (uint32_t)r.type; h.tile = (uint32_t)T; h.tx = (uint32_t)tx; h.ty = (uint32_t)ty;
    std::vector<QtsEntry> tab(n);
    uint64_t off = sizeof(h) + n * sizeof(QtsEntry), total = 0;
    for (size_t k = 0; k < n; k++) { tab[k].off = off; tab[k].size = sizes[k]; off += sizes[k]; total += sizes[k]; }
    std::vector<uint8_t> file(off);
    memcpy(file.data(), &h, sizeof(h));
    memcpy(file.data() + sizeof(h), tab.data(), n * sizeof(QtsEntry));
    for (size_t k = 0; k < n; k++)
        if (hipMemcpy(file.data() + tab[k].off, (const uint8_t *)d_dst.p + k * pitch, sizes[k], hipMemcpyDeviceToHost) != hipSuccess) return fail("download failed");
    if (!write_file(out, file.data(), file.size())) return fail("cannot write " + out);
    if (verbose)
        printf("%zu x %zu x %zu, %zu tiles of %zu^2: %zu -> %llu bytes (%.2f %%), upload %.2f ms, coding %.2f ms (%.1f MPixel/s)\n", r.w, r.h, r.bands, n, T,
               r.px.size(), (unsigned long long)total, 100.0 * total / r.px.size(), 1e3 * (t1 - t0), 1e3 * (t2 - t1), n * T * T / (t2 - t1) / 1e6);
    return 0;
}

bool parse_qts(const std::vector<uint8_t> &f, QtsHeader &h, std::vector<QtsEntry> &tab) {
    if (f.size() < sizeof(h) || memcmp(f.data(), "QTS1", 4)) return false;
    memcpy(&h, f.data(), sizeof(h));
    const size_t n = (size_t)h.tx * h.ty;
    if (!n || f.size() < sizeof(h) + n * sizeof(QtsEntry) || h.bands < 1 || h.bands > 16 || h.dtype > 7 || h.tile < 4) return false;
    tab.resize(n);
    memcpy(tab.data(), f.data() + sizeof(h), n * sizeof(QtsEntry));
    for (auto &e : tab) if (e.off > f.size() || e.size > f.size() - e.off || e.size < 15) return false;
    return true;
}

int decode(const std::string &in, const std::string &out, bool verbose) {
    std::vector<uint8_t> f;
    QtsHeader h;
    std::vector<QtsEntry> tab;
    if (!read_file(in, f) || !parse_qts(f, h, tab)) return fail(in + " is not a tile set");
    const size_t n = tab.size(), T = h.tile, tsz = (h.dtype < 2) ? 1 : (h.dtype < 4) ? 2 : (h.dtype < 6) ? 4 : 8, pix = h.bands * tsz, raw = T * T * pix;
    size_t pitch = 0;
    for (auto &e : tab) pitch = std::max(pitch, (size_t)e.size);
    pitch = (pitch + 3) / 4 * 4;
    // containers at a fixed pitch on the device; the handle is parsed from tile 0 (tiles of another kind -- raw-stored
    // ones -- are found and decoded on their own by qb3x_decode_tiles)
    DevMem d_src(n * pitch), d_dst(n * raw);
    if (!d_src.p || !d_dst.p) return fail("out of device memory");
    std::vector<size_t> sizes(n);
    for (size_t k = 0; k < n; k++) {
        sizes[k] = tab[k].size;
        if (hipMemcpy((uint8_t *)d_src.p + k * pitch, f.data() + tab[k].off, tab[k].size, hipMemcpyHostToDevice) != hipSuccess) return fail("upload failed");
    }
    size_t dims[3];
    decsp d = qb3_read_start(f.data() + tab[0].off, tab[0].size, dims);
    if (!d || !qb3_read_info(d) || dims[0] != T || dims[1] != T || dims[2] != h.bands) { if (d) qb3_destroy_decoder(d); return fail("tile 0 does not parse"); }
    const double t0 = now();
    const size_t done = qb3x_decode_tiles(d, d_src.p, n, pitch, sizes.data(), d_dst.p, raw, nullptr, nullptr);
    const double t1 = now();
    if (done != n) {
        std::string bad;
        for (size_t k = 0; k < n && bad.size() < 60; k++) if (!qb3x_decode_tile_ok(d, k)) bad += " " + std::to_string(k);
        qb3_destroy_decoder(d);
        return fail("tiles that did not decode:" + bad);
    }
    qb3_destroy_decoder(d);
    std::vector<uint8_t> tiles(n * raw);
    if (hipMemcpy(tiles.data(), d_dst.p, tiles.size(), hipMemcpyDeviceToHost) != hipSuccess) return fail("download failed");
    Raster r;
    r.w = h.w; r.h = h.h; r.bands = h.bands; r.type = (int)h.dtype;
    r.px.resize(r.w * r.h * pix);
    for (size_t j = 0; j < h.ty; j++)
        for (size_t i = 0; i < h.tx; i++) {
            const uint8_t *t = tiles.data() + (j * h.tx + i) * raw;
            const size_t x0 = i * T, nx = x0 + T <= r.w ? T : r.w - x0;
            for (size_t y = 0; y < T && j * T + y < r.h; y++) memcpy(r.px.data() + ((j * T + y) * r.w + x0) * pix, t + y * T * pix, nx * pix);
        }
    const bool pnm = (r.bands == 1 || r.bands == 3) && (r.type == QB3_U8 || r.type == QB3_U16);
    if (!(pnm ? write_pnm(out, r) : write_file(out, r.px.data(), r.px.size()))) return fail("cannot write " + out);
    if (verbose) printf("%zu tiles of %zu^2 decoded in %.2f ms (%.1f MPixel/s), %zu x %zu x %zu written%s\n", n, T, 1e3 * (t1 - t0), n * T * T / (t1 - t0) / 1e6,
                        r.w, r.h, r.bands, pnm ? "" : " (headerless)");
    return 0;
}

int extract(const std::string &in, size_t k, const std::string &out) {
    std::vector<uint8_t> f;
    QtsHeader h;
    std::vector<QtsEntry> tab;
    if (!read_file(in, f) || !parse_qts(f, h, tab)) return fail(in + " is not a tile set");
    if (k >= tab.size()) return fail("no such tile");
    return write_file(out, f.data() + tab[k].off, tab[k].size) ? 0 : fail("cannot write " + out);
}

}  // namespace

int main(int argc, char **argv) {
    bool enc = false, dec = false, ext = false, verbose = false;
    int tables = 0;
    int mode = QB3M_DEFAULT;
    size_t T = 512;
    std::vector<std::string> pos;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "-e") enc = true;
        else if (a == "-d") dec = true;
        else if (a == "-x") ext = true;
        else if (a == "-v") verbose = true;
        else if (a == "-b") mode = QB3M_BEST;
        else if (a == "-f") mode = QB3M_FTL;
        else if (a == "-i") tables = 1;
        else if (a == "-I") tables = 2;
        else if (a == "-t" && i + 1 < argc) T = strtoull(argv[++i], nullptr, 10);
        else pos.push_back(a);
    }
    if (qb3x_device_count() < 1) return fail("no usable HIP device (the block codec has no CPU fallback)");
    if (enc && pos.size() == 2) return encode(pos[0], pos[1], T, mode, verbose, tables);
    if (dec && pos.size() == 2) return decode(pos[0], pos[1], verbose);
    if (ext && pos.size() == 3) return extract(pos[0], strtoull(pos[1].c_str(), nullptr, 10), pos[2]);
    fprintf(stderr, "qb3tiles -e [-v] [-b|-f] [-i|-I] [-t N] input.pnm out.qts      (-i: tiles carry restart tables, -I: with block lengths)\nqb3tiles -d [-v] in.qts output.pnm\nqb3tiles -x in.qts K out.qb3\n");
    return 2;
}
