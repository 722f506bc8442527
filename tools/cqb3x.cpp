// tools/cqb3x.cpp -- command line caller of libQB3.so (MI355X), the counterpart of the reference's cqb3
// (cqb3.cpp:68-260 options, :405-493 encode flow, :276-323 decode flow).
//
// The reference converts PNG/JPEG <-> QB3 through libicd, which this image does not have; this tool speaks
// binary PNM instead (P5 = 1 band, P6 = 3 bands, maxval < 256 -> QB3_U8, otherwise QB3_U16 big endian) and
// headerless rasters (-s w,h,bands,type).  Everything it does goes through the 21-symbol C API of
// include/QB3.h, i.e. it builds against the reference library unchanged.
//
//   cqb3x [-v] [-b] [-f] [-l] [-r] [-t] [-q [+]N] [-m [map]] [-s w,h,bands,type] in [out.qb3]
//   cqb3x -d [-v] [-s] in.qb3 [out.pnm|out.raw]
//
// Option letters and their effect on the mode follow the reference tool:
//   -b best, -f fastest (FTL), -l legacy (Z curve), -r toggle the RLE0 pass, -t trim to multiples of 4,
//   -q quanta (a leading + rounds away from zero), -m band map ("-m" alone = identity), -d decode, -v verbose.
#include "QB3.h"
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

struct Options {
    bool verbose = false, best = false, ftl = false, legacy = false, rle = false, trim = false, decode = false;
    bool raw = false, have_map = false;
    uint64_t quanta = 1;
    bool away = false;
    std::string map, in, out;
    size_t rw = 0, rh = 0, rb = 0;
    int rtype = QB3_U8;
};

struct Raster {
    size_t w = 0, h = 0, bands = 0;
    int type = QB3_U8;
    std::vector<uint8_t> px;      // native byte order, band interleaved
};

const int kTypeSize[8] = { 1, 1, 2, 2, 4, 4, 8, 8 };

int fail(const std::string &msg) {
    fprintf(stderr, "cqb3x: %s\n", msg.c_str());
    return 1;
}

int usage(const char *why) {
    fprintf(stderr, "%s\n\n"
        "cqb3x [-v] [-b|-f] [-l] [-r] [-t] [-q [+]N] [-m [b0,b1,..]] [-s w,h,bands,type] input [output]\n"
        "cqb3x -d [-v] input.qb3 [output]\n"
        "  input is binary PNM (P5/P6) or, with -s, a headerless raster; type is 0..7 as qb3_dtype\n"
        "  -b best  -f fastest  -l legacy Z order  -r toggle RLE0  -t trim to x4  -q quantize  -m band map\n",
        why);
    return 2;
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

bool write_file(const std::string &name, const void *hdr, size_t nh, const void *body, size_t nb) {
    FILE *f = fopen(name.c_str(), "wb");
    if (!f) return false;
    bool ok = (!nh || fwrite(hdr, 1, nh, f) == nh) && (!nb || fwrite(body, 1, nb, f) == nb);
    return fclose(f) == 0 && ok;
}

// one whitespace/comment separated unsigned of a PNM header
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

bool parse_pnm(const std::vector<uint8_t> &file, Raster &r) {
    if (file.size() < 8 || file[0] != 'P' || (file[1] != '5' && file[1] != '6')) return false;
    size_t pos = 2, maxval = 0;
    r.bands = file[1] == '5' ? 1 : 3;
    if (!pnm_number(file, pos, r.w) || !pnm_number(file, pos, r.h) || !pnm_number(file, pos, maxval)) return false;
    pos++;      // the single whitespace that ends the header
    if (!maxval || maxval > 65535) return false;
    r.type = maxval < 256 ? QB3_U8 : QB3_U16;
    size_t n = r.w * r.h * r.bands * kTypeSize[r.type];
    if (pos + n > file.size()) return false;
    r.px.assign(file.begin() + pos, file.begin() + pos + n);
    if (r.type == QB3_U16)      // PNM samples are big endian
        for (size_t i = 0; i + 1 < n; i += 2) std::swap(r.px[i], r.px[i + 1]);
    return true;
}

qb3_mode pick_mode(const Options &o) {      // reference cqb3.cpp:435-462
    qb3_mode m = o.best ? QB3M_BEST : QB3M_BASE;
    if (o.legacy) m = (m == QB3M_BEST) ? QB3M_CF_RLE : QB3M_BASE_Z;
    if (o.rle) {
        if (m == QB3M_BEST) m = QB3M_CF_H;
        else if (m == QB3M_BASE) m = QB3M_RLE_H;
        else if (m == QB3M_BASE_Z) m = QB3M_RLE;
        else if (m == QB3M_CF_RLE) m = QB3M_CF;
    }
    return o.ftl ? QB3M_FTL : m;
}

int do_encode(const Options &o) {
    std::vector<uint8_t> file;
    if (!read_file(o.in, file)) return fail("can't read " + o.in);
    Raster r;
    if (o.raw) {
        r.w = o.rw; r.h = o.rh; r.bands = o.rb; r.type = o.rtype;
        if (file.size() < r.w * r.h * r.bands * kTypeSize[r.type]) return fail("raw input shorter than w*h*bands*typesize");
        r.px.swap(file);
    } else if (!parse_pnm(file, r))
        return fail("input is not a binary PNM (P5/P6); use -s for headerless rasters");
    // the encoder's line stride is in values, not bytes (QB3.h: "in dtype units"; reference cqb3.cpp:393,406)
    const size_t px_bytes = r.bands * kTypeSize[r.type], stride = r.w * r.bands;
    size_t offset = 0;
    if (o.trim) {           // drop the first column/line when that leaves the larger multiple of 4 (cqb3.cpp:393-402)
        if (r.w % 4 > 1) offset += px_bytes;
        if (r.h % 4 > 1) offset += stride * kTypeSize[r.type];
        r.w -= r.w % 4; r.h -= r.h % 4;
        if (o.verbose) printf("Trimmed to %zux%zu\n", r.w, r.h);
    }
    encsp e = qb3_create_encoder(r.w, r.h, r.bands, (qb3_dtype)r.type);
    if (!e) return fail("invalid raster shape for QB3");
    qb3_set_encoder_stride(e, stride);
    if (o.have_map) {
        size_t bmap[QB3_MAXBANDS];
        const char *s = o.map.c_str();
        for (size_t i = 0; i < r.bands; i++) {
            bmap[i] = i;
            if (*s) { char *end; bmap[i] = strtoul(s, &end, 10); while (*end == ',') end++; s = end; }
        }
        if (!qb3_set_encoder_coreband(e, r.bands, bmap)) fprintf(stderr, "Invalid band mapping, adjusted\n");
    }
    const qb3_mode mode = pick_mode(o);
    if (mode != qb3_set_encoder_mode(e, mode)) { qb3_destroy_encoder(e); return fail("invalid mode"); }
    if (o.quanta > 1) {
        if (!qb3_set_encoder_quanta(e, o.quanta, o.away)) { qb3_destroy_encoder(e); return fail("invalid quanta"); }
        if (o.verbose) printf("Lossy compression, quantized by %s%llu\n", o.away ? "+" : "", (unsigned long long)o.quanta);
    }
    std::vector<uint8_t> dest(qb3_max_encoded_size(e));
    auto t1 = std::chrono::steady_clock::now();
    size_t n = qb3_encode(e, r.px.data() + offset, dest.data());
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
    int err = qb3_get_encoder_state(e);
    qb3_destroy_encoder(e);
    if (!n) return fail("qb3_encode failed, state " + std::to_string(err));
    if (!write_file(o.out, nullptr, 0, dest.data(), n)) return fail("can't write " + o.out);
    if (o.verbose) {
        const double raw = double(r.w) * r.h * px_bytes;
        printf("%zux%zu@%zu type %d\nmode %d  %zu bytes  ratio %.4f  encode %.3f ms  %.1f MB/s\n", r.w, r.h, r.bands, r.type,
               (int)mode, n, n / raw, dt * 1e3, raw / dt / 1e6);
    }
    return 0;
}

int do_decode(const Options &o) {
    std::vector<uint8_t> file;
    if (!read_file(o.in, file)) return fail("can't read " + o.in);
    size_t dims[3];
    decsp d = qb3_read_start(file.data(), file.size(), dims);
    if (!d) return fail("not a QB3 stream");
    if (!qb3_read_info(d)) { qb3_destroy_decoder(d); return fail("bad QB3 headers"); }
    const int type = (int)qb3_get_type(d);
    std::vector<uint8_t> px(qb3_decoded_size(d));
    auto t1 = std::chrono::steady_clock::now();
    size_t n = qb3_read_data(d, px.data());
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
    const int mode = (int)qb3_get_mode(d);
    const unsigned long long q = qb3_get_quanta(d);
    qb3_destroy_decoder(d);
    if (!n) return fail("qb3_read_data failed");
    const bool pnm = !o.raw && (dims[2] == 1 || dims[2] == 3) && (type == QB3_U8 || type == QB3_U16);
    char hdr[64] = "";
    if (pnm) {
        snprintf(hdr, sizeof(hdr), "P%c\n%zu %zu\n%d\n", dims[2] == 1 ? '5' : '6', dims[0], dims[1], type == QB3_U8 ? 255 : 65535);
        if (type == QB3_U16) for (size_t i = 0; i + 1 < n; i += 2) std::swap(px[i], px[i + 1]);
    } else if (!o.raw)
        fprintf(stderr, "cqb3x: %zu bands of type %d do not fit PNM, writing a headerless raster\n", dims[2], type);
    if (!write_file(o.out, hdr, strlen(hdr), px.data(), n)) return fail("can't write " + o.out);
    if (o.verbose)
        printf("%zux%zu@%zu type %d mode %d quanta %llu\ndecode %.3f ms  %.1f MB/s\n", dims[0], dims[1], dims[2], type, mode, q,
               dt * 1e3, n / dt / 1e6);
    return 0;
}

}  // namespace

int main(int argc, char **argv) {
    Options o;
    for (int i = 1; i < argc; i++) {
        const char *a = argv[i];
        if (a[0] != '-' || !a[1]) {
            if (o.in.empty()) o.in = a;
            else if (o.out.empty()) o.out = a;
            else return usage("Too many positional arguments provided");
            continue;
        }
        const bool more = i + 1 < argc;
        switch (a[1]) {
        case 'v': o.verbose = true; break;
        case 'b': o.best = true; break;
        case 'd': o.decode = true; break;
        case 'f': o.ftl = true; break;
        case 't': o.trim = true; break;
        case 'l': o.legacy = true; break;
        case 'r': o.rle = true; break;
        case 'q':
            o.quanta = 2;
            if (more && (argv[i + 1][0] == '+' || (argv[i + 1][0] >= '0' && argv[i + 1][0] <= '9'))) {
                o.away = argv[i + 1][0] == '+';
                o.quanta = strtoull(argv[i + 1] + (o.away ? 1 : 0), nullptr, 10);
                i++;
            }
            break;
        case 'm':
            o.have_map = true;
            if (more && argv[i + 1][0] >= '0' && argv[i + 1][0] <= '9') o.map = argv[++i];
            break;
        case 's':
            o.raw = true;
            if (more) {     // a shape follows when encoding; "-d -s" just asks for a headerless output
                unsigned long long w, h, b; int t;
                if (4 == sscanf(argv[i + 1], "%llu,%llu,%llu,%d", &w, &h, &b, &t)) {
                    if (t < 0 || t > 7) return usage("-s takes w,h,bands,type with type 0..7");
                    o.rw = w; o.rh = h; o.rb = b; o.rtype = t; i++;
                }
            }
            break;
        default: return usage("Unknown option provided");
        }
    }
    if (o.in.empty()) return usage("Need at least the input file name");
    if (o.raw && !o.decode && !o.rw) return usage("-s needs w,h,bands,type when encoding");
    if (o.ftl) o.best = o.rle = o.legacy = false;
    if (o.decode && (o.trim || o.best)) return usage("-t and -b are invalid for QB3 decoding");
    if (o.out.empty()) {
        std::string stem = o.in;
        size_t dot = stem.find_last_of('.'), sep = stem.find_last_of("/\\");
        if (dot != std::string::npos && (sep == std::string::npos || dot > sep)) stem.resize(dot);
        o.out = stem + (o.decode ? (o.raw ? ".raw" : ".pnm") : ".qb3");
    }
    return o.decode ? do_decode(o) : do_encode(o);
}
