/*
 * oracle/qb3o.c -- container format, handle semantics and type dispatch of the CPU restatement.
 * TEST INFRASTRUCTURE ONLY (see qb3o.h for scope, citations and how it is pinned).
 */
#include "qb3o.h"
#include "qb3o_bits.h"
#include <stdlib.h>
#include <limits.h>

#include <stdio.h>
/* debugging aid for the tests: when set, the encoder writes one character per unit (0 zero/one-bit unit,
 * N normal, C common factor, I index) */
FILE *qb3o_trace = NULL;
void qb3o_set_trace(const char *path) { if (qb3o_trace) fclose(qb3o_trace); qb3o_trace = path ? fopen(path, "w") : NULL; }

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)

#define T uint8_t
#define UB 3
#define SFX(n) CAT(n, _u8)
#include "qb3o_codec.inc"
#undef T
#undef UB
#undef SFX

#define T uint16_t
#define UB 4
#define SFX(n) CAT(n, _u16)
#include "qb3o_codec.inc"
#undef T
#undef UB
#undef SFX

#define T uint32_t
#define UB 5
#define SFX(n) CAT(n, _u32)
#include "qb3o_codec.inc"
#undef T
#undef UB
#undef SFX

#define T uint64_t
#define UB 6
#define SFX(n) CAT(n, _u64)
#include "qb3o_codec.inc"
#undef T
#undef UB
#undef SFX

static const int typesizes[8] = { 1, 1, 2, 2, 4, 4, 8, 8 };
int qb3o_typesize(int dtype) { return (dtype < 0 || dtype > QB3O_I64) ? 0 : typesizes[dtype]; }

/* ---------------- encoder handle (QB3encode.cpp:26-134) ---------------- */

void qb3o_encoder_reset(qb3o_encoder *p) {
    for (size_t c = 0; c < p->nbands; c++) p->band[c].prev = p->band[c].runbits = p->band[c].cf = 0;
    p->error = 0;
}

int qb3o_encoder_init(qb3o_encoder *p, size_t w, size_t h, size_t b, int dt) {
    if (w == 0 || w > 0x10000 || h == 0 || h > 0x10000 || b == 0 || b > QB3O_MAXBANDS || dt < 0 || dt > QB3O_I64)
        return 0;
    memset(p, 0, sizeof(*p));
    p->xsize = w; p->ysize = h; p->nbands = b; p->type = dt;
    p->quanta = 1; p->mode = QB3O_FTL;
    for (size_t c = 0; c < b; c++) p->cband[c] = c;
    if (b == 3 || b == 4) p->cband[0] = p->cband[2] = 1;
    qb3o_encoder_reset(p);
    qb3o_tables_init();
    return 1;
}

int qb3o_set_coreband(qb3o_encoder *p, size_t b, size_t *bands) {
    if (b != p->nbands) return 0;
    for (size_t i = 0; i < b; i++) p->cband[i] = (uint8_t)((bands[i] < b) ? bands[i] : i);
    for (size_t i = 0; i < b; i++) if (p->cband[i] != i) p->cband[p->cband[i]] = p->cband[i];
    for (size_t i = 0; i < b; i++) bands[i] = p->cband[i];
    return 1;
}

int qb3o_set_quanta(qb3o_encoder *p, uint64_t q, int away) {
    if (q < 1) return 0;
    p->quanta = q; p->away = away;
    if (q == 1) return 1;
    /* cumulative range checks, as the fall-through switch at QB3encode.cpp:96-107 */
    static const uint64_t lim[8] = { UINT8_MAX, INT8_MAX, UINT16_MAX, INT16_MAX, UINT32_MAX, INT32_MAX, UINT64_MAX, INT64_MAX };
    int bad = 0;
    static const int order[7] = { QB3O_I8, QB3O_U8, QB3O_I16, QB3O_U16, QB3O_I32, QB3O_U32, QB3O_I64 };
    int start = -1;
    for (int i = 0; i < 7; i++) if (order[i] == p->type) start = i;
    if (start >= 0) for (int i = start; i < 7; i++) bad |= q > lim[order[i]];
    return !bad;
}

int qb3o_set_mode(qb3o_encoder *p, int mode) {
    if (mode >= 0 && mode < QB3O_MODE_END) p->mode = mode;
    if (p->mode <= QB3O_CF_RLE_Z) p->order = QB3O_ZCURVE;   /* sticky, QB3encode.cpp:124-132 */
    return p->mode;
}

size_t qb3o_max_encoded_size(const qb3o_encoder *p) {
    size_t n = 16 * ((p->xsize + 3) / 4) * ((p->ysize + 3) / 4) * p->nbands;
    double bpv = 17.0 / 16.0 + 8 * qb3o_typesize(p->type);
    return 1024 + (size_t)(bpv * n / 8);
}

/* ---------------- quantization (QB3encode.cpp:137-186, QB3decode.cpp:77-107) ---------------- */

#define DEF_QUANT(NAME, TS)                                                                     \
static void NAME(TS *s, size_t n, uint64_t quanta, int away) {                                 \
    const TS q = (TS)quanta;                                                                    \
    if (q == 2) { for (size_t i = 0; i < n; i++) s[i] = away ? (TS)(s[i] / 2 + s[i] % 2) : (TS)(s[i] / 2); } \
    else if (q == 3) { for (size_t i = 0; i < n; i++) s[i] = (TS)(s[i] / 3 + (s[i] % 3) / 2); } \
    else if (q == 4) { for (size_t i = 0; i < n; i++) s[i] = away ? (TS)(s[i] / 4 + (s[i] % 4) / 2) : (TS)(s[i] / 4 + (s[i] % 4) / 3); } \
    else if (away) {                                                                            \
        const TS h = (TS)(q / 2 + q % 2);                                                       \
        for (size_t i = 0; i < n; i++) { TS v = s[i], m = (TS)(v % q);                          \
            s[i] = (TS)(v / q + (!(v < 0) & (m >= h)) - ((v < 0) & ((TS)(m + h) <= 0))); }      \
    } else {                                                                                    \
        const TS h = (TS)(q / 2);                                                               \
        for (size_t i = 0; i < n; i++) { TS v = s[i], m = (TS)(v % q);                          \
            s[i] = (TS)(v / q + (!(v < 0) & (m > h)) - ((v < 0) & ((TS)(m + h) < 0))); }        \
    }                                                                                           \
}
DEF_QUANT(quant_u8, uint8_t)   DEF_QUANT(quant_i8, int8_t)
DEF_QUANT(quant_u16, uint16_t) DEF_QUANT(quant_i16, int16_t)
DEF_QUANT(quant_u32, uint32_t) DEF_QUANT(quant_i32, int32_t)
DEF_QUANT(quant_u64, uint64_t) DEF_QUANT(quant_i64, int64_t)

static void quantize(void *buf, size_t n, int type, uint64_t q, int away) {
    switch (type) {
    case QB3O_U8: quant_u8((uint8_t *)buf, n, q, away); break;    case QB3O_I8: quant_i8((int8_t *)buf, n, q, away); break;
    case QB3O_U16: quant_u16((uint16_t *)buf, n, q, away); break; case QB3O_I16: quant_i16((int16_t *)buf, n, q, away); break;
    case QB3O_U32: quant_u32((uint32_t *)buf, n, q, away); break; case QB3O_I32: quant_i32((int32_t *)buf, n, q, away); break;
    case QB3O_U64: quant_u64((uint64_t *)buf, n, q, away); break; case QB3O_I64: quant_i64((int64_t *)buf, n, q, away); break;
    }
}

#define DEF_DEQUANT(NAME, TS, TMAX, TMIN, SIGNED)                                               \
static void NAME(TS *d, size_t n, uint64_t quanta) {                                           \
    const TS q = (TS)quanta, mai = (TS)(TMAX / q), mii = (TS)(TMIN / q);                        \
    for (size_t i = 0; i < n; i++) { TS v = d[i];                                               \
        d[i] = (v <= mai) ? (TS)(v * q) : TMAX;                                                 \
        if (SIGNED && q > 2 && v < mii) d[i] = TMIN; }                                          \
}
DEF_DEQUANT(dequant_u8, uint8_t, UINT8_MAX, 0, 0)     DEF_DEQUANT(dequant_i8, int8_t, INT8_MAX, INT8_MIN, 1)
DEF_DEQUANT(dequant_u16, uint16_t, UINT16_MAX, 0, 0)  DEF_DEQUANT(dequant_i16, int16_t, INT16_MAX, INT16_MIN, 1)
DEF_DEQUANT(dequant_u32, uint32_t, UINT32_MAX, 0, 0)  DEF_DEQUANT(dequant_i32, int32_t, INT32_MAX, INT32_MIN, 1)
DEF_DEQUANT(dequant_u64, uint64_t, UINT64_MAX, 0, 0)  DEF_DEQUANT(dequant_i64, int64_t, INT64_MAX, INT64_MIN, 1)

static void dequantize_line(void *buf, size_t n, int type, uint64_t q) {
    switch (type) {
    case QB3O_U8: dequant_u8((uint8_t *)buf, n, q); break;    case QB3O_I8: dequant_i8((int8_t *)buf, n, q); break;
    case QB3O_U16: dequant_u16((uint16_t *)buf, n, q); break; case QB3O_I16: dequant_i16((int16_t *)buf, n, q); break;
    case QB3O_U32: dequant_u32((uint32_t *)buf, n, q); break; case QB3O_I32: dequant_i32((int32_t *)buf, n, q); break;
    case QB3O_U64: dequant_u64((uint64_t *)buf, n, q); break; case QB3O_I64: dequant_i64((int64_t *)buf, n, q); break;
    }
}

/* ---------------- headers (QB3encode.cpp:189-268) ---------------- */

static void put_sig(qb3o_bw *s, const char *sig) { bw_put(s, (uint8_t)sig[0] | ((uint64_t)(uint8_t)sig[1] << 8), 16); }

static void write_headers(const qb3o_encoder *p, qb3o_bw *s) {
    bw_put(s, 0x80334251u, 32);                 /* "QB3\200" */
    bw_put(s, (p->xsize - 1) & 0xffff, 16);
    bw_put(s, (p->ysize - 1) & 0xffff, 16);
    bw_put(s, (p->nbands - 1) & 0xff, 8);
    bw_put(s, (uint8_t)p->type, 8);
    bw_put(s, (uint8_t)p->mode, 8);
    int diff = 0;
    for (size_t c = 0; c < p->nbands; c++) diff |= (p->cband[c] != c);
    if (p->mode != QB3O_STORED && diff) {
        put_sig(s, "CB"); bw_put(s, p->nbands, 16);
        for (size_t c = 0; c < p->nbands; c++) bw_put(s, p->cband[c] & 0xff, 8);
    }
    if (p->quanta >= 2) {
        unsigned qbytes = 1 + qb3o_topbit(p->quanta) / 8;
        put_sig(s, "QV"); bw_put(s, qbytes, 16);
        bw_put(s, qbytes < 8 ? p->quanta & (~0ull >> (64 - 8 * qbytes)) : p->quanta, 8 * qbytes);
    }
    if (p->order != QB3O_ZCURVE && p->mode != QB3O_STORED) {
        put_sig(s, "SC"); bw_put(s, 8, 16);
        bw_put(s, p->order ? p->order : QB3O_HILBERT, 64);
    }
    put_sig(s, "DT");
}

static size_t raw_size(const qb3o_encoder *p) { return p->xsize * p->ysize * p->nbands * (size_t)qb3o_typesize(p->type); }

/* QB3encode.cpp:461-485; the strided copy is done correctly here (reference defect B-7) */
static size_t stored_encode(qb3o_encoder *p, const void *src, void *dst) {
    qb3o_bw s; bw_init(&s, (uint8_t *)dst);
    p->mode = QB3O_STORED;
    write_headers(p, &s);
    if (p->error) return 0;
    size_t hdr = bw_flush(&s);
    const size_t tsz = (size_t)qb3o_typesize(p->type), line = p->xsize * p->nbands * tsz;
    const size_t stride = (p->stride ? p->stride : p->xsize * p->nbands) * tsz;
    for (size_t y = 0; y < p->ysize; y++)
        memcpy((uint8_t *)dst + hdr + y * line, (const uint8_t *)src + y * stride, line);
    return hdr + raw_size(p);
}

/* ---------------- RLE0 (QB3encode.cpp:271-332, QB3decode.cpp:267-307) ---------------- */

static size_t zero_run(const uint8_t *s, size_t len) {
    if (len > 0xfe) len = 0xfe;
    size_t i = 0;
    while (i < len && !s[i]) i++;
    return i;
}

/* one pass, dst == NULL only counts */
static size_t rle0_pass(const uint8_t *src, size_t len, uint8_t *dst) {
    size_t i = 0, o = 0;
    uint8_t last = 0;
    while (i + 2 < len) {
        uint8_t c = src[i++];
        const size_t rem = len - i;     /* bytes after c */
        int special = (c == 0 || c == 0xff) && c == src[i];
        if (special && c == 0 && (last == 0xff || rem < 3 || src[i + 1] || src[i + 2])) special = 0;
        if (!special) { if (dst) dst[o] = c; o++; last = c; continue; }
        i++;
        if (c == 0) { i += 2; size_t r = zero_run(src + i, len - i); i += r; c = (uint8_t)r; }
        last = 0;
        if (dst) { dst[o] = 0xff; dst[o + 1] = 0xff; dst[o + 2] = c; }
        o += 3;
    }
    while (i < len) { if (dst) dst[o] = src[i]; o++; i++; }
    return o;
}
size_t qb3o_rle0(const uint8_t *src, size_t len, uint8_t *dst) { return rle0_pass(src, len, dst); }
size_t qb3o_rle0_size(const uint8_t *src, size_t len) { return rle0_pass(src, len, NULL); }

int64_t qb3o_derle0(const uint8_t *src, size_t slen, uint8_t *d, size_t dlen) {
    size_t i = 0, o = 0;
    while (o < dlen && i + 2 < slen) {
        uint8_t c = src[i++];
        if (c != 0xff || src[i] != 0xff) { d[o++] = c; continue; }
        size_t count = 2;
        if (src[i + 1] != 0xff) { c = 0; count = 4 + (size_t)src[i + 1]; }
        if (dlen - o < count) return (int64_t)o - (int64_t)dlen;
        i += 2;
        while (count--) d[o++] = c;
    }
    while (i < slen && o < dlen) d[o++] = src[i++];
    return (int64_t)(dlen - o) - (int64_t)(slen - i);
}

size_t qb3o_derle0_size(const uint8_t *src, size_t len) {
    size_t i = 0, count = 0;
    while (i + 2 < len) {
        if (src[i] != 0xff || src[i + 1] != 0xff) { count++; i++; continue; }
        count += (src[i + 2] == 0xff) ? 2 : 4 + (size_t)src[i + 2];
        i += 3;
    }
    return count + (len - i);
}

/* ---------------- encode (QB3encode.cpp:345-459, 488-574) ---------------- */

static int is_fast(int mode) { return mode == QB3O_BASE_H || mode == QB3O_BASE_Z || mode == QB3O_FTL; }

static int encode_typed(const void *src, qb3o_bw *s, qb3o_encoder *p) {
    const int best = !is_fast(p->mode), dostep = (p->mode != QB3O_FTL);
    switch (p->type) {
    case QB3O_U8: case QB3O_I8:   return encode_u8((const uint8_t *)src, s, p, best, dostep);
    case QB3O_U16: case QB3O_I16: return encode_u16((const uint16_t *)src, s, p, best, dostep);
    case QB3O_U32: case QB3O_I32: return encode_u32((const uint32_t *)src, s, p, best, dostep);
    case QB3O_U64: case QB3O_I64: return encode_u64((const uint64_t *)src, s, p, best, dostep);
    }
    return 1;
}

/* block stream for the whole image, including the narrow-image remap and the quantised path */
static int enc_image(const void *source, qb3o_bw *s, qb3o_encoder *p) {
    const size_t tsz = (size_t)qb3o_typesize(p->type);
    qb3o_encoder small;
    uint8_t *tmp = NULL, *qbuf = NULL;
    int err;
    if (p->xsize < 4 || p->ysize < 4) {
        /* QB3encode.cpp:351-389, implemented per its intent (the reference has a use-after-scope here, defect B-3) */
        small = *p;
        const size_t ngroups = (p->xsize * p->ysize + 15) / 16, pix = p->nbands * tsz;
        tmp = (uint8_t *)calloc(ngroups * 16 * p->nbands, tsz);
        const size_t stride = (p->stride ? p->stride : p->xsize * p->nbands) * tsz;
        uint8_t *d = tmp;
        if (p->xsize < 4) {
            for (size_t y = 0; y < p->ysize; y++, d += p->xsize * pix)
                memcpy(d, (const uint8_t *)source + y * stride, p->xsize * pix);
            small.xsize = 4; small.ysize = ngroups * 4;
        } else {
            for (size_t x = 0; x < p->xsize; x++)
                for (size_t y = 0; y < p->ysize; y++, d += pix)
                    memcpy(d, (const uint8_t *)source + y * stride + x * pix, pix);
            small.xsize = ngroups * 4; small.ysize = 4;
        }
        small.stride = 0;
        source = tmp; p = &small;
    }
    if (p->quanta < 2)
        err = encode_typed(source, s, p);
    else {
        /* QB3encode.cpp:405-455 quantises 4-line strips into a scratch copy and carries the band state
         * across strips on a COPY of the handle; equivalent to encoding a quantised copy of the image. */
        qb3o_encoder sub = *p;
        const size_t line = p->xsize * p->nbands * tsz;
        const size_t stride = (p->stride ? p->stride : p->xsize * p->nbands) * tsz;
        qbuf = (uint8_t *)malloc(line * p->ysize);
        for (size_t y = 0; y < p->ysize; y++) memcpy(qbuf + y * line, (const uint8_t *)source + y * stride, line);
        quantize(qbuf, p->xsize * p->ysize * p->nbands, p->type, p->quanta, p->away);
        sub.stride = 0;
        err = encode_typed(qbuf, s, &sub);
    }
    free(tmp); free(qbuf);
    return err;
}

uint64_t qb3o_encode_raw(qb3o_encoder *p, const void *src, void *dst) {
    qb3o_tables_init();
    qb3o_bw s; bw_init(&s, (uint8_t *)dst);
    p->error = enc_image(src, &s, p);
    uint64_t bits = bw_bits(&s);
    bw_flush(&s);
    return p->error ? 0 : bits;
}

size_t qb3o_encode(qb3o_encoder *p, const void *source, void *destination) {
    qb3o_tables_init();
    if (p->xsize * p->ysize <= 16) return stored_encode(p, source, destination);
    const int mode = p->mode;
    const int rle = (mode == QB3O_RLE_Z || mode == QB3O_CF_RLE_Z || mode == QB3O_RLE_H || mode == QB3O_CF_RLE_H);
    if (rle) p->mode = mode - 2;       /* 2->0, 3->1, 6->4, 7->5 (QB3encode.cpp:497-501) */
    uint8_t *d = (uint8_t *)destination;
    qb3o_bw s; bw_init(&s, d);
    write_headers(p, &s);
    const size_t data_position = (size_t)(bw_bits(&s) / 8);
    if (p->error) return 0;     /* a stale error blocks the handle until reset (QB3encode.cpp:514) */
    p->error = enc_image(source, &s, p);
    const size_t len = bw_flush(&s);
    if (rle) {
        p->mode = mode;
        if (p->error) return 0;
        if (len <= qb3o_max_encoded_size(p) / 2) {
            const size_t data_size = len - data_position, available = qb3o_max_encoded_size(p) - len;
            const size_t rle_size = qb3o_rle0_size(d + data_position, data_size);
            if (rle_size <= available && rle_size < data_size) {
                qb3o_rle0(d + data_position, data_size, d + len);
                qb3o_bw h; bw_init(&h, d);
                write_headers(p, &h);
                const size_t hdr = bw_flush(&h);
                memmove(d + hdr, d + len, rle_size);
                return hdr + rle_size;
            }
        }
    }
    if (p->error) return 0;
    if (raw_size(p) > len) return len;
    return stored_encode(p, source, destination);
}

/* ---------------- decode (QB3decode.cpp:130-264, 316-464) ---------------- */

int qb3o_read_start(qb3o_decoder *p, const void *source, size_t n, size_t *dims) {
    qb3o_tables_init();
    if (n < 15 || !dims) return 0;
    const uint8_t *b = (const uint8_t *)source;
    if (b[0] != 'Q' || b[1] != 'B' || b[2] != '3' || b[3] != 0x80) return 0;
    memset(p, 0, sizeof(*p));
    p->xsize = 1 + (size_t)(b[4] | (b[5] << 8));
    p->ysize = 1 + (size_t)(b[6] | (b[7] << 8));
    p->nbands = 1 + (size_t)b[8];
    p->type = b[9];
    p->mode = b[10];
    if (p->nbands > QB3O_MAXBANDS || (p->mode >= QB3O_MODE_END && p->mode != QB3O_STORED)
        || ((b[11] | b[12]) & 0x80) || p->type > QB3O_I64)
        return 0;
    p->s_in = b + 11; p->s_size = n - 11;
    dims[0] = p->xsize; dims[1] = p->ysize; dims[2] = p->nbands;
    if (p->mode <= QB3O_CF_RLE_Z) p->order = QB3O_ZCURVE;
    p->stage = 1;
    return 1;
}

static int valid_curve(uint64_t v) {
    unsigned mask = 0;
    for (int i = 0; i < 16; i++, v >>= 4) mask |= 1u << (v & 15);
    return mask == 0xffff;
}

int qb3o_read_info(qb3o_decoder *p) {
    if (p->stage != 1 || p->error || !p->s_in || p->s_size < 4) { if (!p->error) p->error = 1; return 0; }
    qb3o_br s; br_init(&s, p->s_in, p->s_size);
    do {
        uint64_t val = br_peek(&s);
        unsigned c0 = val & 0xff, c1 = (val >> 8) & 0xff, len = (val >> 16) & 0xffff;
        if (c0 == 'Q' && c1 == 'V') {
            if (len > 4 || len < 1) { p->error = 1; break; }
            br_adv(&s, 32);
            p->quanta = br_pull(&s, len * 8);
            if (p->quanta < 2) p->error = 1;
        } else if (c0 == 'C' && c1 == 'B') {
            if (len != p->nbands) { p->error = 1; break; }
            br_adv(&s, 32);
            for (size_t i = 0; i < p->nbands; i++) {
                p->cband[i] = (uint8_t)br_pull(&s, 8);
                if (p->cband[i] >= p->nbands) p->error = 1;
            }
            p->identity_cband = 0;      /* an explicit map always wins */
        } else if (c0 == 'D' && c1 == 'T') {
            br_adv(&s, 16);
            size_t used = (size_t)(s.pos / 8);
            if (p->s_size <= used) { p->error = 1; break; }
            p->s_in += used; p->s_size -= used; p->stage = 2;
        } else if (c0 == 'S' && c1 == 'C') {
            if (len != 8) { p->error = 1; break; }
            if (p->mode < QB3O_BASE_H || p->mode == QB3O_STORED) { p->error = 1; break; }
            br_adv(&s, 32);
            p->order = br_pull(&s, 64);
            if (!valid_curve(p->order)) { p->error = 1; break; }
        } else {
            /* QB3decode.cpp:251-258: lower-case chunks are skipped by len bytes FROM THE CHUNK START (defect B-7) */
            if ((c0 & 0x20) && len) br_adv(&s, (uint64_t)len * 8);
            else p->error = 2;
        }
    } while (p->stage != 2 && !p->error && br_avail(&s));
    if (!p->error && p->stage != 2) p->error = 1;
    return !p->error;
}

size_t qb3o_decoded_size(const qb3o_decoder *p) { return p->xsize * p->ysize * p->nbands * (size_t)qb3o_typesize(p->type); }

static int decode_typed(const qb3o_decoder *p, const uint8_t *src, size_t len, void *dst) {
    switch (p->type) {
    case QB3O_U8: case QB3O_I8:   return decode_u8(src, len, (uint8_t *)dst, p);
    case QB3O_U16: case QB3O_I16: return decode_u16(src, len, (uint16_t *)dst, p);
    case QB3O_U32: case QB3O_I32: return decode_u32(src, len, (uint32_t *)dst, p);
    case QB3O_U64: case QB3O_I64: return decode_u64(src, len, (uint64_t *)dst, p);
    }
    return 3;
}

int qb3o_decode_raw(const qb3o_decoder *pin, const uint8_t *src, size_t len, void *dst) {
    qb3o_tables_init();
    qb3o_decoder d = *pin;
    const qb3o_decoder *p = &d;
    if (d.identity_cband) for (size_t c = 0; c < d.nbands; c++) d.cband[c] = (uint8_t)c;
    if (p->xsize >= 4 && p->ysize >= 4) return decode_typed(p, src, len, dst);
    /* narrow image: decode the remapped image, then scatter (QB3decode.cpp:321-353) */
    const size_t tsz = (size_t)qb3o_typesize(p->type), pix = p->nbands * tsz;
    const size_t ngroups = (p->xsize * p->ysize + 15) / 16;
    qb3o_decoder a = d;
    a.stride = 0;
    a.xsize = p->xsize < 4 ? 4 : ngroups * 4;
    a.ysize = p->xsize < 4 ? ngroups * 4 : 4;
    uint8_t *tmp = (uint8_t *)calloc(ngroups * 16 * p->nbands, tsz);
    int err = decode_typed(&a, src, len, tmp);
    if (!err) {
        const size_t stride = (p->stride ? p->stride : p->xsize * p->nbands) * tsz;
        const uint8_t *s = tmp;
        if (p->xsize < 4)
            for (size_t y = 0; y < p->ysize; y++, s += p->xsize * pix) memcpy((uint8_t *)dst + y * stride, s, p->xsize * pix);
        else
            for (size_t x = 0; x < p->xsize; x++)
                for (size_t y = 0; y < p->ysize; y++, s += pix) memcpy((uint8_t *)dst + y * stride + x * pix, s, pix);
    }
    free(tmp);
    return err;
}

size_t qb3o_read_data(qb3o_decoder *p, void *dst) {
    if (p->stage != 2 || p->error || !p->s_in || !p->s_size) { if (!p->error) p->error = 1; return 0; }
    const uint8_t *src = p->s_in;
    size_t n = p->s_size;
    const size_t tsz = (size_t)qb3o_typesize(p->type), line = p->xsize * p->nbands * tsz;
    if (p->mode == QB3O_STORED) {
        if (n != qb3o_decoded_size(p)) { p->error = 1; return 0; }
        if (!p->stride) memcpy(dst, src, n);
        else for (size_t y = 0; y < p->ysize; y++) memcpy((uint8_t *)dst + y * p->stride, src + y * line, line);
        return n;
    }
    if (p->xsize * p->ysize < 16) { p->error = 1; return 0; }
    uint8_t *buf = NULL;
    if (p->mode == QB3O_RLE_Z || p->mode == QB3O_CF_RLE_Z || p->mode == QB3O_RLE_H || p->mode == QB3O_CF_RLE_H) {
        size_t sz = qb3o_derle0_size(src, n);
        if (sz > qb3o_decoded_size(p)) { p->error = 3; return 0; }
        buf = (uint8_t *)malloc(sz ? sz : 1);
        if (qb3o_derle0(src, n, buf, sz)) { free(buf); p->error = 1; return 0; }
        src = buf; n = sz;
    }
    int err = qb3o_decode_raw(p, src, n, dst);
    free(buf);
    if (!err && p->quanta > 1) {
        /* the reference takes the decoder stride in bytes here (QB3decode.cpp:83) */
        const size_t stride = p->stride ? p->stride : line;
        for (size_t y = 0; y < p->ysize; y++)
            dequantize_line((uint8_t *)dst + y * stride, p->xsize * p->nbands, p->type, p->quanta);
    }
    return err ? 0 : qb3o_decoded_size(p);
}

/* ---------------- heap handles and field setters, for the ctypes test harness ---------------- */

qb3o_encoder *qb3o_encoder_new(size_t w, size_t h, size_t b, int dt) {
    qb3o_encoder *p = (qb3o_encoder *)malloc(sizeof(*p));
    if (p && !qb3o_encoder_init(p, w, h, b, dt)) { free(p); p = NULL; }
    return p;
}
qb3o_decoder *qb3o_decoder_new(const void *src, size_t n, size_t *dims) {
    qb3o_decoder *p = (qb3o_decoder *)malloc(sizeof(*p));
    if (p && !qb3o_read_start(p, src, n, dims)) { free(p); p = NULL; }
    return p;
}
void qb3o_free(void *p) { free(p); }
void qb3o_set_stride(qb3o_encoder *p, size_t stride) { p->stride = stride; }
void qb3o_set_fix_b2(qb3o_encoder *p, int on) { p->fix_b2 = on; }
/* the reference's API cannot set a custom curve, but its format carries one (SC chunk) and its decoder honours it */
void qb3o_set_order(qb3o_encoder *p, uint64_t order) { p->order = order; }
int  qb3o_get_error(const qb3o_encoder *p) { return p->error; }
int  qb3o_get_encoder_mode(const qb3o_encoder *p) { return p->mode; }
void qb3o_get_band_state(const qb3o_encoder *p, uint64_t *out3n) {
    for (size_t c = 0; c < p->nbands; c++) { out3n[3 * c] = p->band[c].prev; out3n[3 * c + 1] = p->band[c].runbits; out3n[3 * c + 2] = p->band[c].cf; }
}
void qb3o_decoder_set_stride(qb3o_decoder *p, size_t stride) { p->stride = stride; }
void qb3o_decoder_set_identity(qb3o_decoder *p, int on) { p->identity_cband = on; }
int  qb3o_decoder_mode(const qb3o_decoder *p) { return p->stage == 2 ? p->mode : -1; }
int  qb3o_decoder_type(const qb3o_decoder *p) { return p->type; }
int  qb3o_decoder_error(const qb3o_decoder *p) { return p->error; }
uint64_t qb3o_decoder_quanta(const qb3o_decoder *p) { return p->stage == 2 ? p->quanta : 0; }
uint64_t qb3o_decoder_order(const qb3o_decoder *p) { return p->stage != 2 ? 0 : (p->order ? p->order : QB3O_ZCURVE); }
int  qb3o_decoder_coreband(const qb3o_decoder *p, size_t *cb) {
    if (p->stage != 2) return 0;
    for (size_t c = 0; c < p->nbands; c++) cb[c] = p->cband[c];
    return 1;
}
