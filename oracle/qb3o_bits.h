/*
 * oracle/qb3o_bits.h -- LSB-first bit I/O and the QB3 code rules, shared by all type instances.
 * TEST INFRASTRUCTURE ONLY (see qb3o.h).
 *
 * Bit order follows bitstream.h:25-126: bit k of the stream is bit (k&7) of byte (k>>3).
 * Code rules follow QB3encode.h:132-141 (three-length code) and the middle swap described at
 * QB3encode.h:30-33 / inline tables :185-186, :196-197; rung-switch code QB3encode.h:79-89.
 * All tables here are GENERATED from those rules at first use, none is transcribed.
 */
#ifndef QB3O_BITS_H
#define QB3O_BITS_H
#include <stdint.h>
#include <stddef.h>
#include <string.h>

/* ---------- writer ---------- */
typedef struct {
    uint8_t *base;
    size_t bytepos;     /* bytes already stored */
    uint64_t acc;       /* pending bits, LSB first */
    unsigned n;         /* number of pending bits, < 64 */
} qb3o_bw;

static inline void bw_init(qb3o_bw *w, uint8_t *dst) { w->base = dst; w->bytepos = 0; w->acc = 0; w->n = 0; }
static inline uint64_t bw_bits(const qb3o_bw *w) { return (uint64_t)w->bytepos * 8 + w->n; }

/* append the low nbits (0..64) of v; v must have no bits set at or above nbits */
static inline void bw_put(qb3o_bw *w, uint64_t v, unsigned nbits) {
    if (!nbits) return;
    w->acc |= v << w->n;
    if (w->n + nbits >= 64) {
        memcpy(w->base + w->bytepos, &w->acc, 8);
        w->bytepos += 8;
        unsigned used = 64 - w->n;          /* bits of v consumed, 1..64 */
        w->acc = used < 64 ? v >> used : 0;
        w->n = w->n + nbits - 64;
    } else
        w->n += nbits;
}

/* store pending bits (zero padded to a byte); returns total size in bytes */
static inline size_t bw_flush(qb3o_bw *w) {
    unsigned nb = (w->n + 7) / 8;
    memcpy(w->base + w->bytepos, &w->acc, nb);
    return w->bytepos + nb;
}

/* append nbits from a flushed byte buffer */
static inline void bw_append(qb3o_bw *w, const uint8_t *src, uint64_t nbits) {
    while (nbits >= 64) {
        uint64_t v; memcpy(&v, src, 8); src += 8; nbits -= 64;
        bw_put(w, v, 64);
    }
    if (nbits) {
        uint64_t v = 0; memcpy(&v, src, (size_t)(nbits + 7) / 8);
        bw_put(w, v & (~0ull >> (64 - nbits)), (unsigned)nbits);
    }
}

/* ---------- reader: never reads past the end, position clamps at the end (bitstream.h:36) ---------- */
typedef struct {
    const uint8_t *p;
    uint64_t len;   /* bits */
    uint64_t pos;   /* bits */
} qb3o_br;

static inline void br_init(qb3o_br *r, const uint8_t *src, size_t bytes) { r->p = src; r->len = (uint64_t)bytes * 8; r->pos = 0; }
static inline uint64_t br_avail(const qb3o_br *r) { return r->len - r->pos; }
static inline void br_adv(qb3o_br *r, uint64_t d) { r->pos = (r->pos + d < r->len) ? r->pos + d : r->len; }

static inline uint64_t br_peek(const qb3o_br *r) {
    uint64_t byte = r->pos >> 3; unsigned sh = (unsigned)(r->pos & 7);
    uint64_t total = r->len >> 3;
    if (byte + 9 <= total) {
        uint64_t lo; memcpy(&lo, r->p + byte, 8);
        if (!sh) return lo;
        return (lo >> sh) | ((uint64_t)r->p[byte + 8] << (64 - sh));
    }
    uint64_t v = 0;
    for (unsigned i = 0; i < 9 && byte + i < total; i++) {
        uint64_t b = r->p[byte + i];
        if (i * 8 >= sh) { if (i * 8 - sh < 64) v |= b << (i * 8 - sh); }
        else v |= b >> sh;
    }
    return v;
}
static inline uint64_t br_pull(qb3o_br *r, unsigned n) {
    uint64_t v = br_peek(r) & (~0ull >> (64 - n));
    br_adv(r, n);
    return v;
}

/* ---------- the three-length value code, rung r >= 1, no swap ---------- */
/* returns length; *lo = low 64 code bits, *hi = bit 64 (only for r == 63 long codes) */
static inline unsigned qb3o_code(uint64_t v, unsigned r, uint64_t *lo, unsigned *hi) {
    *hi = 0;
    const uint64_t half = 1ull << (r - 1);
    if (v < half) { *lo = v << 1; return r; }                               /* short   .x0 */
    if ((v >> r) == 0) { *lo = ((v - half) << 2) | 1; return r + 1; }       /* nominal .01 */
    uint64_t pay = v - (1ull << r);                                         /* long    .11 */
    *lo = (pay << 2) | 3;
    *hi = (unsigned)((pay >> 62) & 1);
    return r + 2;
}

static inline void bw_put_code(qb3o_bw *w, uint64_t v, unsigned r) {
    uint64_t lo; unsigned hi;
    unsigned len = qb3o_code(v, r, &lo, &hi);
    if (len <= 64) bw_put(w, lo, len);
    else { bw_put(w, lo, 64); bw_put(w, hi, 1); }
}

/* inverse of qb3o_code given the next 64 stream bits; returns length (up to 65).
 * For a 65 bit code value bit 62 is NOT filled in (it is the next stream bit). */
static inline unsigned qb3o_decode_code(uint64_t acc, unsigned r, uint64_t *v) {
    const uint64_t rb = 1ull << r;
    if (!(acc & 1)) { *v = (acc & (rb - 1)) >> 1; return r; }
    if (!(acc & 2)) { *v = ((acc >> 2) & ((rb >> 1) - 1)) | (rb >> 1); return r + 1; }
    *v = ((acc >> 2) & (rb - 1)) | rb;
    return r + 2;
}

/* middle swap for group values at rungs 1..7 */
static inline uint64_t qb3o_swap(uint64_t v, unsigned r) {
    const uint64_t top = 1ull << r;
    return (v == top || v == top - 1) ? v ^ (2 * top - 1) : v;
}

/* ---------- generated tables ---------- */
typedef struct {
    uint16_t eg[8][256];    /* group value code, swapped (rungs 1..7), entry = len<<12 | code */
    uint16_t ev[8][256];    /* single value code as used for cf/index values: rungs 1,2 unswapped, 3..7 swapped */
    uint16_t dg[8][512];    /* inverse of eg, indexed by the next r+2 bits: len<<12 | value */
    uint16_t dv[8][512];    /* inverse of ev */
    uint16_t csw[7][64];    /* rung switch code incl. change flag, [UBITS][delta mod 2^UBITS] */
    uint16_t dsw[7][128];   /* inverse, indexed by the UBITS+1 bits after the flag: (len incl flag)<<12 | delta; signal -> delta 0 */
    uint16_t signal[7];     /* the unused switch code, len<<12 | code */
    int ready;
} qb3o_tables;

extern qb3o_tables qb3o_tab;
void qb3o_tables_init(void);

static inline unsigned qb3o_topbit(uint64_t v) { return 63u - (unsigned)__builtin_clzll(v); }

#endif
