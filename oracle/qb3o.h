/*
 * oracle/qb3o.h -- CPU restatement of the QB3 codec hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or call this.
 * The product (qb3_amd/, libqb3_amd.so) never includes, links or falls back to anything here.
 *
 * What it restates (file:line relative to /root/reference/QB3lib):
 *   encode  QB3encode.h:155-280 (group codes), :283-361 (common factor), :376-451 (FTL/BASE driver),
 *           :557-613 (index coding), :617-724 (BEST driver); QB3encode.cpp:26-134 (handle + setters),
 *           :137-186 (quantize), :189-268 (headers), :271-332 (RLE0), :461-574 (stored + top level)
 *   decode  QB3decode.h:119-290 (group decode), :293-570 (FTL driver), :578-741 (BASE/BEST driver);
 *           QB3decode.cpp:77-107 (dequantize), :130-264 (header parse), :267-307 (deRLE0), :356-452
 *   bits    bitstream.h:25-126 (LSB-first bit I/O), QB3common.h:127-166 (mag-sign, step)
 *
 * Pinning: the reference holds no golden vectors.  oracle/_ref cannot be built
 * under this project's rules (QB3common.h:20 includes a CMake-generated export header; the reference
 * build system may not be run and generated code may not be stood in for).  The restatement is pinned
 * by SURVEY.md Appendix C: stream byte count + FNV-1a64 of ~70 streams minted from the reference,
 * on inputs defined by qb3o_gen.h.  tests/test_oracle_anchors.py checks them.
 */
#ifndef QB3O_H
#define QB3O_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QB3O_MAXBANDS 16

/* numeric values follow QB3.h:40 and QB3.h:50-74 */
enum { QB3O_U8 = 0, QB3O_I8, QB3O_U16, QB3O_I16, QB3O_U32, QB3O_I32, QB3O_U64, QB3O_I64 };
enum { QB3O_BASE_Z = 0, QB3O_CF_Z, QB3O_RLE_Z, QB3O_CF_RLE_Z, QB3O_BASE_H, QB3O_CF_H, QB3O_RLE_H,
       QB3O_CF_RLE_H, QB3O_FTL, QB3O_MODE_END, QB3O_STORED = 255 };

#define QB3O_ZCURVE  0x0145236789cdabefull
#define QB3O_HILBERT 0x01548cd9aefb7623ull

typedef struct { uint64_t prev, runbits, cf; } qb3o_band_state;

typedef struct {
    size_t xsize, ysize, nbands;
    size_t stride;              /* line stride in values, 0 = xsize*nbands */
    uint64_t order;             /* 0 = Hilbert, else curve nibbles */
    uint64_t quanta;
    qb3o_band_state band[QB3O_MAXBANDS];
    size_t cband[QB3O_MAXBANDS];
    int error;
    int mode;
    int type;
    int away;
    int fix_b2;                 /* 0 = reproduce reference defect B-2 (u64 index sentinel), 1 = correct */
} qb3o_encoder;

typedef struct {
    size_t xsize, ysize, nbands;
    size_t stride;
    uint64_t order;
    uint64_t quanta;
    int error, stage;
    uint8_t cband[QB3O_MAXBANDS];
    int mode, type;
    const uint8_t *s_in;
    size_t s_size;
    int identity_cband;         /* 0 = reference behaviour (defect B-1: zero-filled map), 1 = identity */
} qb3o_decoder;

/* Encoder handle semantics as QB3encode.cpp:26-134 */
int    qb3o_encoder_init(qb3o_encoder *p, size_t w, size_t h, size_t bands, int dtype);
void   qb3o_encoder_reset(qb3o_encoder *p);
int    qb3o_set_coreband(qb3o_encoder *p, size_t bands, size_t *cband);
int    qb3o_set_quanta(qb3o_encoder *p, uint64_t q, int away);
int    qb3o_set_mode(qb3o_encoder *p, int mode);
size_t qb3o_max_encoded_size(const qb3o_encoder *p);
size_t qb3o_encode(qb3o_encoder *p, const void *src, void *dst);

/* Decoder: read_start / read_info / read_data as QB3decode.cpp:130-264,380-464 */
int    qb3o_read_start(qb3o_decoder *p, const void *src, size_t n, size_t *dims3);
int    qb3o_read_info(qb3o_decoder *p);
size_t qb3o_decoded_size(const qb3o_decoder *p);
size_t qb3o_read_data(qb3o_decoder *p, void *dst);

/* Raw block stream (no container): what the device kernels must reproduce bit for bit.
 * Encodes with the handle's mode/order/cband/band state; returns the number of BITS written,
 * dst is zero padded to a byte. */
uint64_t qb3o_encode_raw(qb3o_encoder *p, const void *src, void *dst);
/* Decodes a raw block stream of len bytes; returns 0 on success, nonzero on failure. */
int    qb3o_decode_raw(const qb3o_decoder *p, const uint8_t *src, size_t len, void *dst);

/* RLE0 byte post-pass (QB3encode.cpp:271-332, QB3decode.cpp:267-307) */
size_t  qb3o_rle0(const uint8_t *src, size_t len, uint8_t *dst);
size_t  qb3o_rle0_size(const uint8_t *src, size_t len);
int64_t qb3o_derle0(const uint8_t *src, size_t slen, uint8_t *dst, size_t dlen);
size_t  qb3o_derle0_size(const uint8_t *src, size_t len);

int qb3o_typesize(int dtype);

#ifdef __cplusplus
}
#endif
#endif
