/*
 * include/QB3.h -- C ABI of the MI355X-native QB3 codec (libQB3.so built from qb3_amd/csrc).
 *
 * Drop-in boundary: these 21 entry points have the names, argument meaning, return conventions and
 * enum values of the reference library's public header (reference QB3lib/QB3.h:36-162), so a caller
 * linked against the reference libQB3 (cqb3.cpp:405-493,276-323; test_qb3.cpp:84-142; GDAL MRF) can
 * be relinked against this library unchanged.  Each declaration cites the reference definition it
 * replaces.  All pointers are HOST pointers; device-pointer variants live in qb3x.h.
 *
 * The block coding itself runs on the GPU (HIP, gfx950).  There is no CPU fallback: if no HIP device
 * is usable, qb3_encode / qb3_read_data return 0 and the handle's error is QB3E_LIBERR.
 */
#ifndef QB3_AMD_QB3_H
#define QB3_AMD_QB3_H
#include <stddef.h>
#include <stdint.h>
#if !defined(__cplusplus)
#include <stdbool.h>
#endif

#if defined(__cplusplus)
extern "C" {
#endif

#define QB3_MAXBANDS 16         /* reference QB3.h:34 */
#define QB3_HAS_FTL 1           /* reference QB3.h:43 */

typedef struct encs *encsp;     /* opaque encoder handle, reference QB3.h:36 */
typedef struct decs *decsp;     /* opaque decoder handle, reference QB3.h:37 */

/* reference QB3.h:40 */
typedef enum qb3_dtype { QB3_U8 = 0, QB3_I8, QB3_U16, QB3_I16, QB3_U32, QB3_I32, QB3_U64, QB3_I64 } qb3_dtype;

/* reference QB3.h:50-74 */
typedef enum qb3_mode {
    QB3M_BASE_Z = 0, QB3M_CF = 1, QB3M_RLE = 2, QB3M_CF_RLE = 3,          /* legacy Z-curve modes */
    QB3M_BASE_H = 4, QB3M_CF_H = 5, QB3M_RLE_H = 6, QB3M_CF_RLE_H = 7,    /* Hilbert-curve modes */
    QB3M_FTL = 8, QB3M_END,
    QB3M_DEFAULT = 8, QB3M_BASE = 4, QB3M_BEST = 7,                        /* aliases */
    QB3M_STORED = 255, QB3M_INVALID = -1
} qb3_mode;

/* reference QB3.h:77-83 */
typedef enum qb3_error { QB3E_OK = 0, QB3E_EINV, QB3E_UNKN, QB3E_ERR, QB3E_LIBERR = 255 } qb3_error;

/* ---- encoder ---- */
encsp    qb3_create_encoder(size_t width, size_t height, size_t bands, qb3_dtype dt);  /* QB3encode.cpp:26 */
void     qb3_destroy_encoder(encsp p);                                                 /* QB3encode.cpp:59 */
void     qb3_reset_encoder(encsp p);                                                   /* QB3encode.cpp:50 */
bool     qb3_set_encoder_coreband(encsp p, size_t bands, size_t *cband);               /* QB3encode.cpp:63 */
bool     qb3_set_encoder_quanta(encsp p, uint64_t q, bool away);                       /* QB3encode.cpp:87 */
size_t   qb3_max_encoded_size(const encsp p);                                          /* QB3encode.cpp:112 */
qb3_mode qb3_set_encoder_mode(encsp p, qb3_mode mode);                                 /* QB3encode.cpp:120 */
void     qb3_set_encoder_stride(encsp p, size_t stride);                               /* QB3encode.cpp:79 */
size_t   qb3_encode(encsp p, void *source, void *destination);                         /* QB3encode.cpp:488 */
int      qb3_get_encoder_state(encsp p);                                               /* QB3encode.cpp:338 */

/* ---- decoder ---- */
decsp    qb3_read_start(void *source, size_t source_size, size_t *image_size);         /* QB3decode.cpp:130 */
bool     qb3_read_info(decsp p);                                                       /* QB3decode.cpp:176 */
size_t   qb3_read_data(decsp p, void *destination);                                    /* QB3decode.cpp:455 */
void     qb3_destroy_decoder(decsp p);                                                 /* QB3decode.cpp:36 */
size_t   qb3_decoded_size(const decsp p);                                              /* QB3decode.cpp:40 */
qb3_dtype qb3_get_type(const decsp p);                                                 /* QB3decode.cpp:44 */
void     qb3_set_decoder_stride(decsp p, size_t stride);                               /* QB3decode.cpp:72 */
qb3_mode qb3_get_mode(const decsp p);                                                  /* QB3decode.cpp:48 */
uint64_t qb3_get_quanta(const decsp p);                                                /* QB3decode.cpp:52 */
uint64_t qb3_get_order(const decsp p);                                                 /* QB3decode.cpp:56 */
bool     qb3_get_coreband(const decsp p, size_t *cband);                               /* QB3decode.cpp:63 */

#if defined(__cplusplus)
}
#endif
#endif
