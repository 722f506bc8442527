/*
 * oracle/qb3o_gen.h -- deterministic synthetic rasters + FNV-1a64, TEST INFRASTRUCTURE ONLY.
 *
 * These are the generators SURVEY.md section 8(d) / Appendix C define; the Appendix C anchor table
 * (stream size + FNV-1a64, minted from the reference by the surveyor) is quoted against exactly
 * these inputs, so the oracle restatement is pinned by reproducing those anchors.
 *
 *   r   = splitmix64(seed + idx),  idx = (y*W + x)*bands + c
 *   GRAD      = x + y + 17c
 *   NOISY3    = GRAD + (r & 7)
 *   LANDSAT16 = 7000 + 3x + 2y + 301c + (r & 63)
 *   DEM       = 37(x+y) - 50000 + (r & 63)
 *   TERRACE   = 1000 * (x/16 + y/16 - 100)
 *   FEW       = ((r mod 6) << (bits-6)) - (1 << (bits-4))
 *   PALETTE   = (splitmix64(77 + r mod 5) >> (66-bits)) | 1
 *   RUNG63    = splitmix64(77 + (r mod 5))               (u64 only)
 *   RANDOM    = r
 * all truncated to the width of T.
 */
#ifndef QB3O_GEN_H
#define QB3O_GEN_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum qb3o_gen {
    QB3O_GRAD = 0, QB3O_NOISY3, QB3O_LANDSAT16, QB3O_DEM, QB3O_TERRACE,
    QB3O_FEW, QB3O_PALETTE, QB3O_RANDOM, QB3O_RUNG63, QB3O_CONST, QB3O_GEN_END
};

uint64_t qb3o_splitmix64(uint64_t x);
uint64_t qb3o_fnv1a64(const void *buf, size_t n);

/* Fill dst (w*h*bands values of `tsize` bytes each, band interleaved, y-major). */
void qb3o_generate(void *dst, size_t w, size_t h, size_t bands, int tsize,
                   int gen, uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif
