/* oracle/qb3o_gen.c -- synthetic raster generators + FNV-1a64 (test infrastructure, see qb3o_gen.h) */
#include "qb3o_gen.h"
#include <string.h>

uint64_t qb3o_splitmix64(uint64_t x) {
    uint64_t z = x + 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

uint64_t qb3o_fnv1a64(const void *buf, size_t n) {
    const uint8_t *p = (const uint8_t *)buf;
    uint64_t h = 0xcbf29ce484222325ull;
    for (size_t i = 0; i < n; i++) {
        h ^= p[i];
        h *= 0x100000001b3ull;
    }
    return h;
}

static uint64_t gen_value(int gen, uint64_t x, uint64_t y, uint64_t c, uint64_t r, int bits) {
    switch (gen) {
    case QB3O_GRAD:      return x + y + 17 * c;
    case QB3O_NOISY3:    return x + y + 17 * c + (r & 7);
    case QB3O_LANDSAT16: return 7000 + 3 * x + 2 * y + 301 * c + (r & 63);
    case QB3O_DEM:       return 37 * (x + y) - 50000 + (r & 63);
    case QB3O_TERRACE:   return 1000 * (x / 16 + y / 16 - 100);
    case QB3O_FEW:       return ((r % 6) << (bits - 6)) - (1ull << (bits - 4));
    case QB3O_PALETTE:   return (qb3o_splitmix64(77 + r % 5) >> (66 - bits)) | 1;
    case QB3O_RANDOM:    return r;
    case QB3O_RUNG63:    return qb3o_splitmix64(77 + r % 5);
    case QB3O_CONST:     return 42;
    default:             return 0;
    }
}

void qb3o_generate(void *dst, size_t w, size_t h, size_t bands, int tsize, int gen, uint64_t seed) {
    uint8_t *d = (uint8_t *)dst;
    const int bits = 8 * tsize;
    size_t idx = 0;
    for (size_t y = 0; y < h; y++)
        for (size_t x = 0; x < w; x++)
            for (size_t c = 0; c < bands; c++, idx++) {
                uint64_t r = qb3o_splitmix64(seed + idx);
                uint64_t v = gen_value(gen, x, y, c, r, bits);
                memcpy(d + idx * (size_t)tsize, &v, (size_t)tsize); /* little endian truncate */
            }
}
