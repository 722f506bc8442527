/* oracle/qb3o_tables.c -- code tables generated from the rules in qb3o_bits.h (test infrastructure) */
#include "qb3o_bits.h"

qb3o_tables qb3o_tab;

void qb3o_tables_init(void) {
    qb3o_tables *t = &qb3o_tab;
    if (t->ready) return;
    uint64_t lo; unsigned hi;
    for (unsigned r = 0; r < 8; r++) {
        const unsigned nv = 2u << r;           /* values 0 .. 2^(r+1)-1 */
        const unsigned nx = 4u << r;           /* r+2 index bits */
        if (r == 0) {
            for (unsigned v = 0; v < 2; v++) t->ev[0][v] = t->eg[0][v] = (uint16_t)(0x1000 | v);
            for (unsigned x = 0; x < nx; x++) t->dv[0][x] = t->dg[0][x] = (uint16_t)(0x1000 | (x & 1));
            continue;
        }
        for (unsigned v = 0; v < nv; v++) {
            unsigned len = qb3o_code(qb3o_swap(v, r), r, &lo, &hi);
            t->eg[r][v] = (uint16_t)((len << 12) | lo);
            if (r < 3) {
                len = qb3o_code(v, r, &lo, &hi);
                t->ev[r][v] = (uint16_t)((len << 12) | lo);
            } else
                t->ev[r][v] = t->eg[r][v];
        }
        for (unsigned x = 0; x < nx; x++) {
            uint64_t v;
            unsigned len = qb3o_decode_code(x, r, &v);
            t->dg[r][x] = (uint16_t)((len << 12) | qb3o_swap(v, r));
            t->dv[r][x] = (r < 3) ? (uint16_t)((len << 12) | v) : t->dg[r][x];
        }
    }
    for (unsigned u = 3; u <= 6; u++) {
        const unsigned n = 1u << u;
        t->csw[u][0] = 0x1000;
        for (unsigned d = 1; d < n; d++) {
            unsigned m = (d < n / 2) ? 2 * (d - 1) : 2 * (n - d) - 1;
            unsigned len = qb3o_code(m, u - 1, &lo, &hi);
            t->csw[u][d] = (uint16_t)(((len + 1) << 12) | (lo << 1) | 1);
        }
        unsigned len = qb3o_code(n - 2, u - 1, &lo, &hi);
        t->signal[u] = (uint16_t)(((len + 1) << 12) | (lo << 1) | 1);
        for (unsigned x = 0; x < 2 * n; x++) {
            uint64_t m;
            len = qb3o_decode_code(x, u - 1, &m);
            unsigned d = (m == n - 2) ? 0 : (m & 1) ? (unsigned)(n - (m + 1) / 2) % n : (unsigned)(m / 2 + 1);
            t->dsw[u][x] = (uint16_t)(((len + 1) << 12) | d);
        }
    }
    t->ready = 1;
}
