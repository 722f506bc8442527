/*
 * oracle/qb3o_cli.c -- run one synthetic case through the CPU restatement and print its anchors
 * (test infrastructure).  Usage:
 *   qb3o_cli W H BANDS DTYPE MODE GEN SEED [cb=explicit|cb=identity] [q=N] [away] [reps=N]
 * prints: fnv(input) stream_bytes fnv(stream) header_mode roundtrip(ok|MISMATCH|fail) enc_MPix/s dec_MPix/s
 */
#include "qb3o.h"
#include "qb3o_gen.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int main(int argc, char **argv) {
    if (argc < 8) { fprintf(stderr, "usage: %s W H BANDS DTYPE MODE GEN SEED [cb=explicit] [q=N] [away] [reps=N]\n", argv[0]); return 2; }
    size_t w = strtoull(argv[1], 0, 0), h = strtoull(argv[2], 0, 0), b = strtoull(argv[3], 0, 0);
    int dt = atoi(argv[4]), mode = atoi(argv[5]), gen = atoi(argv[6]);
    uint64_t seed = strtoull(argv[7], 0, 0), q = 1;
    int explicit_cb = 0, identity = 0, away = 0, reps = 1;
    for (int i = 8; i < argc; i++) {
        if (!strcmp(argv[i], "cb=explicit")) explicit_cb = 1;
        else if (!strcmp(argv[i], "cb=identity")) identity = 1;
        else if (!strncmp(argv[i], "q=", 2)) q = strtoull(argv[i] + 2, 0, 0);
        else if (!strcmp(argv[i], "away")) away = 1;
        else if (!strncmp(argv[i], "reps=", 5)) reps = atoi(argv[i] + 5);
    }
    int tsz = qb3o_typesize(dt);
    size_t raw = w * h * b * (size_t)tsz;
    uint8_t *img = malloc(raw), *out = malloc(raw);
    qb3o_generate(img, w, h, b, tsz, gen, seed);
    qb3o_encoder e;
    if (!qb3o_encoder_init(&e, w, h, b, dt)) { fprintf(stderr, "bad parameters\n"); return 2; }
    qb3o_set_mode(&e, mode);
    if (explicit_cb) {      /* the map SURVEY Appendix C uses for the 8-band rows: {1,1,1,3,4,...} */
        size_t cb[QB3O_MAXBANDS];
        for (size_t c = 0; c < b; c++) cb[c] = c;
        if (b >= 3) cb[0] = cb[2] = 1;
        qb3o_set_coreband(&e, b, cb);
    }
    if (q > 1) qb3o_set_quanta(&e, q, away);
    uint8_t *dst = malloc(qb3o_max_encoded_size(&e));
    size_t n = 0;
    double t0 = now();
    for (int r = 0; r < reps; r++) { qb3o_encoder_reset(&e); qb3o_set_mode(&e, mode); n = qb3o_encode(&e, img, dst); }
    double tenc = (now() - t0) / reps;
    if (!n) { printf("encode failed, error %d\n", e.error); return 1; }
    size_t dims[3];
    qb3o_decoder d;
    const char *rt = "fail";
    double tdec = 0;
    if (qb3o_read_start(&d, dst, n, dims)) {
        d.identity_cband = identity;
        if (qb3o_read_info(&d)) {
            t0 = now();
            size_t got = 0;
            for (int r = 0; r < reps; r++) { qb3o_decoder d2 = d; got = qb3o_read_data(&d2, out); }
            tdec = (now() - t0) / reps;
            if (got == raw) rt = memcmp(img, out, raw) ? "MISMATCH" : "ok";
        }
    }
    printf("%016llx %zu %016llx %d %s %.1f %.1f\n", (unsigned long long)qb3o_fnv1a64(img, raw), n,
           (unsigned long long)qb3o_fnv1a64(dst, n), dst[10], rt, 1e-6 * w * h / tenc, tdec > 0 ? 1e-6 * w * h / tdec : 0.0);
    if (strcmp(rt, "ok")) printf("decoded fnv %016llx\n", (unsigned long long)qb3o_fnv1a64(out, raw));
    return 0;
}
