"""Differential sweep: the HIP path against the CPU oracle on randomly drawn shapes, types, modes, band maps, strides
and generators (fixed seeds, so a failure names its case).  The shapes are biased towards what selects different
kernels: widths that are / are not multiples of 4, band counts that do / do not split into groups, rungs above and
below 8, images of one block, one chunk, several chunks."""
import os
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GENS = ["GRAD", "NOISY3", "LANDSAT16", "DEM", "TERRACE", "FEW", "PALETTE", "RANDOM", "CONST"]
WIDTHS = [4, 8, 12, 36, 64, 100, 128, 256, 260, 509, 512, 1024, 1028]
HEIGHTS = [4, 5, 7, 8, 19, 32, 37, 64, 67, 128]


def draw(rng):
    dt = rng.choice([0, 0, 0, 1, 2, 2, 3, 4, 5, 6, 7])
    b = rng.choice([1, 1, 2, 3, 3, 3, 4, 4, 5, 6, 7, 8, 8, 10, 12, 16])
    w = rng.choice(WIDTHS)
    h = rng.choice(HEIGHTS)
    while w * h * b > 600_000:
        w = max(4, w // 2)
    gen = rng.choice(GENS)
    mode = rng.choice([8, 8, 4, 4, 0, 5, 1, 7, 3, 6, 2])
    cb = None
    r = rng.random()
    if b >= 3 and r < 0.35:
        cb = [1, 1, 1] + list(range(3, b))
    elif r < 0.5:
        cb = list(range(b))
    elif r < 0.6:
        cb = [rng.randrange(b) for _ in range(b)]
    stride = 0
    if rng.random() < 0.15:
        stride = w * b + rng.choice([1, 2, 3, 4, 8])
    return w, h, b, dt, gen, mode, cb, stride


@pytest.mark.parametrize("block", range(24))
def test_random_cases_match_the_oracle(qb3, oracle, block):
    rng = random.Random(int(os.environ.get("QB3_FUZZ_SEED", "20260")) + block)      # (QB3_FUZZ_SEED: another draw of the same sweep)
    for k in range(60):
        w, h, b, dt, gen, mode, cb, stride = draw(rng)
        if dt >= 6 and mode in (1, 3, 5, 7) and gen in ("PALETTE", "RANDOM", "FEW"):
            gen = "DEM"                 # 64-bit units over 800 bits trip reference defect B-2: not a parity case
        seed = rng.randrange(1 << 20)
        img = oracle.generate(w, h, b, dt, gen, seed)
        tag = f"case {block}.{k}: {w}x{h}x{b} type {dt} {gen} seed {seed} mode {mode} cband {cb} stride {stride}"
        q, away = 1, False
        if dt <= 5 and rng.random() < 0.15:                  # lossy: quantised on the way in, scaled back on the way out
            q, away = rng.choice([2, 3, 4, 5, 10]), rng.random() < 0.5
            tag += f" quanta {'+' if away else ''}{q}"
        if stride:                      # rows `stride` values apart, junk-free padding (with or without quanta: reference test_qb3.cpp:659-660)
            src = np.zeros((h, stride), dtype=img.dtype)
            src[:, :w * b] = img.reshape(h, w * b)
            e = oracle.Encoder(w, h, b, dt)
            e.set_mode(mode)
            if cb is not None:
                e.set_coreband(cb)
            e.set_stride(stride)
            if q > 1:
                e.set_quanta(q, away)
            ref = e.encode(src)
            got = _encode_strided(qb3, src, w, h, b, dt, mode, cb, stride, quanta=q, away=away)
        else:
            src = img
            ref = oracle.encode(img, dt, mode, cband=cb, quanta=q, away=away)
            got = qb3.encode(img, dt, mode, cband=cb, quanta=q, away=away)
        assert len(got) == len(ref) and np.array_equal(got, ref), tag
        # decode what the reference would have written; identity map when the container has no CB chunk (B-1)
        want, _, _, _ = oracle.decode(ref, identity=True)
        if want is None:                # the reference refuses its own output here (RLE0 on tiny images)
            with pytest.raises(RuntimeError):
                qb3.decode(ref)
            continue
        out, dims, dtype, m = qb3.decode(ref)
        assert dims == (w, h, b) and np.array_equal(out, want), tag
        if stride:                      # ... and into a destination with the same line stride: the pixels land on the lines, nothing between them
            sout = _decode_strided(qb3, ref, h, stride, img.dtype)
            assert np.array_equal(sout[:, :w * b].reshape(-1).view(np.uint8), want) and (sout[:, w * b:].view(np.uint8) == 0xa5).all(), tag + " (strided decode)"
        if rng.random() < 0.3 and mode in (8, 4, 0, 5, 1):      # and through the self-indexing container
            level = 1 + (k & 1)         # the restart table, with block lengths every other time (where the raster takes them)
            s2 = _encode_strided(qb3, src, w, h, b, dt, mode, cb, stride, chunk=level, quanta=q, away=away) if stride else \
                qb3.encode(img, dt, mode, cband=cb, quanta=q, away=away, index_chunk=level)
            out2, _, _, _ = qb3.decode(s2)
            assert np.array_equal(out2, want), tag + " (index chunk, level %d)" % level


def _decode_strided(qb3, stream, h, stride, npdtype):
    """qb3_read_data into lines `stride` values apart (qb3_set_decoder_stride); the gaps keep their fill"""
    import ctypes as C
    L = qb3.lib
    dims = (C.c_size_t * 3)()
    buf = np.ascontiguousarray(stream)
    d = L.qb3_read_start(buf.ctypes.data, buf.size, dims)
    assert d and L.qb3_read_info(d)
    try:
        L.qb3_set_decoder_stride(d, stride)
        out = np.full(h * stride * np.dtype(npdtype).itemsize, 0xa5, np.uint8)
        assert L.qb3_read_data(d, out.ctypes.data), "strided qb3_read_data failed"
        return out.view(npdtype).reshape(h, stride)
    finally:
        L.qb3_destroy_decoder(d)


def _encode_strided(qb3, buf, w, h, b, dt, mode, cb, stride, chunk=False, quanta=1, away=False):
    """qb3.encode() takes (h, w, bands) arrays; strided input goes through the C API directly"""
    import ctypes as C
    L = qb3.lib
    p = L.qb3_create_encoder(w, h, b, dt)
    try:
        L.qb3_set_encoder_mode(p, mode)
        if chunk:
            L.qb3x_set_encoder_index_chunk(p, int(chunk))
        if cb is not None:
            arr = (C.c_size_t * b)(*cb)
            L.qb3_set_encoder_coreband(p, b, arr)
        L.qb3_set_encoder_stride(p, stride)
        if quanta > 1:
            L.qb3_set_encoder_quanta(p, quanta, away)
        dst = np.empty(L.qb3_max_encoded_size(p), dtype=np.uint8)
        src = np.ascontiguousarray(buf)
        n = L.qb3_encode(p, src.ctypes.data, dst.ctypes.data)
        assert n, "qb3_encode failed: state %d" % L.qb3_get_encoder_state(p)
        return dst[:n].copy()
    finally:
        L.qb3_destroy_encoder(p)


@pytest.mark.parametrize("case", [(64, 48, 3, 0, "NOISY3", 8), (128, 36, 1, 0, "GRAD", 4), (96, 64, 8, 2, "LANDSAT16", 4), (64, 64, 1, 5, "DEM", 8),
                                  (64, 64, 1, 7, "DEM", 5), (64, 48, 3, 0, "PALETTE", 5), (40, 44, 4, 0, "NOISY3", 0), (61, 35, 3, 0, "NOISY3", 8)],
                         ids=lambda c: "%dx%dx%d-t%d-%s-m%d" % c)
def test_damaged_streams_fail_or_decode_but_never_fault(qb3, oracle, case):
    """bit flips, byte smashes and truncations of valid containers: every decoder kernel bounds what it reads by the
    stream length and what it writes by the geometry, so the call returns -- pixels or an error -- and the process and
    the GPU stay healthy (checked by decoding the intact stream again afterwards)"""
    w, h, b, dt, gen, mode = case
    img = oracle.generate(w, h, b, dt, gen, 7)
    good = oracle.encode(img, dt, mode, cband=None if b in (1, 3, 4) else list(range(b)))
    chunked = qb3.encode(img, dt, mode if mode not in (2, 3, 6, 7) else 5, cband=None if b in (1, 3, 4) else list(range(b)), index_chunk=True)
    chunked2 = qb3.encode(img, dt, mode if mode not in (2, 3, 6, 7) else 5, cband=None if b in (1, 3, 4) else list(range(b)), index_chunk=2)
    rng = random.Random(99)
    data0 = bytes(good).index(b"DT", 11) + 2
    outcomes = {"ok": 0, "error": 0}
    for trial in range(40):
        base = good if trial % 3 else (chunked2 if trial % 6 else chunked)       # (the table with block lengths, where the raster takes them)
        s = base.copy()
        lo = data0 if base is good else 11          # the self-indexing containers: damage the restart table too
        kind = trial % 4
        if kind == 0:
            for _ in range(rng.randrange(1, 4)):
                at = rng.randrange(lo, len(s)); s[at] ^= 1 << rng.randrange(8)
        elif kind == 1:
            at = rng.randrange(lo, len(s)); n = min(len(s) - at, rng.randrange(1, 64)); s[at:at + n] = rng.randrange(256)
        elif kind == 2:
            s = s[:rng.randrange(lo + 1, len(s))].copy()
        else:
            s = np.concatenate([s, np.frombuffer(bytes(rng.randrange(256) for _ in range(rng.randrange(1, 40))), dtype=np.uint8)])
        try:
            out, dims, dtype, m = qb3.decode(s)
            assert out.size == w * h * b * oracle.TYPESIZE[dt]
            outcomes["ok"] += 1
        except (RuntimeError, ValueError):
            outcomes["error"] += 1
    out, _, _, _ = qb3.decode(good)
    want, _, _, _ = oracle.decode(good, identity=True)
    assert np.array_equal(out, want), outcomes
