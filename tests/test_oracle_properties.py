"""Size-independent properties of the oracle (CPU): round trips over modes x types x shapes, RLE0, quanta,
stride, narrow images, handle statefulness -- the matrix of the reference's test_qb3.cpp:643-723 restated on
synthetic inputs, plus the cases it never runs."""
import numpy as np
import pytest

MODES = list(range(9))
SHAPES = [(64, 48, 1), (37, 21, 3), (16, 16, 4), (40, 24, 2), (12, 20, 8), (4, 4, 1), (5, 4, 3)]


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("dtype", range(8))
@pytest.mark.parametrize("gen", ["NOISY3", "DEM", "TERRACE", "FEW", "PALETTE", "RANDOM", "CONST"])
def test_roundtrip_modes_types(oracle, mode, dtype, gen):
    if dtype >= 6 and mode in (1, 3, 5, 7) and gen in ("FEW", "RANDOM", "PALETTE"):
        pytest.skip("u64 common-factor modes drop units over 800 bits in the reference (defect B-2)")
    for (w, h, b) in SHAPES[:4]:
        img = oracle.generate(w, h, b, dtype, gen, 11)
        cb = None if b in (1, 3, 4) else [0] * b     # avoid decoder defect B-1
        s = oracle.encode(img, dtype, mode, cband=cb)
        out, dims, dt, _ = oracle.decode(s)
        assert out is not None and dims == (w, h, b) and dt == dtype
        assert np.array_equal(out, img.view(np.uint8).ravel()), (w, h, b)


@pytest.mark.parametrize("shape", SHAPES)
def test_roundtrip_shapes(oracle, shape):
    w, h, b = shape
    for dtype, gen in ((0, "NOISY3"), (3, "DEM"), (5, "FEW"), (6, "RUNG63")):
        for mode in (8, 4, 0):
            img = oracle.generate(w, h, b, dtype, gen, 5)
            s = oracle.encode(img, dtype, mode, cband=None if b in (1, 3, 4) else [0] * b)
            out, dims, _, _ = oracle.decode(s)
            assert out is not None and np.array_equal(out, img.view(np.uint8).ravel())


def test_defect_b2_and_its_fix(oracle):
    """u64 + common-factor modes: the reference's index sentinel 800 beats units longer than 800 bits and drops
    them (QB3encode.h:564,705-708).  The oracle reproduces that by default and round-trips with fix_b2."""
    img = oracle.generate(64, 48, 1, 6, "PALETTE", 11)
    faithful = oracle.encode(img, 6, 1)
    out, _, _, _ = oracle.decode(faithful)
    assert out is None or not np.array_equal(out, img.view(np.uint8).ravel())
    e = oracle.Encoder(64, 48, 1, 6)
    e.set_mode(1)
    oracle.lib.qb3o_set_fix_b2(e.p, 1)
    fixed = e.encode(img)
    out, _, _, _ = oracle.decode(fixed)
    assert out is not None and np.array_equal(out, img.view(np.uint8).ravel())


def test_rle0_roundtrip_and_escapes(oracle):
    rng = np.random.default_rng(1)
    cases = [np.zeros(1000, np.uint8), np.full(100, 0xff, np.uint8), rng.integers(0, 256, 5000).astype(np.uint8),
             np.array([0, 0, 0, 0], np.uint8), np.array([0xff, 0xff], np.uint8), np.array([1, 2], np.uint8), np.zeros(0, np.uint8)]
    sparse = rng.integers(0, 256, 4000).astype(np.uint8)
    sparse[rng.random(4000) < 0.8] = 0
    sparse[::97] = 0xff
    sparse[1::97] = 0xff
    cases.append(sparse)
    for src in cases:
        n = oracle.lib.qb3o_rle0_size(src.ctypes.data, src.size)
        dst = np.zeros(max(n, 1) + 8, np.uint8)
        assert oracle.lib.qb3o_rle0(src.ctypes.data, src.size, dst.ctypes.data) == n
        assert oracle.lib.qb3o_derle0_size(dst.ctypes.data, n) == src.size
        back = np.zeros(max(src.size, 1), np.uint8)
        assert oracle.lib.qb3o_derle0(dst.ctypes.data, n, back.ctypes.data, src.size) == 0
        assert np.array_equal(back[:src.size], src)


@pytest.mark.parametrize("dtype", [0, 1, 2, 3, 5, 7])
@pytest.mark.parametrize("q,away", [(2, False), (2, True), (3, False), (4, False), (4, True), (10, False), (10, True)])
def test_quanta_within_half_step(oracle, dtype, q, away):
    """reference test_qb3.cpp:148-161 accepts an error of at most q/2 after quantised round trip"""
    img = oracle.generate(40, 28, 3, dtype, "NOISY3" if dtype < 2 else "DEM", 9)
    s = oracle.encode(img, dtype, 4, quanta=q, away=away)
    out, _, _, _ = oracle.decode(s)
    assert out is not None
    dec = out.view(oracle.NPTYPE[dtype]).reshape(img.shape).astype(np.int64)    # i64 values here stay far from the limits
    src = img.astype(np.int64)
    info = np.iinfo(oracle.NPTYPE[dtype])
    # values whose quantised product would leave the type saturate; compare the rest
    ok = (np.abs(dec - src) <= q // 2 + (q % 2)) | (dec == info.max) | (dec == info.min)
    assert ok.all()


def test_stride_encode_decode(oracle):
    w, h, b = 30, 18, 3
    stride = w * b + 7
    canvas = np.zeros((h, stride), np.uint8)
    img = oracle.generate(w, h, b, 0, "NOISY3", 2)
    canvas[:, :w * b] = img.reshape(h, w * b)
    e = oracle.Encoder(w, h, b, 0)
    e.set_stride(stride)
    s = e.encode(canvas)
    assert np.array_equal(s, oracle.encode(img, 0))             # stride does not change the stream
    out, _, _, _ = oracle.decode(s, stride=stride)
    assert np.array_equal(out.reshape(h, stride)[:, :w * b], img.reshape(h, w * b))


@pytest.mark.parametrize("shape", [(2, 40, 3), (40, 3, 1), (1, 17, 2), (300, 1, 3), (3, 3, 1), (2, 8, 1)])
def test_narrow_images(oracle, shape):
    """narrow images are remapped to 4-wide (or 4-high) stand-ins (reference QB3encode.cpp:351-389, intent)"""
    w, h, b = shape
    img = oracle.generate(w, h, b, 0, "NOISY3", 3)
    s = oracle.encode(img, 0, 8, cband=None if b in (1, 3, 4) else [0] * b)
    out, dims, _, _ = oracle.decode(s)
    assert dims == (w, h, b) and np.array_equal(out, img.ravel())


def test_encoder_state_carries_until_reset(oracle):
    """reference defect/feature B-4: band state persists across qb3_encode calls (QB3encode.h:446-449)"""
    img = oracle.generate(32, 32, 3, 0, "NOISY3", 1)
    e = oracle.Encoder(32, 32, 3, 0)
    a = e.encode(img)
    b = e.encode(img)
    assert not np.array_equal(a, b)
    e.reset()
    assert np.array_equal(e.encode(img), a)
    assert e.band_state()[1][1] > 0       # rung of band 1 after the image


def test_mode_setter_semantics(oracle):
    e = oracle.Encoder(16, 16, 1, 0)
    assert e.set_mode(8) == 8 and e.set_mode(9) == 8 and e.set_mode(-1) == 8      # out of range: unchanged
    assert e.set_mode(2) == 2
    img = oracle.generate(16, 16, 1, 0, "NOISY3", 1)
    e.set_mode(4)                           # Z order is sticky once a Z mode was chosen (QB3encode.cpp:124-132)
    s = e.encode(img)
    assert b"SC" not in bytes(s[:40])
