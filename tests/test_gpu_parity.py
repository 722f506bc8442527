"""GPU parity: the HIP path, called through the C ABI (qb3_encode / qb3_read_*), against the CPU oracle.

Bit-exact is the bar: the encoded container must equal the oracle's byte for byte, and decoding the
oracle's (foreign, index-less) stream must reproduce the input exactly.
"""
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FTL, BASE, BASE_Z = 8, 4, 0


def first_diff(a, b):
    n = min(len(a), len(b))
    d = np.nonzero(a[:n] != b[:n])[0]
    return int(d[0]) if len(d) else n


def check_encode(qb3, oracle, img, dtype, mode, **kw):
    ref = oracle.encode(img, dtype, mode, **kw)
    got = qb3.encode(img, dtype, mode, **kw)
    assert len(got) == len(ref) and np.array_equal(got, ref), \
        f"stream differs: len {len(got)} vs {len(ref)}, first diff at byte {first_diff(got, ref)}"
    return ref


CASES = [
    # w, h, bands, dtype, gen, seed
    (64, 64, 3, 0, "NOISY3", 1),
    (512, 512, 3, 0, "GRAD", 0),
    (512, 512, 3, 0, "NOISY3", 1),
    (509, 515, 3, 0, "NOISY3", 1),        # shifted edge blocks
    (37, 21, 3, 0, "NOISY3", 5),
    (4, 4, 3, 0, "NOISY3", 5),            # a single block
    (1024, 16, 1, 0, "NOISY3", 7),
    (128, 128, 4, 0, "NOISY3", 9),
    (96, 80, 2, 0, "RANDOM", 9),
    (256, 256, 8, 2, "LANDSAT16", 3),
    (61, 67, 5, 3, "DEM", 4),
    (256, 256, 1, 5, "DEM", 4),
    (256, 256, 1, 7, "DEM", 4),
    (128, 128, 1, 5, "TERRACE", 4),
    (128, 128, 1, 7, "FEW", 4),
    (64, 64, 1, 6, "RUNG63", 4),          # 65-bit codes
    # 32/64-bit rasters of one band take the lane-per-block kernels (k_enc_pxw.hip, k_dec_pxw.hip): shifted last column and
    # row, several chunks, rungs below 8 (the piece builder) and far above, unsigned and signed
    (509, 131, 1, 5, "DEM", 6),
    (67, 61, 1, 7, "TERRACE", 4),
    (1030, 12, 1, 4, "RANDOM", 8),
    (1024, 260, 1, 6, "NOISY3", 9),
    (260, 1024, 1, 4, "GRAD", 0),
    (128, 128, 1, 5, "CONST", 0),
    (64, 64, 16, 0, "NOISY3", 11),
    (32, 32, 16, 6, "RANDOM", 12),
    (256, 256, 3, 0, "CONST", 0),
    # 8-bit 1/3/4 bands take the lane-per-block kernels (any width: the odd sizes above read and write unaligned rows);
    # heights that need a shifted last row, more than one 255-block chunk, all three band counts
    (64, 37, 3, 0, "NOISY3", 3),
    (128, 50, 4, 0, "NOISY3", 4),
    (256, 19, 1, 0, "NOISY3", 5),
    (1024, 64, 3, 0, "NOISY3", 6),
    (2048, 8, 4, 0, "RANDOM", 7),
    (4096, 12, 1, 0, "GRAD", 0),
    # 16-bit with width % 4 == 0 and 1-4 bands or an even band count take the lane-per-(block, band group) kernels:
    # every group shape (1, 2, 3, 4 bands per lane; 1, 2, 3, 4, 8 lanes per block), rungs below and above 8,
    # signed data, shifted last rows, several chunks
    (256, 37, 1, 2, "LANDSAT16", 3),
    (512, 64, 2, 2, "LANDSAT16", 4),
    (256, 50, 3, 2, "LANDSAT16", 5),
    (256, 64, 4, 3, "DEM", 6),
    (128, 44, 6, 2, "NOISY3", 7),
    (512, 128, 8, 2, "LANDSAT16", 3),
    (64, 36, 12, 3, "DEM", 8),
    (128, 20, 16, 2, "RANDOM", 9),
    (1024, 16, 3, 2, "RANDOM", 10),
    (256, 256, 4, 2, "NOISY3", 11),
    (64, 64, 2, 3, "TERRACE", 4),
    (128, 64, 3, 2, "CONST", 0),
    (96, 32, 10, 2, "FEW", 5),
    (64, 20, 14, 2, "PALETTE", 6),
]


@pytest.mark.parametrize("mode", [FTL, BASE, BASE_Z])
@pytest.mark.parametrize("case", CASES, ids=lambda c: "%dx%dx%d-t%d-%s" % c[:5])
def test_encode_matches_oracle(qb3, oracle, case, mode):
    w, h, b, dt, gen, seed = case
    img = oracle.generate(w, h, b, dt, gen, seed)
    check_encode(qb3, oracle, img, dt, mode)


@pytest.mark.parametrize("mode", [FTL, BASE, BASE_Z])
@pytest.mark.parametrize("case", CASES, ids=lambda c: "%dx%dx%d-t%d-%s" % c[:5])
def test_decode_foreign_stream(qb3, oracle, case, mode):
    """Streams made by the oracle carry no index: exercises the serial boundary scan + parallel decode."""
    w, h, b, dt, gen, seed = case
    img = oracle.generate(w, h, b, dt, gen, seed)
    cb = None
    if b not in (1, 3, 4):      # the default identity map on such band counts trips reference defect B-1;
        cb = [1, 1, 1] + list(range(3, b)) if b >= 3 else [0] * b    # use an explicit map that round-trips
    stream = oracle.encode(img, dt, mode, cband=cb)
    out, dims, dtype, m = qb3.decode(stream)
    assert dims == (w, h, b) and dtype == dt
    assert np.array_equal(out, img.view(np.uint8).ravel())


# ---------------------------------------------------------------------------------------------------------
# device-resident API (qb3x_*): index, state, tiles -- and the rest of the container on the host API

def dev_roundtrip(qb3, oracle, torch, w, h, b, dt, gen, seed, mode, cband=None):
    from qb3_amd import synth, device as qdev
    img = synth.generate(w, h, b, dt, gen, seed)
    enc = qdev.DeviceEncoder(w, h, b, dt, mode=mode, cband=cband)
    dst, n, index = enc.encode(img)
    stream = dst[:n].cpu().numpy()
    dec = qdev.DeviceDecoder(stream[:min(n, 64)], n)
    raw = img.reshape(-1).view(torch.uint8)
    assert torch.equal(dec.decode(dst, index=index), raw), "indexed decode"
    assert torch.equal(dec.decode(dst, index=None), raw), "index-less decode"
    return img, stream


@pytest.mark.parametrize("mode", [FTL, BASE, BASE_Z])
@pytest.mark.parametrize("case", [(512, 512, 3, 0, "NOISY3", 1), (509, 515, 3, 0, "NOISY3", 1), (1000, 36, 1, 0, "NOISY3", 2),
                                  (256, 256, 8, 2, "LANDSAT16", 3), (300, 200, 1, 5, "DEM", 4), (300, 200, 1, 7, "DEM", 4),
                                  (64, 64, 1, 6, "RANDOM", 4), (96, 96, 16, 3, "DEM", 6), (40, 44, 5, 0, "NOISY3", 8)],
                         ids=lambda c: "%dx%dx%d-t%d-%s" % c[:5])
def test_device_api_with_index(qb3, oracle, case, mode):
    import torch
    w, h, b, dt, gen, seed = case
    cb = None if b in (1, 3, 4) else [1, 1, 1] + list(range(3, b))
    img, stream = dev_roundtrip(qb3, oracle, torch, w, h, b, dt, gen, seed, mode, cb)
    host = img.cpu().numpy().view(oracle.NPTYPE[dt])
    ref = oracle.encode(host, dt, mode, cband=cb)
    assert np.array_equal(stream, ref)


@pytest.mark.parametrize("mode", [FTL, BASE, BASE_Z])
@pytest.mark.parametrize("shape", [(320, 44, 3), (320, 44, 4), (512, 20, 3)])
def test_identity_band_map_on_rgb(qb3, oracle, shape, mode):
    """explicit identity map on 3/4 bands (the other variant of the lane-per-block kernels)"""
    import torch
    w, h, b = shape
    cb = list(range(b))
    img, stream = dev_roundtrip(qb3, oracle, torch, w, h, b, 0, "NOISY3", 21, mode, cb)
    ref = oracle.encode(img.cpu().numpy(), 0, mode, cband=cb)
    assert np.array_equal(stream, ref)


ANCHOR_ROWS = None


def anchors():
    global ANCHOR_ROWS
    if ANCHOR_ROWS is None:
        import json
        import os
        ANCHOR_ROWS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "anchors.json")))
    return ANCHOR_ROWS


def _anchor_ids():
    return ["cfg%s-%dx%dx%d-t%d-m%d-g%d%s" % (a["cfg"], a["w"], a["h"], a["bands"], a["dtype"], a["mode"], a["gen"],
                                               "-cb" if a["explicit_cb"] else "") for a in anchors()]


GEN_NAMES = ["GRAD", "NOISY3", "LANDSAT16", "DEM", "TERRACE", "FEW", "PALETTE", "RANDOM", "RUNG63"]


@pytest.mark.parametrize("a", anchors(), ids=_anchor_ids())
def test_full_size_anchor_on_device(qb3, oracle, a):
    """BASELINE.json's full-size configurations: the container made on the GPU must have the size and FNV-1a64
    the reference produced, and decode back to the input.  Modes the device encoder
    All nine modes are encoded on the device, common factor + index coding included."""
    import torch
    from qb3_amd import synth, device as qdev
    gen = GEN_NAMES[a["gen"]]
    if gen in ("FEW", "PALETTE", "RUNG63"):
        host = oracle.generate(a["w"], a["h"], a["bands"], a["dtype"], gen, a["seed"])
        img = torch.from_numpy(host.view(np.uint8)).cuda()
    else:
        img = synth.generate(a["w"], a["h"], a["bands"], a["dtype"], gen, a["seed"])
        host = None
    raw = img.reshape(-1).view(torch.uint8)
    cb = [1, 1, 1] + list(range(3, a["bands"])) if a["explicit_cb"] else None
    device_can_encode = True
    if device_can_encode:
        enc = qdev.DeviceEncoder(a["w"], a["h"], a["bands"], a["dtype"], mode=a["mode"], cband=cb)
        dst, n, index = enc.encode(img)
        stream = dst[:n].cpu().numpy()
        assert n == a["size"], "container size differs from the reference's"
        assert oracle.fnv(stream) == a["fnv_stream"], "container bytes differ from the reference's"
        assert stream[10] == a["hdr_mode"]
    else:
        if host is None:
            host = img.cpu().numpy()
        stream = oracle.encode(host.view(oracle.NPTYPE[a["dtype"]]).reshape(a["h"], a["w"], a["bands"]), a["dtype"], a["mode"], cband=cb)
        assert len(stream) == a["size"]
        n = len(stream)
        dst = torch.from_numpy(np.concatenate([stream, np.zeros((-n) % 4 + 8, np.uint8)])).cuda()
        index = None
    dec = qdev.DeviceDecoder(stream[:min(n, 64)], n)
    if not a["roundtrip"]:
        # reference defect B-1: compat flag reproduces the reference's (wrong) output, the default round-trips
        qb3.lib.qb3x_set_decoder_compat(dec.p, qb3.QB3X_REF_CBAND0)
        out = dec.decode(dst, index=index)
        assert oracle.fnv(out.cpu().numpy()) == a["ref_decoded_fnv"]
        qb3.lib.qb3x_set_decoder_compat(dec.p, 0)
    out = dec.decode(dst, index=index)
    assert torch.equal(out, raw), "decode(encode(x)) != x"


@pytest.mark.parametrize("mode", [1, 3, 5, 7, 2, 6])
@pytest.mark.parametrize("case", [(64, 48, 3, 0, "NOISY3", 1), (96, 64, 1, 5, "TERRACE", 4), (64, 64, 1, 5, "FEW", 4), (64, 64, 1, 7, "DEM", 4),
                                  (37, 29, 3, 3, "DEM", 2), (64, 64, 1, 2, "PALETTE", 7), (128, 128, 1, 0, "CONST", 0)],
                         ids=lambda c: "%dx%dx%d-t%d-%s" % c[:5])
def test_decode_common_factor_and_rle_streams(qb3, oracle, case, mode):
    """common-factor / index units and the RLE0 wrapper, decoded on the GPU from the oracle's streams"""
    w, h, b, dt, gen, seed = case
    img = oracle.generate(w, h, b, dt, gen, seed)
    stream = oracle.encode(img, dt, mode)
    out, dims, dtype, m = qb3.decode(stream)
    assert dims == (w, h, b) and dtype == dt and m == stream[10]
    assert np.array_equal(out, img.view(np.uint8).ravel())


@pytest.mark.parametrize("mode", [1, 3, 5, 7])
@pytest.mark.parametrize("case", CASES + [(64, 48, 3, 0, "PALETTE", 3), (96, 64, 1, 5, "TERRACE", 4), (64, 64, 1, 5, "FEW", 4), (64, 64, 1, 2, "PALETTE", 7),
                                          (64, 64, 3, 3, "TERRACE", 2), (48, 48, 1, 7, "TERRACE", 4), (48, 48, 1, 6, "PALETTE", 5), (33, 47, 2, 4, "FEW", 9)],
                         ids=lambda c: "%dx%dx%d-t%d-%s" % c[:5])
def test_encode_common_factor_modes(qb3, oracle, case, mode):
    """common factor + index coding on the device (reference encode_best, QB3encode.h:617-724).  64-bit data is
    compared with the oracle's corrected index sentinel (reference defect B-2 drops units over 800 bits)."""
    w, h, b, dt, gen, seed = case
    img = oracle.generate(w, h, b, dt, gen, seed)
    ref = oracle.encode(img, dt, mode, fix_b2=True)
    got = qb3.encode(img, dt, mode)
    assert len(got) == len(ref) and np.array_equal(got, ref), \
        f"stream differs: len {len(got)} vs {len(ref)}, first diff at byte {first_diff(got, ref)}"
    out, dims, _, _ = qb3.decode(got, compat=0)
    if b in (1, 3, 4):
        assert np.array_equal(out, img.view(np.uint8).ravel())


@pytest.mark.parametrize("mode", [2, 6])
def test_encode_rle_modes(qb3, oracle, mode):
    for (w, h, b, dt, gen) in ((256, 256, 1, 0, "CONST"), (128, 96, 1, 5, "TERRACE"), (64, 64, 3, 0, "NOISY3")):
        img = oracle.generate(w, h, b, dt, gen, 4)
        check_encode(qb3, oracle, img, dt, mode)


def _rle_patterns():
    """byte strings that exercise every branch of the reference's RLE0 loops: runs of 00 and ff of every length around the
    thresholds (4, 258, 2 x 258), ff before zeros, runs that reach the end, runs across the 4 KB chunks the device skips by"""
    rng = np.random.default_rng(11)
    pats = []
    for n in (0, 1, 2, 3, 4, 5, 6, 7):
        pats.append(np.zeros(n, np.uint8)); pats.append(np.full(n, 0xff, np.uint8))
    for L in (3, 4, 5, 257, 258, 259, 261, 262, 516, 517, 520, 4095, 4096, 4097, 8192 + 5, 3 * 4096 + 259):
        for lead in (b"", b"\x07", b"\xff", b"\xff\xff", b"\xff\xff\xff", b"\x00\x01"):
            for tail in (b"", b"\x09", b"\xff", b"\x00\x09", b"\x09\x09\x09"):
                pats.append(np.frombuffer(lead + bytes(L) + tail, np.uint8))
                pats.append(np.frombuffer(lead + b"\xff" * L + tail, np.uint8))
    for _ in range(40):             # random mixtures of short runs of 00, ff and other bytes
        parts = []
        for _ in range(int(rng.integers(1, 60))):
            kind = int(rng.integers(0, 4))
            k = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 9, 258, 259, 300]))
            parts.append(bytes(k) if kind == 0 else b"\xff" * k if kind == 1 else rng.integers(1, 255, k, dtype=np.uint8).tobytes() if kind == 2
                         else rng.choice([0, 0xff, 5], k).astype(np.uint8).tobytes())
        pats.append(np.frombuffer(b"".join(parts), np.uint8))
    big = rng.integers(0, 256, 300000, dtype=np.uint8)          # a larger one: mostly noise with planted runs
    for off, k, v in ((10, 5000, 0), (9000, 9000, 0xff), (100000, 258 * 40 + 3, 0), (250000, 49999, 0), (299990, 10, 0xff)):
        big[off:off + k] = v
    pats.append(big)
    return [p for p in pats if p.size]


def test_rle0_on_the_device_matches_the_reference_loops(qb3, oracle):
    """qb3x_rle0_device (k_rle0.hip) against the oracle's restatement of the reference's serial RLE0 coder and decoder:
    coded bytes identical, the expansion of the coded bytes identical to the input"""
    import ctypes as C
    import torch
    L = qb3.lib
    O = oracle.lib
    for pat in _rle_patterns():
        n = pat.size
        src = np.ascontiguousarray(pat)
        ref = np.zeros(n * 3 // 2 + 16, np.uint8)
        rsz = O.qb3o_rle0(src.ctypes.data, n, ref.ctypes.data)
        assert rsz == O.qb3o_rle0_size(src.ctypes.data, n)
        d_src = torch.from_numpy(src.copy()).cuda()
        assert L.qb3x_rle0_device(d_src.data_ptr(), n, None, 0, 0, None) == rsz, (n, bytes(src[:32]))
        d_dst = torch.zeros(rsz + 8, dtype=torch.uint8, device="cuda")
        assert L.qb3x_rle0_device(d_src.data_ptr(), n, d_dst.data_ptr(), rsz, 0, None) == rsz
        assert np.array_equal(d_dst[:rsz].cpu().numpy(), ref[:rsz]), (n, bytes(src[:32]))
        # and back: the expansion of the coded form, and of the RAW pattern read as coded bytes (any bytes are a valid input)
        for coded in (ref[:rsz].copy(), src):
            m = coded.size
            want = O.qb3o_derle0_size(coded.ctypes.data, m)
            exp = np.zeros(want + 1, np.uint8)
            assert O.qb3o_derle0(coded.ctypes.data, m, exp.ctypes.data, want) == 0
            d_c = torch.from_numpy(coded.copy()).cuda()
            assert L.qb3x_rle0_device(d_c.data_ptr(), m, None, 0, 1, None) == want
            if want:
                d_e = torch.zeros(want, dtype=torch.uint8, device="cuda")
                assert L.qb3x_rle0_device(d_c.data_ptr(), m, d_e.data_ptr(), want, 1, None) == want
                assert np.array_equal(d_e.cpu().numpy(), exp[:want]), (m, bytes(coded[:32]))


@pytest.mark.parametrize("dtype", [0, 1, 3, 5])
@pytest.mark.parametrize("q,away", [(2, False), (2, True), (3, False), (4, True), (10, False), (10, True)])
def test_quanta(qb3, oracle, dtype, q, away):
    img = oracle.generate(60, 44, 3, dtype, "NOISY3" if dtype < 2 else "DEM", 9)
    ref = check_encode(qb3, oracle, img, dtype, 4, quanta=q, away=away)
    out, _, _, _ = qb3.decode(ref)
    ref_out, _, _, _ = oracle.decode(ref)
    assert np.array_equal(out, ref_out)


@pytest.mark.parametrize("case", [(30, 18, 3, 0, "NOISY3", FTL), (70, 26, 1, 5, "DEM", FTL), (70, 26, 1, 7, "DEM", BASE), (70, 26, 1, 5, "DEM", 5),
                                  (70, 26, 1, 3, "DEM", 5), (70, 26, 1, 6, "TERRACE", 5), (66, 22, 8, 2, "LANDSAT16", BASE)],
                         ids=lambda c: "%dx%dx%d-t%d-%s-m%d" % c)
def test_stride(qb3, oracle, case):
    """a line stride (in values, QB3.h:116-120,146-148) on both sides: rows that start at odd multiples of the value size --
    for the lane-per-block kernels of every width (the 32/64-bit ones read and write rows as 16-byte pieces at value
    alignment only), FTL, BASE and the common-factor modes"""
    import ctypes as C
    w, h, b, dt, gen, mode = case
    npdt = {0: np.uint8, 2: np.uint16, 3: np.int16, 5: np.int32, 6: np.uint64, 7: np.int64}[dt]
    stride = w * b + 7
    canvas = np.zeros((h, stride), npdt)
    img = oracle.generate(w, h, b, dt, gen, 2)
    canvas[:, :w * b] = img.reshape(h, w * b)
    cb = None if b in (1, 3, 4) else list(range(b))
    L = qb3.lib
    p = L.qb3_create_encoder(w, h, b, dt)
    L.qb3_set_encoder_mode(p, mode)
    if cb is not None:
        arr = (C.c_size_t * b)(*cb)
        L.qb3_set_encoder_coreband(p, b, arr)
    L.qb3_set_encoder_stride(p, stride)
    dst = np.zeros(L.qb3_max_encoded_size(p), np.uint8)
    n = L.qb3_encode(p, canvas.ctypes.data, dst.ctypes.data)
    L.qb3_destroy_encoder(p)
    ref = oracle.encode(img, dt, mode, cband=cb)
    assert n == len(ref) and np.array_equal(dst[:n], ref)
    dims = (C.c_size_t * 3)()
    d = L.qb3_read_start(ref.ctypes.data, ref.size, dims)
    assert L.qb3_read_info(d)
    L.qb3_set_decoder_stride(d, stride)
    out = np.zeros((h, stride), npdt)
    assert L.qb3_read_data(d, out.ctypes.data) == img.nbytes
    L.qb3_destroy_decoder(d)
    assert np.array_equal(out[:, :w * b], img.reshape(h, w * b)) and not out[:, w * b:].any()
    # ... and straight into a strided device buffer (the device flavour decodes in place: no compact copy in between)
    import torch
    from qb3_amd import device as qdev
    dref = torch.from_numpy(ref.copy()).cuda()
    dec = qdev.DeviceDecoder(dref, len(ref))
    L.qb3_set_decoder_stride(dec.p, stride)
    dout = torch.zeros(h * stride * img.itemsize, dtype=torch.uint8, device="cuda")
    assert L.qb3x_decode_device(dec.p, dref.data_ptr(), dout.data_ptr(), None, None) == img.nbytes
    got = dout.cpu().numpy().view(npdt).reshape(h, stride)
    assert np.array_equal(got[:, :w * b], img.reshape(h, w * b)) and not got[:, w * b:].any()


@pytest.mark.parametrize("bands,mode", [(8, FTL), (8, BASE), (4, FTL), (2, BASE)])
def test_16bit_segments_far_longer_than_the_average(qb3, oracle, bands, mode):
    """the 16-bit lane-per-block decoder sizes its LDS staging for a third above the stream's AVERAGE segment; a flat image
    with a patch of noise has segments many times the average: the first launch reports them (status bit 4), the call is
    run again with the worst-case staging, and the pixels are exact -- with the index, from the stream alone, in a batch"""
    import ctypes as C
    import torch
    from qb3_amd import device as qdev
    w, h = 512, 256
    rng = np.random.default_rng(5)
    img = np.full((h, w, bands), 1000, dtype=np.uint16)
    img[64:160, 128:320, :] = rng.integers(0, 65536, size=(96, 192, bands), dtype=np.uint16)
    cb = [1, 1, 1] + list(range(3, bands)) if bands == 8 else None
    stream = oracle.encode(img, 2, mode, cband=cb)
    assert stream[10] != 255                                            # coded, not stored raw
    out, dims, dtype, _ = qb3.decode(stream)                            # host API: from the stream alone
    assert dims == (w, h, bands) and np.array_equal(out, img.view(np.uint8).ravel())
    enc = qdev.DeviceEncoder(w, h, bands, 2, mode=mode, cband=cb)
    dimg = torch.from_numpy(img.view(np.uint8).ravel().copy()).cuda()
    dst, n, index = enc.encode(dimg)
    assert n == len(stream) and np.array_equal(dst[:n].cpu().numpy(), stream)
    dec = qdev.DeviceDecoder(dst, n)
    assert torch.equal(dec.decode(dst, index=index).view(torch.uint8), dimg)
    assert torch.equal(dec.decode(dst, index=None).view(torch.uint8), dimg)


@pytest.mark.parametrize("dt", [4, 7])
@pytest.mark.parametrize("mode", [FTL, BASE, 7])
def test_wide_segments_far_longer_than_the_average(qb3, oracle, dt, mode):
    """the 32/64-bit lane-per-block decoders (dec_pxw_kernel, dec_pxw_best_kernel) size their LDS staging for half above the
    stream's AVERAGE segment like the 16-bit one: a flat raster with a patch of noise has segments many times that -- status
    bit 4, the call again with the worst case, exact pixels -- with the index, from the stream alone (the exit walk), from the
    container's own table, through the host API"""
    import torch
    from qb3_amd import device as qdev
    if dt == 7 and mode == 7:
        pytest.skip("64-bit noise in a common-factor mode: the reference's index-size sentinel (defect B-2) makes its stream undecodable; "
                    "the device encoder deliberately differs (DESIGN.md section 7) -- covered by test_encode_common_factor_modes")
    w, h = 512, 256
    rng = np.random.default_rng(7)
    npdt = np.uint32 if dt == 4 else np.int64
    img = np.full((h, w, 1), 1000, dtype=npdt)
    hi = 2 ** 31 if dt == 4 else 2 ** 62
    img[64:160, 128:320, :] = rng.integers(0, hi, size=(96, 192, 1)).astype(npdt)
    stream = oracle.encode(img, dt, mode)
    assert stream[10] != 255                                            # coded, not stored raw
    out, dims, dtype, _ = qb3.decode(stream)                            # host API: from the stream alone
    assert dims == (w, h, 1) and np.array_equal(out, img.view(np.uint8).ravel())
    got = qb3.encode(img, dt, mode)
    assert np.array_equal(got, stream)
    self_indexed = qb3.encode(img, dt, mode, index_chunk=2)
    out, _, _, _ = qb3.decode(self_indexed)
    assert np.array_equal(out, img.view(np.uint8).ravel())
    if stream[10] in (2, 3, 6, 7):
        # the RLE0 pass won: the table stands between the reference's header and its "DT", the RLE0 bytes are the reference's
        dt_at = bytes(stream).index(b"DT", 11)
        extra = len(self_indexed) - len(stream)
        assert extra > 0 and bytes(self_indexed[:dt_at]) == bytes(stream[:dt_at]) and bytes(self_indexed[dt_at + extra:]) == bytes(stream[dt_at:])
        want, _, _, _ = oracle.decode(self_indexed, identity=True)
        assert want is not None and np.array_equal(want, img.view(np.uint8).ravel()), "the reference decoder must step over the chunks"
        return                                                          # (... and the device flavour below is the plain modes')
    assert len(self_indexed) > len(stream)
    enc = qdev.DeviceEncoder(w, h, 1, dt, mode=mode)
    dimg = torch.from_numpy(img.view(np.uint8).ravel().copy()).cuda()
    dst, n, index = enc.encode(dimg)
    assert n == len(stream) and np.array_equal(dst[:n].cpu().numpy(), stream)
    dec = qdev.DeviceDecoder(dst, n)
    assert torch.equal(dec.decode(dst, index=index).view(torch.uint8), dimg)
    assert torch.equal(dec.decode(dst, index=None).view(torch.uint8), dimg)


@pytest.mark.parametrize("case", [(260, 130, 16, BASE), (515, 257, 12, FTL), (300, 200, 10, BASE), (640, 480, 14, BASE), (333, 222, 2, FTL), (400, 300, 1, BASE)],
                         ids=lambda c: "%dx%dx%d-m%d" % c)
def test_plain_16bit_streams_of_many_bands(qb3, oracle, case):
    """reference-made 16-bit containers (no index, no table) through the 16-bit table walk: blocks of up to 16 units, longer
    than a window of the walk, so the walk changes windows inside blocks; one and two bands take their own loops"""
    w, h, b, mode = case
    img = oracle.generate(w, h, b, 2, "LANDSAT16", 11)
    cb = [1, 1, 1] + list(range(3, b)) if b >= 3 else None
    stream = oracle.encode(img, 2, mode, cband=cb)
    out, dims, dtype, _ = qb3.decode(stream)
    assert dims == (w, h, b) and dtype == 2 and np.array_equal(out, img.view(np.uint8).ravel())


@pytest.mark.parametrize("shape", [(2, 40, 3), (40, 3, 1), (1, 17, 2), (300, 1, 3), (3, 3, 1)])
def test_narrow_images(qb3, oracle, shape):
    w, h, b = shape
    img = oracle.generate(w, h, b, 0, "NOISY3", 3)
    cb = None if b in (1, 3, 4) else [0] * b
    ref = check_encode(qb3, oracle, img, 0, 8, cband=cb)
    out, dims, _, _ = qb3.decode(ref)
    assert dims == (w, h, b) and np.array_equal(out, img.ravel())


def test_handle_state_carries_like_the_reference(qb3, oracle):
    """second qb3_encode without reset continues from the first image's band state (QB3encode.h:446-449)"""
    img = oracle.generate(64, 64, 3, 0, "NOISY3", 1)
    L = qb3.lib
    p = L.qb3_create_encoder(64, 64, 3, 0)
    e = oracle.Encoder(64, 64, 3, 0)
    dst = np.zeros(L.qb3_max_encoded_size(p), np.uint8)
    for _ in range(3):
        n = L.qb3_encode(p, img.ctypes.data, dst.ctypes.data)
        ref = e.encode(img)
        assert n == len(ref) and np.array_equal(dst[:n], ref)
    L.qb3_reset_encoder(p)
    e.reset()
    n = L.qb3_encode(p, img.ctypes.data, dst.ctypes.data)
    assert np.array_equal(dst[:n], e.encode(img))
    L.qb3_destroy_encoder(p)


def test_stored_fallback_and_sticky_mode(qb3, oracle):
    """incompressible data falls back to STORED and leaves the handle's mode at 255 (QB3encode.cpp:464,571-573)"""
    img = oracle.generate(64, 64, 1, 5, "RANDOM", 4)
    L = qb3.lib
    p = L.qb3_create_encoder(64, 64, 1, 5)
    e = oracle.Encoder(64, 64, 1, 5)
    dst = np.zeros(L.qb3_max_encoded_size(p), np.uint8)
    for _ in range(2):      # the second call runs the common-factor coder under the sticky STORED mode (defect B-5)
        n = L.qb3_encode(p, img.ctypes.data, dst.ctypes.data)
        ref = e.encode(img)
        assert n == len(ref) and np.array_equal(dst[:n], ref) and dst[10] == 255
    assert L.qb3_set_encoder_mode(p, 99) == 255          # the handle's mode stays at STORED
    L.qb3_destroy_encoder(p)
    out, _, _, mode = qb3.decode(dst[:n])
    assert mode == 255 and np.array_equal(out, img.view(np.uint8).ravel())


def test_table_room_survives_a_raw_fallback(qb3, oracle):
    """qb3_max_encoded_size on a handle whose mode a raw fallback left at QB3M_STORED still includes the restart table's room:
    a caller that sizes its buffer again then and sets a coding mode afterwards gets what the next call writes"""
    L = qb3.lib
    w = h = 128
    p = L.qb3_create_encoder(w, h, 1, 0)
    L.qb3x_set_encoder_index_chunk(p, 2)
    L.qb3_set_encoder_mode(p, FTL)
    room = L.qb3_max_encoded_size(p)
    dst = np.zeros(room, np.uint8)
    rnd = oracle.generate(w, h, 1, 0, "RANDOM", 4)
    n = L.qb3_encode(p, rnd.ctypes.data, dst.ctypes.data)
    assert n and dst[10] == 255 and L.qb3_set_encoder_mode(p, 99) == 255      # raw fallback, mode left at STORED
    assert L.qb3_max_encoded_size(p) == room
    L.qb3_reset_encoder(p)
    L.qb3_set_encoder_mode(p, FTL)
    assert L.qb3_max_encoded_size(p) == room
    img = oracle.generate(w, h, 1, 0, "NOISY3", 4)
    n = L.qb3_encode(p, img.ctypes.data, dst.ctypes.data)
    assert 0 < n <= room and dst[10] == FTL
    out, _, _, _ = qb3.decode(dst[:n])
    assert np.array_equal(out, img.ravel())
    L.qb3_destroy_encoder(p)


@pytest.mark.parametrize("case", [(8192, 4099, 3, 0, "NOISY3", FTL), (8192, 4099, 3, 0, "NOISY3", BASE), (16384, 8196, 1, 0, "NOISY3", FTL),
                                  (8192, 4100, 1, 5, "DEM", BASE), (4096, 8200, 4, 2, "LANDSAT16", BASE),
                                  (8192, 4099, 3, 0, "NOISY3", 5), (8192, 4100, 1, 5, "DEM", 5)],
                         ids=lambda c: "%dx%dx%d-t%d-m%d" % (c[0], c[1], c[2], c[3], c[5]))
def test_host_api_pipelined_strips(qb3, oracle, case):
    """qb3_encode / qb3_read_data on rasters large enough for the strip pipeline (encode_pipelined / decode_pipelined, qb3_api.cpp:
    three or more scan groups of chunks, upload / coding / download of different strips at once): the container is the
    oracle's byte for byte -- with a shifted last block row, for the 8-bit, 16-bit and 32-bit lane-per-block kernels, FTL and
    BASE -- plain and self-indexed (the table chunks aside), and decodes exactly through the host API (the self-indexed one strip
    by strip).  The common-factor modes (mode 5: QB3M_CF_H) code in one piece -- a factor is carried across chunks -- but their
    self-indexed containers decode strip by strip through the lane-per-block common-factor decoders"""
    w, h, b, dt, gen, mode = case
    img = oracle.generate(w, h, b, dt, gen, 5)
    ref = oracle.encode(img, dt, mode)
    assert ref[10] == mode
    got = qb3.encode(img, dt, mode)
    assert len(got) == len(ref) and np.array_equal(got, ref), f"first diff at byte {first_diff(got, ref)} of {len(ref)}"
    two = qb3.encode(img, dt, mode, index_chunk=2)
    extra, dt_at = len(two) - len(ref), bytes(ref[:64]).index(b"DT", 11)
    assert extra > 0 and np.array_equal(two[:dt_at], ref[:dt_at]) and np.array_equal(two[dt_at + extra:], ref[dt_at:])
    out, dims, _, _ = qb3.decode(two)
    assert dims == (w, h, b) and np.array_equal(out, img.view(np.uint8).ravel())
    # the handle's band state after the pipelined call is what the reference's is: a second call on the same handle
    L = qb3.lib
    p = L.qb3_create_encoder(w, h, b, dt)
    e = oracle.Encoder(w, h, b, dt)
    L.qb3_set_encoder_mode(p, mode)
    e.set_mode(mode)
    dst = np.empty(L.qb3_max_encoded_size(p), np.uint8)
    for _ in range(2):
        n = L.qb3_encode(p, img.ctypes.data, dst.ctypes.data)
        want = e.encode(img)
        assert n == len(want) and np.array_equal(dst[:n], want)
    L.qb3_destroy_encoder(p)


def test_overlong_stream_fails_like_the_reference(qb3, oracle):
    """more than 7 unused bits after the last unit => read_data returns 0 (QB3decode.h:411,569)"""
    img = oracle.generate(32, 32, 3, 0, "NOISY3", 1)
    s = oracle.encode(img, 0, 8)
    assert oracle.decode(np.concatenate([s, np.zeros(2, np.uint8)]))[0] is None
    with pytest.raises(RuntimeError):
        qb3.decode(np.concatenate([s, np.zeros(2, np.uint8)]))
    qb3.decode(s)


@pytest.mark.parametrize("shape", [(256, 256, 3, 0), (251, 121, 3, 0), (130, 67, 8, 2), (61, 35, 1, 5)],
                         ids=lambda c: "%dx%dx%d-t%d" % c)
def test_tiles_api(qb3, oracle, shape):
    """one call, several tiles: aligned 8-bit tiles, tiles whose size makes every pitch odd (unaligned rows and tile
    starts), 16-bit groups, a 32-bit tile through the generic kernels; with the index and without"""
    import ctypes as C
    import torch
    from qb3_amd import synth
    L = qb3.lib
    w, h, b, dt = shape
    tsz = oracle.TYPESIZE[dt]
    n = 6
    gen = "NOISY3" if dt == 0 else ("LANDSAT16" if dt == 2 else "DEM")
    imgs = torch.stack([synth.generate(w, h, b, dt, gen, 1000 + t) for t in range(n)])
    p = L.qb3_create_encoder(w, h, b, dt)
    pitch = (L.qb3_max_encoded_size(p) + 3) // 4 * 4
    isz = L.qb3x_index_size(p)
    dst = torch.zeros(n * pitch, dtype=torch.uint8, device="cuda")
    idx = torch.zeros(n * isz, dtype=torch.uint8, device="cuda")
    sizes = (C.c_size_t * n)()
    raw = w * h * b * tsz
    assert L.qb3x_encode_tiles(p, imgs.data_ptr(), n, raw, dst.data_ptr(), pitch, idx.data_ptr(), sizes, None) == n
    L.qb3_destroy_encoder(p)
    host = dst.cpu().numpy()
    for t in range(n):
        ref = oracle.encode(imgs[t].cpu().numpy().view(oracle.NPTYPE[dt]), dt, 8)
        assert sizes[t] == len(ref) and np.array_equal(host[t * pitch:t * pitch + sizes[t]], ref)
    dims = (C.c_size_t * 3)()
    hdr = host[:64].copy()
    d = L.qb3_read_start(hdr.ctypes.data, sizes[0], dims)
    assert L.qb3_read_info(d)
    if b not in (1, 3, 4):
        L.qb3x_set_decoder_compat(d, 0)     # identity map when the container has no CB chunk (the default)
    for index in (idx.data_ptr(), None):
        out = torch.zeros_like(imgs)
        assert L.qb3x_decode_tiles(d, dst.data_ptr(), n, pitch, sizes, out.data_ptr(), raw, index, None) == n
        assert torch.equal(out, imgs)
    L.qb3_destroy_decoder(d)


@pytest.mark.parametrize("shape", [(256, 192, 3, 0, FTL), (200, 136, 1, 0, BASE), (256, 128, 8, 2, BASE), (192, 128, 1, 5, FTL), (256, 192, 3, 0, 5)],
                         ids=lambda s: "%dx%dx%d-t%d-m%d" % s)
def test_tiles_with_restart_tables(qb3, oracle, shape):
    """qb3x_encode_tiles with qb3x_set_encoder_index_chunk: every tile's container is what qb3_encode writes for it with the
    switch on (the reference's stream between this library's table chunks), and qb3x_decode_tiles(index = NULL) decodes the
    batch from the containers alone -- also when one tile of the batch is raw-stored and one lost its table"""
    import ctypes as C
    import torch
    from qb3_amd import synth
    L = qb3.lib
    w, h, b, dt, mode = shape
    tsz = oracle.TYPESIZE[dt]
    n = 5
    gen = "NOISY3" if dt == 0 else ("LANDSAT16" if dt == 2 else "DEM")
    imgs = torch.stack([synth.generate(w, h, b, dt, gen, 4000 + t) for t in range(n)])
    if dt == 0 and b == 3 and mode == FTL:
        imgs[3] = synth.generate(w, h, b, dt, "RANDOM", 7)               # this one falls back to QB3M_STORED
    p = L.qb3_create_encoder(w, h, b, dt)
    L.qb3_set_encoder_mode(p, mode)
    L.qb3x_set_encoder_index_chunk(p, 1)
    pitch = (L.qb3_max_encoded_size(p) + 3) // 4 * 4
    dst = torch.zeros(n * pitch, dtype=torch.uint8, device="cuda")
    sizes = (C.c_size_t * n)()
    raw = w * h * b * tsz
    assert L.qb3x_encode_tiles(p, imgs.data_ptr(), n, raw, dst.data_ptr(), pitch, None, sizes, None) == n
    host = dst.cpu().numpy()
    for t in range(n):
        one = host[t * pitch:t * pitch + sizes[t]]
        ref = oracle.encode(imgs[t].cpu().numpy().view(oracle.NPTYPE[dt]), dt, mode)
        if one[10] == 255:
            assert np.array_equal(one, ref)
            continue
        # the reference's container with this library's chunks inserted in front of "DT"
        extra = len(one) - len(ref)
        dt_at = bytes(ref).index(b"DT", 11)
        assert extra > 0 and bytes(one[dt_at:dt_at + 2]) == b"ix"
        assert bytes(one[:dt_at]) == bytes(ref[:dt_at]) and bytes(one[dt_at + extra:]) == bytes(ref[dt_at:]), "tile %d" % t
    dims = (C.c_size_t * 3)()
    need = L.qb3x_header_size_bound(host[:64].copy().ctypes.data, 64)
    hdr = host[:max(need, 64)].copy()
    d = L.qb3x_read_start(hdr.ctypes.data, hdr.size, sizes[0], dims)
    assert d and L.qb3_read_info(d)
    if b not in (1, 3, 4):
        L.qb3x_set_decoder_compat(d, 0)     # identity map when the container has no CB chunk (the default)
    out = torch.zeros_like(imgs)
    L.qb3x_profile_reset()
    L.qb3x_profile_enable(1)
    assert L.qb3x_decode_tiles(d, dst.data_ptr(), n, pitch, sizes, out.data_ptr(), raw, None, None) == n
    L.qb3x_profile_enable(0)
    names = C.create_string_buffer(1024)
    L.qb3x_profile_names(names, 1024)
    assert torch.equal(out, imgs)
    # the tables were used: no table of unit lengths by position was built, no index rebuilt by parsing
    assert b"dec_index_table" not in names.value, names.value
    if mode == 5:
        assert b"dec_index_serial" not in names.value, names.value
    L.qb3_destroy_decoder(d)
    L.qb3_destroy_encoder(p)


@pytest.mark.parametrize("stored_at", [0, 2, 5])
def test_tiles_api_with_a_stored_tile(qb3, oracle, stored_at):
    """a batch in which one tile is incompressible: qb3x_encode_tiles writes it raw (QB3M_STORED, another header layout),
    the handle's mode survives the call, and qb3x_decode_tiles decodes the whole batch whatever the position of that
    tile -- also first, where it is the tile the decoder handle was parsed from"""
    import ctypes as C
    import torch
    from qb3_amd import synth
    L = qb3.lib
    w, h, b, dt, n = 128, 64, 3, 0, 6
    imgs = torch.stack([synth.generate(w, h, b, dt, "RANDOM" if t == stored_at else "NOISY3", 2000 + t) for t in range(n)])
    p = L.qb3_create_encoder(w, h, b, dt)
    pitch = (L.qb3_max_encoded_size(p) + 3) // 4 * 4
    isz = L.qb3x_index_size(p)
    dst = torch.zeros(n * pitch, dtype=torch.uint8, device="cuda")
    idx = torch.zeros(n * isz, dtype=torch.uint8, device="cuda")
    sizes = (C.c_size_t * n)()
    raw = w * h * b
    for rep in range(2):        # the second call must not inherit QB3M_STORED from the fallback inside the first
        assert L.qb3x_encode_tiles(p, imgs.data_ptr(), n, raw, dst.data_ptr(), pitch, idx.data_ptr(), sizes, None) == n
        host = dst.cpu().numpy()
        for t in range(n):
            ref = oracle.encode(imgs[t].cpu().numpy(), dt, 8)
            assert sizes[t] == len(ref) and np.array_equal(host[t * pitch:t * pitch + sizes[t]], ref)
            assert (host[t * pitch + 10] == 255) == (t == stored_at)
    L.qb3_destroy_encoder(p)
    dims = (C.c_size_t * 3)()
    hdr = host[:64].copy()
    d = L.qb3_read_start(hdr.ctypes.data, sizes[0], dims)
    assert L.qb3_read_info(d)
    for index in (idx.data_ptr(), None):
        out = torch.zeros_like(imgs)
        assert L.qb3x_decode_tiles(d, dst.data_ptr(), n, pitch, sizes, out.data_ptr(), raw, index, None) == n
        assert all(L.qb3x_decode_tile_ok(d, t) == 1 for t in range(n))
        assert torch.equal(out, imgs)
    # a damaged tile is reported, the others still decode
    dst[3 * pitch + 40:3 * pitch + 40 + 64] = 0xff if stored_at != 3 else 0
    bad_sizes = (C.c_size_t * n)(*sizes)
    bad_sizes[3] = sizes[3] - 7 if stored_at == 3 else sizes[3] + 9
    out = torch.zeros_like(imgs)
    k = L.qb3x_decode_tiles(d, dst.data_ptr(), n, pitch, bad_sizes, out.data_ptr(), raw, None, None)
    assert k == n - 1 and L.qb3x_decode_tile_ok(d, 3) == 0 and all(L.qb3x_decode_tile_ok(d, t) == 1 for t in range(n) if t != 3)
    L.qb3_destroy_decoder(d)


def test_strided_stored_containers(qb3, oracle):
    """the line stride is in VALUES on every path (QB3.h:116-120,146-148), also when a 16-bit image falls back to raw
    storage: strided encode of incompressible data, strided decode of the STORED container, host and device flavour"""
    import ctypes as C
    import torch
    L = qb3.lib
    w, h, b, dt = 48, 20, 2, 2
    img = oracle.generate(w, h, b, dt, "RANDOM", 3)
    stride = w * b + 6
    wide = np.zeros((h, stride), dtype=np.uint16)
    wide[:, :w * b] = img.reshape(h, w * b)
    ref = oracle.encode(img, dt, 8)
    assert ref[10] == 255
    got = qb3.encode(img, dt, 8)
    assert np.array_equal(got, ref)
    p = L.qb3_create_encoder(w, h, b, dt)
    L.qb3_set_encoder_stride(p, stride)
    buf = np.zeros(L.qb3_max_encoded_size(p), dtype=np.uint8)
    n = L.qb3_encode(p, wide.ctypes.data, buf.ctypes.data)
    L.qb3_destroy_encoder(p)
    assert n == len(ref) and np.array_equal(buf[:n], ref)
    dims = (C.c_size_t * 3)()
    for device in (False, True):
        d = L.qb3_read_start(ref.ctypes.data, len(ref), dims)
        assert L.qb3_read_info(d)
        L.qb3_set_decoder_stride(d, stride)
        out = np.zeros((h, stride), dtype=np.uint16)
        if device:
            dsrc = torch.from_numpy(np.concatenate([ref, np.zeros((-len(ref)) % 4, np.uint8)])).cuda()
            dout = torch.zeros(h * stride * 2, dtype=torch.uint8, device="cuda")
            assert L.qb3x_decode_device(d, dsrc.data_ptr(), dout.data_ptr(), None, None) == w * h * b * 2
            out = dout.cpu().numpy().view(np.uint16).reshape(h, stride)
        else:
            assert L.qb3_read_data(d, out.ctypes.data) == w * h * b * 2
        L.qb3_destroy_decoder(d)
        assert np.array_equal(out[:, :w * b], img.reshape(h, w * b)) and not out[:, w * b:].any()


@pytest.mark.parametrize("case", [(2048, 2048, 1, 7, "TERRACE", 4, 1), (2048, 2048, 1, 7, "TERRACE", 4, 5), (2048, 1024, 3, 0, "NOISY3", 2, 8),
                                  (1024, 1024, 8, 2, "LANDSAT16", 3, 4), (2048, 2048, 1, 5, "FEW", 4, 7), (1021, 515, 3, 0, "NOISY3", 1, 4)],
                         ids=lambda c: "%dx%dx%d-t%d-%s-m%d" % (c[0], c[1], c[2], c[3], c[4], c[6]))
def test_encode_is_repeatable(qb3, oracle, case):
    """The same input must give the same bytes every time (a single-shot parity check can miss a racy kernel:
    an earlier build flipped isolated bits in i64 common-factor units in about half of the runs)."""
    import torch
    from qb3_amd import synth, device as qdev
    w, h, b, dt, gen, seed, mode = case
    img = synth.generate(w, h, b, dt, gen, seed)
    cb = None if b in (1, 3, 4) else [1, 1, 1] + list(range(3, b))
    ref = oracle.encode(img.cpu().numpy().view(oracle.NPTYPE[dt]), dt, mode, cband=cb, fix_b2=True)
    enc = qdev.DeviceEncoder(w, h, b, dt, mode=mode, cband=cb)
    for rep in range(12):
        if rep % 3 == 1 and enc.dst is not None:
            enc.dst = torch.full_like(enc.dst, 0xAA)       # a different, dirty destination buffer
        dst, n, _ = enc.encode(img)
        got = dst[:n].cpu().numpy()
        assert n == len(ref) and np.array_equal(got, ref), f"run {rep}: first diff at byte {first_diff(got, ref)}"


@pytest.mark.parametrize("order", [0x0123456789abcdef, 0xfedcba9876543210, 0x048c159d26ae37bf, 0x05af16b827c93de4])
@pytest.mark.parametrize("case", [(64, 48, 3, 0, "NOISY3", 1, 8), (96, 64, 1, 5, "DEM", 4, 4), (40, 44, 4, 0, "NOISY3", 8, 5), (64, 64, 1, 7, "TERRACE", 4, 7)],
                         ids=lambda c: "%dx%dx%d-t%d-%s-m%d" % (c[0], c[1], c[2], c[3], c[4], c[6]))
def test_decode_custom_scan_curve(qb3, oracle, case, order):
    """streams carrying their own scan curve in an SC chunk (reference QB3decode.cpp:231-250): the reference's
    encoder API cannot make them, its decoder reads them -- so must this one (generic kernels, run-time curve)"""
    import ctypes as C
    w, h, b, dt, gen, seed, mode = case
    img = oracle.generate(w, h, b, dt, gen, seed)
    stream = oracle.encode(img, dt, mode, order=order)
    assert bytes(stream[11:13]) in (b"CB", b"SC")
    out, dims, dtype, m = qb3.decode(stream)
    assert dims == (w, h, b) and np.array_equal(out, img.view(np.uint8).ravel())
    dimsv = (C.c_size_t * 3)()
    p = qb3.lib.qb3_read_start(stream.ctypes.data, stream.size, dimsv)
    assert qb3.lib.qb3_read_info(p) and qb3.lib.qb3_get_order(p) == order
    qb3.lib.qb3_destroy_decoder(p)


def _cli(*args):
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "qb3_amd", "cqb3x")
    return subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True, timeout=300)


@pytest.mark.parametrize("case", [
    # (w, h, bands, dtype, gen, flags, reference mode, quanta, trim)
    (203, 131, 3, 0, "NOISY3", [], 4, 1, False),            # cqb3 default: QB3M_BASE
    (203, 131, 3, 0, "NOISY3", ["-f"], 8, 1, False),
    (203, 131, 3, 0, "NOISY3", ["-b"], 7, 1, False),
    (203, 131, 3, 0, "NOISY3", ["-l"], 0, 1, False),
    (203, 131, 3, 0, "NOISY3", ["-b", "-l"], 3, 1, False),
    (203, 131, 3, 0, "NOISY3", ["-r"], 6, 1, False),
    (203, 131, 3, 0, "NOISY3", ["-l", "-r"], 2, 1, False),
    (203, 131, 3, 0, "NOISY3", ["-b", "-r"], 5, 1, False),
    (203, 131, 3, 0, "NOISY3", ["-b", "-l", "-r"], 1, 1, False),
    (203, 131, 3, 0, "NOISY3", ["-m"], 4, 1, False),
    (203, 131, 3, 0, "NOISY3", ["-m", "2,2,2"], 4, 1, False),
    (203, 131, 3, 0, "NOISY3", ["-q", "5"], 4, 5, False),
    (203, 131, 3, 0, "NOISY3", ["-q", "+4"], 4, 4, False),
    (203, 131, 3, 0, "NOISY3", ["-t"], 4, 1, True),
    (130, 67, 1, 2, "LANDSAT16", ["-b"], 7, 1, False),
    (130, 67, 1, 2, "LANDSAT16", ["-t", "-f"], 8, 1, True),
], ids=lambda c: "%dx%dx%d-t%d%s" % (c[0], c[1], c[2], c[3], "".join(c[5])))
def test_cli_pnm_roundtrip(qb3, oracle, tmp_path, case):
    """tools/cqb3x.cpp, the cqb3 counterpart (reference cqb3.cpp:405-493 encode, :276-323 decode): PNM in, the same
    .qb3 bytes the reference library writes for those options, PNM back out"""
    w, h, b, dt, gen, flags, mode, quanta, trim = case
    img = oracle.generate(w, h, b, dt, gen, 3)
    pnm = tmp_path / "in.pnm"
    body = img.astype(">u2").tobytes() if dt == 2 else img.tobytes()
    pnm.write_bytes(b"P%c\n# made by the test\n%d %d\n%d\n" % (b"56"[b == 3], w, h, 255 if dt == 0 else 65535) + body)
    r = _cli("-v", *flags, pnm, tmp_path / "out.qb3")
    assert r.returncode == 0, r.stderr
    got = np.fromfile(tmp_path / "out.qb3", dtype=np.uint8)
    src, stride = img, 0
    if trim:        # cqb3.cpp:393-402
        x0, y0 = int(w % 4 > 1), int(h % 4 > 1)
        src = np.ascontiguousarray(img[y0:y0 + h - h % 4, x0:x0 + w - w % 4])
    cband = None
    if "-m" in flags:
        cband = [2, 2, 2] if "2,2,2" in flags else list(range(b))
    away = any(f.startswith("+") for f in flags)
    want = oracle.encode(src, dt, mode, cband=cband, quanta=quanta, away=away)
    assert np.array_equal(got, want)
    r = _cli("-d", "-v", tmp_path / "out.qb3", tmp_path / "back.pnm")
    assert r.returncode == 0, r.stderr
    back = (tmp_path / "back.pnm").read_bytes()
    hh, ww = src.shape[:2]
    hdr = b"P%c\n%d %d\n%d\n" % (b"56"[b == 3], ww, hh, 255 if dt == 0 else 65535)
    assert back.startswith(hdr)
    ref, _, _, _ = oracle.decode(want, identity=True)      # no CB chunk = identity map
    px = np.frombuffer(back[len(hdr):], dtype=np.uint8)
    if dt == 2:
        px = px.view(">u2").astype("<u2").view(np.uint8)
    assert np.array_equal(px, ref)
    if quanta == 1:
        assert np.array_equal(px, src.view(np.uint8).ravel())


def test_cli_raw_and_errors(qb3, oracle, tmp_path):
    img = oracle.generate(64, 40, 5, 5, "DEM", 2)       # 5 bands of int32: no PNM form
    (tmp_path / "a.raw").write_bytes(img.tobytes())
    r = _cli("-s", "64,40,5,5", "-m", tmp_path / "a.raw")
    assert r.returncode == 0, r.stderr
    got = np.fromfile(tmp_path / "a.qb3", dtype=np.uint8)
    assert np.array_equal(got, oracle.encode(img, 5, 4, cband=list(range(5))))
    r = _cli("-d", "-s", tmp_path / "a.qb3")
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "a.raw").read_bytes() == img.tobytes()
    assert _cli("-d", tmp_path / "a.raw").returncode == 1           # not a QB3 stream
    assert _cli("-x", tmp_path / "a.raw").returncode == 2
    assert _cli().returncode == 2
    # a truncated stream is not an error in the reference: its reader returns zeros past the end (bitstream.h:36)
    # and only MORE than 7 unused bits fail (QB3decode.h:411,569,740) -- same pixels as the oracle, no crash
    cut = got[:200].copy()
    (tmp_path / "t.qb3").write_bytes(cut.tobytes())
    ref, _, _, _ = oracle.decode(cut, identity=True)
    r = _cli("-d", "-s", tmp_path / "t.qb3", tmp_path / "t.raw")
    if ref is None:
        assert r.returncode == 1
    else:
        assert r.returncode == 0 and (tmp_path / "t.raw").read_bytes() == ref.tobytes()
    (tmp_path / "l.qb3").write_bytes(got.tobytes() + b"\0\0")      # two spare bytes at the end: over-long, fails
    assert _cli("-d", "-s", tmp_path / "l.qb3", tmp_path / "l.raw").returncode == 1


@pytest.mark.parametrize("mode", [FTL, BASE, BASE_Z])
@pytest.mark.parametrize("case", [(256, 64, 8, 2, "LANDSAT16", 3, [1, 1, 1, 3, 4, 5, 6, 7]), (128, 36, 12, 3, "DEM", 5, [1, 1, 1] + list(range(3, 12))),
                                  (64, 64, 16, 2, "NOISY3", 6, [1, 1, 1] + list(range(3, 16))), (128, 64, 4, 2, "LANDSAT16", 7, [1, 1, 1, 3]),
                                  (128, 64, 8, 2, "LANDSAT16", 8, [0, 1, 2, 3, 5, 5, 5, 7]), (64, 32, 6, 2, "LANDSAT16", 9, [1, 1, 1, 3, 4, 5])],
                         ids=lambda c: "%dx%dx%d-t%d" % c[:4] if isinstance(c, tuple) else None)
def test_band_maps_on_multiband_16bit(qb3, oracle, case, mode):
    """SURVEY.md B-1 recommends {1,1,1,3,4,5,6,7} for the 8-band configuration: R-G,G,B-G on the first three bands
    inside a 16-bit band group takes the lane-per-block kernels, other maps the generic ones; all must match"""
    w, h, b, dt, gen, seed, cband = case
    img = oracle.generate(w, h, b, dt, gen, seed)
    ref = check_encode(qb3, oracle, img, dt, mode, cband=cband)
    out, dims, dtype, m = qb3.decode(ref)
    assert np.array_equal(out, img.view(np.uint8).ravel())


IX_CASES = [
    # w, h, bands, dtype, gen, seed, mode
    (512, 512, 3, 0, "NOISY3", 1, FTL),          # 8-bit lane-per-block kernels, length walker
    (509, 203, 3, 0, "NOISY3", 2, BASE),         # odd size: generic kernels, value walker
    (256, 256, 8, 2, "LANDSAT16", 3, BASE),      # 16-bit groups
    (1024, 64, 1, 0, "GRAD", 0, BASE_Z),
    (256, 128, 1, 5, "DEM", 4, FTL),             # 32-bit: generic
    (128, 128, 1, 7, "DEM", 4, 5),               # 64-bit common factor: entries carry cf
    (96, 64, 1, 5, "TERRACE", 4, 5),
    (64, 48, 3, 0, "PALETTE", 3, 1),
    (2048, 2048, 3, 0, "NOISY3", 5, FTL),        # more segments than entries fit: several segments per entry
]


def walk_chunks(c, fixed_b7):
    """Chunk walk of a container the way the reference does it (QB3decode.cpp:193-258): known upper-case chunks by their
    payload length, unknown lower-case ones by `len` bytes from the chunk START (defect B-7) -- or, fixed_b7, by the
    4 head bytes plus `len`, what a corrected reader would do.  Returns (offset of "DT", list of (tag, offset, len))."""
    pos, seen = 11, []
    while True:
        tag, ln = bytes(c[pos:pos + 2]), int(c[pos + 2]) | int(c[pos + 3]) << 8
        if tag == b"DT":
            return pos, seen
        seen.append((tag, pos, ln))
        if tag in (b"CB", b"QV", b"SC"):
            pos += 4 + ln
        else:
            assert tag[0] & 0x20 and ln, "unknown upper-case chunk"
            pos += ln + (4 if fixed_b7 else 0)
        assert pos < len(c)


IX_CASES.append((8192, 4096, 3, 0, "NOISY3", 6, FTL))      # more entries than one 64 KB chunk holds
IX_CASES.append((256, 128, 1, 5, "DEM", 4, 7))              # an RLE0 mode whose RLE0 pass does not win: the table stays
IX_CASES.append((520, 300, 1, 7, "DEM", 4, BASE))           # 64-bit: the lengths-only walk, one read a code
IX_CASES.append((256, 256, 1, 6, "RUNG63", 4, FTL))         # ... with 65-bit codes
IX_CASES.append((160, 120, 5, 4, "DEM", 4, FTL))            # ... 32-bit, five bands, a band map


@pytest.mark.parametrize("case", IX_CASES, ids=lambda c: "%dx%dx%d-t%d-%s-m%d" % (c[0], c[1], c[2], c[3], c[4], c[6]))
def test_index_chunk(qb3, oracle, case):
    """qb3x_set_encoder_index_chunk: the container gains ignorable chunks ("ix" + "zz" pad pairs) in front of "DT" and
    nothing else changes; the reference's decoder (the oracle restates its skip rule, QB3decode.cpp:251-255) steps
    over them, and so would a reader with defect B-7 fixed; this library decodes through them, host and device flavour"""
    import torch
    from qb3_amd import synth, device as qdev
    w, h, b, dt, gen, seed, mode = case
    cb = None if b in (1, 3, 4) else [1, 1, 1] + list(range(3, b))
    img = oracle.generate(w, h, b, dt, gen, seed)
    ref = oracle.encode(img, dt, mode, cband=cb)
    got = qb3.encode(img, dt, mode, cband=cb, index_chunk=True)
    plain = qb3.encode(img, dt, mode, cband=cb)
    assert np.array_equal(plain, ref)
    # same container with the table's chunks inserted in front of "DT"
    extra = len(got) - len(ref)
    dt_at = bytes(ref).index(b"DT", 11)
    assert bytes(got[:dt_at]) == bytes(ref[:dt_at]) and bytes(got[dt_at + extra:]) == bytes(ref[dt_at:])
    for fixed in (False, True):
        at, seen = walk_chunks(got, fixed)
        assert at == dt_at + extra
    _, seen = walk_chunks(got, False)
    mine = [c for c in seen if c[1] >= dt_at]
    assert len(mine) % 2 == 0 and len(mine) >= 2
    entries = 0
    for ix, zz in zip(mine[0::2], mine[1::2]):
        assert ix[0] == b"ix" and zz[0] == b"zz" and zz[2] == 4 and zz[1] == ix[1] + ix[2] and got[ix[1] + 4] == 3        # (version 3: a check of the entries in the head's reserved bytes)
        entries += (ix[2] - 12)
    assert extra == entries + 16 * (len(mine) // 2)
    if w * h >= 8192 * 4096:
        assert len(mine) >= 4, "this case is meant to need more than one chunk"
    raw = img.view(np.uint8).ravel()
    out, _, _, _ = oracle.decode(got, identity=True)
    assert out is not None and np.array_equal(out, raw), "the reference decoder must step over the chunks"
    out, dims, dtype, m = qb3.decode(got)
    assert np.array_equal(out, raw)
    # device flavour: container made on the device, decoded with and without the out-of-band index
    enc = qdev.DeviceEncoder(w, h, b, dt, mode=mode, cband=cb, index_chunk=True)
    dimg = torch.from_numpy(img.view(np.uint8).copy()).cuda()
    dst, n, index = enc.encode(dimg)
    assert n == len(got) and np.array_equal(dst[:n].cpu().numpy(), got)
    dec = qdev.DeviceDecoder(dst, n)
    draw = dimg.reshape(-1).view(torch.uint8)
    assert torch.equal(dec.decode(dst, index=None), draw)
    assert torch.equal(dec.decode(dst, index=index), draw)


def test_index_chunk_version_1_still_decodes(qb3, oracle):
    """round-1 containers carry ONE "ix" chunk, version 1, no pad chunk, no check: rebuilt here from a current container; a
    round-2 container (version 2: pads, no check) likewise"""
    img = oracle.generate(512, 512, 3, 0, "NOISY3", 1)
    got = qb3.encode(img, 0, FTL, index_chunk=True)
    at = bytes(got).index(b"ix", 11)
    ln = int(got[at + 2]) | int(got[at + 3]) << 8
    assert bytes(got[at + ln:at + ln + 4]) == b"zz\x04\x00" and bytes(got[at + ln + 4:at + ln + 6]) == b"DT"
    v1 = np.concatenate([got[:at + ln], got[at + ln + 4:]])
    v1[at + 4] = 1
    v1[at + 6] = v1[at + 7] = 0
    out, _, _, _ = qb3.decode(v1)
    assert np.array_equal(out, img.ravel())
    out, _, _, _ = oracle.decode(v1, identity=True)
    assert np.array_equal(out, img.ravel())
    v2 = got.copy()
    v2[at + 4] = 2
    v2[at + 6] = v2[at + 7] = 0
    out, _, _, _ = qb3.decode(v2)
    assert np.array_equal(out, img.ravel())


def test_index_chunk_not_written_where_it_cannot_be(qb3, oracle):
    """narrow images and STORED output carry no table; a winning RLE0 pass keeps it (the table describes the block stream the
    decoder has again after the expansion), like the base mode's container when the pass loses"""
    img = oracle.generate(96, 64, 1, 5, "TERRACE", 4)
    for mode in (2, 3, 6, 7):
        ref, got = oracle.encode(img, 5, mode), qb3.encode(img, 5, mode, index_chunk=True)
        dt_at, extra = bytes(ref).index(b"DT", 11), len(got) - len(ref)
        assert extra > 0 and bytes(got[dt_at:dt_at + 2]) == b"ix"
        assert np.array_equal(np.concatenate([got[:dt_at], got[dt_at + extra:]]), ref)
        out, _, _, _ = qb3.decode(got)
        assert np.array_equal(out, img.view(np.uint8).ravel())
    narrow = oracle.generate(2, 40, 3, 0, "NOISY3", 1)
    assert np.array_equal(qb3.encode(narrow, 0, FTL, index_chunk=True), oracle.encode(narrow, 0, FTL))
    noise = oracle.generate(64, 64, 3, 0, "RANDOM", 1)         # incompressible: falls back to STORED
    assert np.array_equal(qb3.encode(noise, 0, FTL, index_chunk=True), oracle.encode(noise, 0, FTL))


def test_index_chunk_is_checked_not_trusted(qb3, oracle):
    """a table that does not fit the geometry is ignored (serial walk), a damaged one cannot crash the decoder"""
    img = oracle.generate(256, 256, 3, 0, "NOISY3", 9)
    got = qb3.encode(img, 0, FTL, index_chunk=True)
    at = bytes(got).index(b"ix", 11)
    bad = got.copy()
    bad[at + 8] ^= 1                    # blocks per entry no longer a multiple of the segment size
    out, _, _, _ = qb3.decode(bad)
    assert np.array_equal(out, img.ravel())
    worse = got.copy()
    worse[at + 12 + 2] ^= 0x55          # a bit position pointing elsewhere: the chunk's check fails, the stream is walked instead
    out, _, _, _ = qb3.decode(worse)
    assert np.array_equal(out, img.ravel())
    out, _, _, _ = qb3.decode(got)      # and the handle-free API is still healthy
    assert np.array_equal(out, img.ravel())


@pytest.mark.parametrize("mode", [FTL, BASE])
@pytest.mark.parametrize("off", [1, 2, 3])
@pytest.mark.parametrize("shape", [(251, 37, 3), (64, 20, 4), (509, 12, 1)])
def test_unaligned_device_pointers(qb3, oracle, shape, off, mode):
    """the 8-bit lane-per-block kernels read and write rows at any alignment: odd widths, and rasters that start at an
    odd address inside a device buffer (source and destination)"""
    import torch
    from qb3_amd import synth, device as qdev
    w, h, b = shape
    img = synth.generate(w, h, b, 0, "NOISY3", 11)
    raw = img.reshape(-1)
    buf = torch.zeros(raw.numel() + 8, dtype=torch.uint8, device="cuda")
    src = buf[off:off + raw.numel()]
    src.copy_(raw)
    enc = qdev.DeviceEncoder(w, h, b, 0, mode=mode)
    dst, n, index = enc.encode(src)
    ref = oracle.encode(img.cpu().numpy(), 0, mode)
    assert n == len(ref) and np.array_equal(dst[:n].cpu().numpy(), ref)
    dec = qdev.DeviceDecoder(dst, n)
    guard = torch.full((raw.numel() + 8,), 0xA5, dtype=torch.uint8, device="cuda")
    out = guard[off:off + raw.numel()]
    dec.decode(dst, out=out, index=index)
    assert torch.equal(out, raw)
    assert bool((guard[:off] == 0xA5).all()) and bool((guard[off + raw.numel():] == 0xA5).all()), "wrote outside the raster"


def test_handles_are_independent_across_threads(qb3, oracle):
    """handles share nothing.  Four host threads, each with its own encoder and decoder handles and its own
    raster, code concurrently (ctypes drops the GIL inside the library); every stream must equal the oracle's"""
    import threading
    cases = [(256, 128, 3, 0, "NOISY3", 8), (200, 100, 1, 5, "DEM", 4), (128, 128, 8, 2, "LANDSAT16", 8), (96, 64, 1, 7, "DEM", 5)]
    imgs = [oracle.generate(w, h, b, dt, gen, 30 + i) for i, (w, h, b, dt, gen, m) in enumerate(cases)]
    refs = [oracle.encode(img, c[3], c[5], cband=None if c[2] in (1, 3, 4) else list(range(c[2]))) for img, c in zip(imgs, cases)]
    errors = []

    def work(i):
        w, h, b, dt, gen, mode = cases[i]
        cb = None if b in (1, 3, 4) else list(range(b))
        try:
            for _ in range(8):
                got = qb3.encode(imgs[i], dt, mode, cband=cb)
                if not np.array_equal(got, refs[i]):
                    errors.append("thread %d: stream differs" % i)
                    return
                out, _, _, _ = qb3.decode(got)
                if not np.array_equal(out, imgs[i].view(np.uint8).ravel()):
                    errors.append("thread %d: pixels differ" % i)
                    return
        except Exception as e:      # noqa: BLE001 -- report whatever a worker thread hit
            errors.append("thread %d: %r" % (i, e))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(cases))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


@pytest.mark.parametrize("case", [(512, 384, 3, 0, "NOISY3", FTL), (512, 384, 1, 0, "NOISY3", BASE), (512, 384, 4, 0, "NOISY3", 5), (512, 384, 3, 0, "NOISY3", 5),
                                  (384, 256, 8, 2, "LANDSAT16", BASE), (384, 256, 4, 3, "DEM", FTL), (384, 256, 1, 2, "LANDSAT16", BASE), (384, 256, 3, 2, "LANDSAT16", FTL),
                                  (384, 256, 1, 3, "DEM", 5), (384, 256, 1, 2, "LANDSAT16", 7), (320, 256, 1, 5, "DEM", FTL), (320, 256, 1, 7, "DEM", BASE),
                                  (320, 256, 1, 5, "DEM", 5), (320, 256, 1, 6, "TERRACE", 5), (320, 256, 5, 0, "NOISY3", FTL), (320, 256, 2, 5, "DEM", 5)],
                         ids=lambda c: "%dx%dx%d-t%d-%s-m%d" % c)
def test_self_indexed_containers_decode_from_their_table(qb3, oracle, case):
    """a container written with its restart table decodes FROM the table: the decode reports status 0 -- no fallback to the walk
    (bit 5 says the table was dropped: a table the decoder cannot use costs time silently otherwise, as the 16-bit common-factor
    table did while two fill kernels wrote it), for every raster shape that gets a table, at both levels"""
    import torch
    from qb3_amd import device as qdev, synth
    w, h, b, dt, gen, mode = case
    img = synth.generate(w, h, b, dt, gen, 12)
    raw = img.reshape(-1).view(torch.uint8)
    cb = None if b in (1, 3, 4) else list(range(b))
    for level in (1, 2):
        enc = qdev.DeviceEncoder(w, h, b, dt, mode=mode, cband=cb, index_chunk=level, want_index=False)
        dst, n, _ = enc.encode(img)
        if int(dst[10]) == 255:
            continue                                    # raw-stored: no table (a winning RLE0 pass keeps it)
        dec = qdev.DeviceDecoder(dst, n)
        assert qb3.lib.qb3x_decoder_table_entries(dec.p) > 0, level
        out = dec.decode(dst, index=None)
        assert torch.equal(out.view(torch.uint8), raw), level
        assert qb3.lib.qb3x_last_decode_status(dec.p) == 0, (level, qb3.lib.qb3x_last_decode_status(dec.p))


def test_pipelined_host_calls_from_two_threads(qb3, oracle):
    """the strip pipeline (three streams, two rings of pinned slices a handle, ONE pool of copy threads a process) under two
    host threads at once, each with its own handles and a raster large enough for it: streams equal the oracle's, pixels exact"""
    import threading
    cases = [(8192, 4099, 3, 0, "NOISY3", FTL), (8192, 4100, 1, 5, "DEM", BASE)]
    imgs = [oracle.generate(w, h, b, dt, gen, 40 + i) for i, (w, h, b, dt, gen, m) in enumerate(cases)]
    refs = [oracle.encode(img, c[3], c[5]) for img, c in zip(imgs, cases)]
    errors = []

    def work(i):
        w, h, b, dt, gen, mode = cases[i]
        try:
            for _ in range(3):
                got = qb3.encode(imgs[i], dt, mode)
                if not np.array_equal(got, refs[i]):
                    errors.append("thread %d: stream differs" % i)
                    return
                two = qb3.encode(imgs[i], dt, mode, index_chunk=2)
                out, _, _, _ = qb3.decode(two)
                if not np.array_equal(out, imgs[i].view(np.uint8).ravel()):
                    errors.append("thread %d: pixels differ" % i)
                    return
        except Exception as e:      # noqa: BLE001 -- report whatever a worker thread hit
            errors.append("thread %d: %r" % (i, e))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(cases))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    qb3.lib.qb3x_trim()                                 # (the rings and pooled buffers of the handles that went)


@pytest.mark.parametrize("case", [(1024, 1024, 3, 0, "NOISY3", 4), (509, 259, 1, 0, "GRAD", 8), (1024, 768, 8, 2, "LANDSAT16", 4),
                                  (300, 200, 2, 2, "LANDSAT16", 8), (512, 512, 1, 4, "DEM", 8), (256, 256, 1, 7, "DEM", 4)])
def test_restart_table_does_not_depend_on_the_index_request(qb3, oracle, case):
    """a self-indexed container is the same bytes whether the caller asks for the out-of-band index or not (without one the
    library keeps an index of its own that holds segment entries only -- no unit lengths), and both decode from the container
    alone; minus the table chunks they are the reference's container"""
    import torch
    from qb3_amd import synth, device as qdev
    w, h, b, dt, gen, mode = case
    img = synth.generate(w, h, b, dt, gen, 21)
    cb = None if b in (1, 3, 4) else list(range(b))
    e0 = qdev.DeviceEncoder(w, h, b, dt, mode=mode, cband=cb, want_index=False, index_chunk=True)
    e1 = qdev.DeviceEncoder(w, h, b, dt, mode=mode, cband=cb, want_index=True, index_chunk=True)
    d0, n0, _ = e0.encode(img)
    d1, n1, index = e1.encode(img)
    assert n0 == n1 and torch.equal(d0[:n0], d1[:n1])
    host = d0[:n0].cpu().numpy()
    assert bytes(host.tobytes()).find(b"ix", 11) > 0
    raw = img.reshape(-1).view(torch.uint8)
    dec = qdev.DeviceDecoder(d0, n0)
    assert torch.equal(dec.decode(d0, index=None), raw)
    assert torch.equal(dec.decode(d0, index=index), raw)
    ref = oracle.encode(oracle.generate(w, h, b, dt, gen, 21), dt, mode, cband=cb)
    extra, dt_at = n0 - len(ref), bytes(ref).index(b"DT", 11)
    assert bytes(host[:dt_at]) == bytes(ref[:dt_at]) and bytes(host[dt_at + extra:]) == bytes(ref[dt_at:])


BL_CASES = [(256, 256, 3, 0, "NOISY3", FTL), (509, 259, 3, 0, "NOISY3", BASE), (768, 512, 1, 0, "GRAD", FTL), (333, 77, 4, 0, "NOISY3", BASE),
            (1024, 1024, 3, 0, "NOISY3", 0), (2048, 1536, 1, 0, "NOISY3", FTL), (4096, 2048, 3, 0, "NOISY3", FTL), (64, 16, 3, 0, "RANDOM", FTL),
            # 16-bit, eight and four bands: two band-pair lengths per lane of the decoder's wave
            (256, 256, 8, 2, "LANDSAT16", BASE), (509, 259, 8, 2, "LANDSAT16", FTL), (300, 200, 4, 2, "LANDSAT16", BASE), (640, 384, 8, 3, "GRAD", FTL),
            (2048, 1024, 8, 2, "LANDSAT16", 0), (333, 77, 4, 3, "DEM", FTL),
            (768, 512, 1, 2, "LANDSAT16", FTL), (509, 259, 1, 3, "DEM", BASE), (2048, 1024, 1, 2, "DEM", 0),     # one band: a field per lane
            (300, 200, 3, 2, "LANDSAT16", FTL), (256, 256, 2, 3, "DEM", BASE), (320, 240, 6, 2, "LANDSAT16", FTL), (1024, 512, 3, 2, "LANDSAT16", 0),
            # 32/64-bit: a twelve-bit length per unit, the unit-parallel decoder
            (256, 256, 1, 4, "DEM", FTL), (509, 259, 1, 5, "DEM", BASE), (300, 200, 3, 4, "DEM", FTL), (256, 256, 1, 6, "RUNG63", FTL),
            (520, 300, 1, 7, "DEM", BASE), (160, 120, 5, 4, "DEM", FTL), (1024, 1024, 1, 7, "DEM", 0)]


@pytest.mark.parametrize("case", BL_CASES, ids=lambda c: "%dx%dx%d-t%d-%s-m%d" % c)
def test_restart_table_with_block_lengths(qb3, oracle, case, tmp_path):
    """qb3x_set_encoder_index_chunk level 2: the table's entries end with the bit lengths of their segment's blocks (ten bits
    each), and the 8-bit lane-per-block decoder then needs neither a walk nor an index -- ONE kernel.  The container is the
    reference's plus ignorable chunks (which the reference's reader steps over), the host flavour decodes it too, the walker
    still can (QB3_NO_BLOCK_LENGTHS), and lengths that are not the stream's are reported, not followed"""
    import subprocess
    import sys
    import ctypes as C
    import torch
    from qb3_amd import synth, device as qdev
    w, h, b, dt, gen, mode = case
    cb = None if b in (1, 3, 4) else list(range(b))
    himg = oracle.generate(w, h, b, dt, gen, 31)
    img = torch.from_numpy(himg.view(np.uint8).copy()).cuda().view(-1)
    raw = img.reshape(-1).view(torch.uint8)
    ref = oracle.encode(himg, dt, mode, cband=cb)
    enc = qdev.DeviceEncoder(w, h, b, dt, mode=mode, cband=cb, want_index=False, index_chunk=2)
    dst, n, _ = enc.encode(img)
    host = dst[:n].cpu().numpy()
    if ref[10] == 255:                  # raw-stored: no table at all
        assert n == len(ref) and np.array_equal(host, ref)
        return
    extra, dt_at = n - len(ref), bytes(ref).index(b"DT", 11)
    assert bytes(host[:dt_at]) == bytes(ref[:dt_at]) and bytes(host[dt_at + extra:]) == bytes(ref[dt_at:])
    _, seen = walk_chunks(host, False)
    mine = [c for c in seen if c[1] >= dt_at]
    nblocks = ((w + 3) // 4) * ((h + 3) // 4)
    bg16 = b if b <= 4 else (4 if b % 4 == 0 else 2)    # 16-bit: bands a lane of the decoder owns
    per_seg = 64 if dt == 0 else 64 // (b // bg16)      # blocks of a decoder wave (8- and 16-bit data)
    assert mine[0][0] == b"ix" and host[mine[0][1] + 5] & 2, "entries are flagged as carrying block lengths"
    if dt <= 3:
        lens_bytes = 80 if (dt == 0 or b == 1) else 160
        nseg, entry = (nblocks + per_seg - 1) // per_seg, 6 + (2 if dt == 0 else 3) * b + lens_bytes
        assert sum(c[2] - 12 for c in mine if c[0] == b"ix") == nseg * entry
    else:                               # blocks per entry are in the chunk head; a twelve-bit field per unit
        per_seg = int.from_bytes(bytes(host[mine[0][1] + 8:mine[0][1] + 12]), "little")
        lens_bytes = (per_seg * b * 12 + 7) // 8
        entry = 6 + b * (1 + oracle.TYPESIZE[dt]) + lens_bytes
        assert sum(c[2] - 12 for c in mine if c[0] == b"ix") == ((nblocks + per_seg - 1) // per_seg) * entry
    out, _, _, _ = oracle.decode(host, identity=True)
    assert out is not None and np.array_equal(out, raw.cpu().numpy()), "the reference decoder must step over the chunks"
    L = qb3.lib
    L.qb3x_profile_enable(1); L.qb3x_profile_reset()
    out, _, _, _ = qb3.decode(host)
    hbuf = C.create_string_buffer(1024)
    L.qb3x_profile_names(hbuf, 1024)
    L.qb3x_profile_enable(0)
    assert np.array_equal(out, raw.cpu().numpy())
    assert hbuf.value == b"dec_units", hbuf.value        # qb3_read_data on host buffers: the table was used (its check passed), no walk
    got = qb3.encode(himg, dt, mode, cband=cb, index_chunk=2)
    assert np.array_equal(got, host), "host and device flavour write the same container"
    L = qb3.lib
    L.qb3x_profile_enable(1); L.qb3x_profile_reset()
    dec = qdev.DeviceDecoder(dst, n)
    if cb is not None:
        L.qb3x_set_decoder_compat(dec.p, 0)
    res = dec.decode(dst, index=None)
    torch.cuda.synchronize()
    buf = C.create_string_buffer(1024)
    L.qb3x_profile_names(buf, 1024)
    L.qb3x_profile_enable(0)
    assert torch.equal(res, raw)
    assert buf.value == b"dec_units", buf.value          # no walk, no index rebuild
    # lengths that are not the stream's: the chunk's check fails and the decoder walks the stream instead -- the right pixels
    at = mine[0][1] + 12 + entry - lens_bytes
    bad = dst.clone()
    bad[at + 1] ^= 0x5a
    res = qdev.DeviceDecoder(bad, n).decode(bad, index=None)
    assert torch.equal(res, raw)
    assert torch.equal(dec.decode(dst, index=None), raw)
    if w * h <= 512 * 512:              # ... and any byte of the table: positions, rungs, values, lengths -- no crash, whatever comes out
        rng = np.random.default_rng(w * h + b)
        t0, t1 = mine[0][1], dt_at + extra
        for _ in range(24):
            bad = dst.clone()
            at = int(rng.integers(t0 + 12, t1))
            bad[at] ^= int(rng.integers(1, 256))
            try:
                qdev.DeviceDecoder(bad, n).decode(bad, index=None)
            except (RuntimeError, ValueError):
                pass
        assert torch.equal(dec.decode(dst, index=None), raw)
    # the same container through the walker (the switch is read once per process: a child)
    f = tmp_path / "c.qb3"
    f.write_bytes(bytes(host))
    code = """
import sys, numpy as np, torch
sys.path.insert(0, %r)
import qb3_amd
from qb3_amd import synth, device as qdev
c = torch.from_numpy(np.fromfile(%r, dtype=np.uint8)).cuda()
from oracle import pyoracle as o
raw = torch.from_numpy(o.generate(%d, %d, %d, %d, %r, 31).view(np.uint8).copy()).cuda().view(-1)
d = qdev.DeviceDecoder(c, c.numel())
qb3_amd.lib.qb3x_set_decoder_compat(d.p, 0)
assert torch.equal(d.decode(c, index=None), raw)
print("ok")
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), str(f), w, h, b, dt, gen)
    env = dict(os.environ)
    env["QB3_NO_BLOCK_LENGTHS"] = "1"
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_block_lengths_only_where_the_decoder_uses_them(qb3, oracle):
    """Level 2 tables carry what the raster's decoder places its units by.  FTL / BASE rasters of the lane-per-unit decoder
    (k_dec_pxu.hip: 8-bit rasters of 2 or more than 4 bands, 16-bit of an odd band count above 4, 32/64-bit of several bands): a
    twelve-bit length per unit, so level 2 is larger than level 1 and both decode from the container alone.  A common-factor
    stream has a field per block (8-bit grey / RGB / RGBA; one band of wider values) or per unit (everything else: the
    lane-per-unit decoder) at EITHER level -- its bits and the rung it is entered with -- because that is what its decoder
    works from: levels 1 and 2 are the same container."""
    for (w, h, b, dt, gen, mode) in [(256, 128, 5, 2, "LANDSAT16", FTL), (160, 120, 5, 0, "NOISY3", FTL), (96, 200, 2, 0, "NOISY3", 4), (130, 70, 3, 5, "DEM", 4),
                                     (64, 64, 2, 7, "DEM", FTL), (100, 52, 16, 0, "NOISY3", FTL), (77, 41, 7, 2, "LANDSAT16", 4)]:
        img = oracle.generate(w, h, b, dt, gen, 3)
        cb = None if b in (1, 3, 4) else list(range(b))
        ref = oracle.encode(img, dt, mode, cband=cb)
        one = qb3.encode(img, dt, mode, cband=cb, index_chunk=1)
        two = qb3.encode(img, dt, mode, cband=cb, index_chunk=2)
        assert len(two) > len(one) > len(ref), (w, h, b, dt)
        for c in (one, two):
            extra, dt_at = len(c) - len(ref), bytes(ref).index(b"DT", 11)
            assert bytes(c[:dt_at]) == bytes(ref[:dt_at]) and bytes(c[dt_at + extra:]) == bytes(ref[dt_at:])
            out, _, _, _ = qb3.decode(c)
            assert np.array_equal(out, img.view(np.uint8).ravel()), (w, h, b, dt, gen, mode)
        want, _, _, _ = oracle.decode(two, identity=True)
        assert want is not None and np.array_equal(want, img.view(np.uint8).ravel()), "the reference decoder must step over the chunks"
    for (w, h, b, dt, gen, mode) in [(256, 256, 3, 0, "NOISY3", 7), (300, 200, 1, 0, "PALETTE", 5), (256, 192, 8, 2, "LANDSAT16", 1),
                                     (200, 160, 1, 5, "DEM", 5), (128, 128, 1, 7, "TERRACE", 7), (1024, 1024, 3, 0, "FEW", 5),
                                     (160, 120, 5, 0, "NOISY3", 5), (128, 96, 2, 5, "DEM", 5), (96, 64, 3, 7, "TERRACE", 5), (200, 100, 2, 0, "PALETTE", 5),
                                     (120, 88, 3, 2, "LANDSAT16", 5), (64, 48, 16, 2, "FEW", 1)]:
        img = oracle.generate(w, h, b, dt, gen, 3)
        cb = None if b in (1, 3, 4) else list(range(b))
        ref = oracle.encode(img, dt, mode, cband=cb)
        one = qb3.encode(img, dt, mode, cband=cb, index_chunk=1)
        two = qb3.encode(img, dt, mode, cband=cb, index_chunk=2)
        if ref[10] == 255:                          # raw-stored: no table either way
            assert np.array_equal(one, ref) and np.array_equal(two, ref)
            continue
        assert np.array_equal(one, two) and len(one) > len(ref), (w, h, b, dt, gen, mode)
        extra, dt_at = len(two) - len(ref), bytes(ref).index(b"DT", 11)
        assert bytes(two[:dt_at]) == bytes(ref[:dt_at]) and bytes(two[dt_at + extra:]) == bytes(ref[dt_at:])
        want, _, _, _ = oracle.decode(two, identity=True)
        assert want is not None and np.array_equal(want, img.view(np.uint8).ravel()), "the reference decoder must step over the chunks"
        out, _, _, _ = qb3.decode(two)
        assert np.array_equal(out, img.view(np.uint8).ravel()), (w, h, b, dt, gen, mode)


def test_tiles_with_block_length_tables(qb3, oracle):
    """a batch of tiles whose containers carry level 2 tables: decoded from the containers alone, one kernel"""
    import torch
    from qb3_amd import synth, device as qdev
    w, h, n = 512, 384, 6
    imgs = torch.stack([synth.generate(w, h, 3, 0, "NOISY3", 700 + t) for t in range(n)])
    tc = qdev.TileBatchCoder(w, h, 3, 0, n, want_index=False, index_chunk=2)
    tc.encode(imgs)
    host = tc.dst.cpu().numpy()
    for t in range(n):
        ref = oracle.encode(imgs[t].cpu().numpy(), 0, FTL)
        c = host[t * tc.pitch:t * tc.pitch + tc.sizes[t]]
        extra, dt_at = len(c) - len(ref), bytes(ref).index(b"DT", 11)
        assert bytes(c[:dt_at]) == bytes(ref[:dt_at]) and bytes(c[dt_at + extra:]) == bytes(ref[dt_at:]), t
    out = torch.zeros_like(imgs)
    L = qb3.lib
    L.qb3x_profile_enable(1); L.qb3x_profile_reset()
    tc.decode(out, use_index=False)
    torch.cuda.synchronize()
    import ctypes as C
    buf = C.create_string_buffer(1024)
    L.qb3x_profile_names(buf, 1024)
    L.qb3x_profile_enable(0)
    assert torch.equal(out, imgs) and buf.value == b"dec_units", buf.value


def test_common_factor_two_pass_coding_on_small_rasters(qb3, oracle):
    """The common-factor encoders sample a raster before they code it only from 32768 chunks on (below that coding once and
    again where needed costs no more than the sample would save); QB3_BEST_SAMPLE_MIN=0 makes them sample always, which is how
    the two-pass coding -- data whose every unit brings a factor: all values scaled by three -- stays tested on small rasters,
    for the 8-bit lane-per-block kernel, the unit-per-lane kernel and the 32-bit lane-per-block front end.  (A child process:
    the switch is read once.)"""
    import subprocess
    import sys
    code = """
import sys, numpy as np
sys.path.insert(0, %r)
import qb3_amd
from oracle import pyoracle as o
for (w, h, b, dt, gen, mode) in [(1024, 768, 3, 0, "NOISY3", 5), (1024, 768, 3, 0, "SCALED", 5), (512, 512, 2, 2, "SCALED", 7), (1024, 512, 1, 5, "SCALED", 5),
                                 (1024, 512, 1, 5, "DEM", 7), (640, 480, 1, 3, "SCALED", 5), (512, 512, 1, 7, "SCALED", 1)]:
    if gen == "SCALED":
        base = o.generate(w, h, b, dt, "NOISY3" if dt < 4 else "DEM", 9)
        img = ((base.astype(np.int64) // 3) * 3).astype(base.dtype)
    else:
        img = o.generate(w, h, b, dt, gen, 9)
    cb = None if b in (1, 3, 4) else list(range(b))
    ref = o.encode(img, dt, mode, cband=cb)
    for rep in range(2):
        got = qb3_amd.encode(img, dt, mode, cband=cb)
        assert len(got) == len(ref) and np.array_equal(got, ref), (w, h, b, dt, gen, mode, rep)
    out, _, _, _ = qb3_amd.decode(got)
    assert np.array_equal(out, img.view(np.uint8).ravel()), (w, h, b, dt, gen, mode)
print("ok")
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for val in ("0", None):
        env = dict(os.environ)
        if val is not None:
            env["QB3_BEST_SAMPLE_MIN"] = val
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "ok" in r.stdout, (val, r.stdout[-2000:] + r.stderr[-4000:])


@pytest.mark.parametrize("switch", ["QB3_NO_PX", "QB3_SLOW_INDEX", "QB3_SLOW_WALK", "QB3_WALK_TAB_KB=2048"])
def test_alternative_kernel_paths(qb3, oracle, switch, tmp_path):
    """the paths that are not the default -- the generic
    kernels on rasters the lane-per-block kernels would take, the one-lane index rebuild, the one-wave walk of a plain
    stream, the table walk in many rounds (a table of 2 MiB) -- give the same bytes and pixels.
    (The switches are read once per process: a child process each.)"""
    import subprocess
    import sys
    code = """
import sys, numpy as np, torch
sys.path.insert(0, %r)
import qb3_amd
from qb3_amd import synth, device as qdev
from oracle import pyoracle as o
for (w, h, b, dt, gen, seed, mode) in [(1024, 1024, 3, 0, "NOISY3", 2, 8), (509, 259, 3, 0, "NOISY3", 1, 4), (768, 512, 1, 0, "GRAD", 0, 8),
                                       (640, 384, 4, 0, "RANDOM", 5, 8), (512, 256, 8, 2, "LANDSAT16", 3, 4), (1024, 768, 8, 2, "LANDSAT16", 4, 8),
                                       (700, 300, 1, 2, "LANDSAT16", 5, 8), (600, 400, 2, 2, "LANDSAT16", 6, 4), (512, 512, 6, 2, "LANDSAT16", 7, 8)]:
    img = synth.generate(w, h, b, dt, gen, seed)
    ref = o.encode(o.generate(w, h, b, dt, gen, seed), dt, mode)
    enc = qdev.DeviceEncoder(w, h, b, dt, mode=mode)
    for rep in range(3):
        dst, n, index = enc.encode(img)
        assert n == len(ref) and np.array_equal(dst[:n].cpu().numpy(), ref), (w, h, b, rep)
    dec = qdev.DeviceDecoder(dst, n)
    raw = img.reshape(-1).view(torch.uint8)
    if ref[10] != 255:
        assert torch.equal(dec.decode(dst, index=index), raw) and torch.equal(dec.decode(dst, index=None), raw)
# a batch of tiles through the same paths
import ctypes as C
L = qb3_amd.lib
w, h, b, n = 256, 192, 3, 5
imgs = torch.stack([synth.generate(w, h, b, 0, "NOISY3", 3000 + t) for t in range(n)])
p = L.qb3_create_encoder(w, h, b, 0)
pitch = (L.qb3_max_encoded_size(p) + 3) // 4 * 4
dst = torch.zeros(n * pitch, dtype=torch.uint8, device="cuda")
sizes = (C.c_size_t * n)()
assert L.qb3x_encode_tiles(p, imgs.data_ptr(), n, w * h * b, dst.data_ptr(), pitch, None, sizes, None) == n
host = dst.cpu().numpy()
for t in range(n):
    ref = o.encode(imgs[t].cpu().numpy(), 0, 8)
    assert sizes[t] == len(ref) and np.array_equal(host[t * pitch:t * pitch + sizes[t]], ref), t
dims = (C.c_size_t * 3)()
hdr = host[:64].copy()
d = L.qb3_read_start(hdr.ctypes.data, sizes[0], dims)
assert L.qb3_read_info(d)
out = torch.zeros_like(imgs)
assert L.qb3x_decode_tiles(d, dst.data_ptr(), n, pitch, sizes, out.data_ptr(), w * h * b, None, None) == n     # the streams alone
assert torch.equal(out, imgs)
print("ok")
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    name, _, value = switch.partition("=")
    env[name] = value or "1"
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("case", [(1100, 700, 3, 0, 256, []), (1024, 512, 1, 0, 512, ["-b"]), (600, 300, 3, 2, 128, ["-f"]), (1100, 700, 3, 0, 256, ["-f", "-i"]),
                                  (1100, 700, 3, 0, 256, ["-f", "-I"])],
                         ids=lambda c: "%dx%dx%d-t%d-tile%d%s" % (c[0], c[1], c[2], c[3], c[4], "".join(c[5])))
def test_tile_batcher_tool(qb3, oracle, tmp_path, case):
    """tools/qb3tiles.cpp, the GDAL-MRF-style caller of qb3x_encode_tiles / qb3x_decode_tiles: a raster cut into tiles,
    every tile the container qb3_encode writes for it (the reference's per-tile calling pattern: README.md:30-31,
    cqb3.cpp:614-641), edge tiles padded by replication, the raster back exactly"""
    import subprocess
    w, h, b, dt, T, flags = case
    tool = os.path.join(os.path.dirname(qb3.LIB_PATH), "qb3tiles")
    gen = "NOISY3" if dt == 0 else "LANDSAT16"
    img = oracle.generate(w, h, b, dt, gen, 7)
    pnm = tmp_path / "in.pnm"
    body = img.astype(">u2").tobytes() if dt == 2 else img.tobytes()
    pnm.write_bytes(b"P%c\n%d %d\n%d\n" % (b"56"[b == 3], w, h, 255 if dt == 0 else 65535) + body)
    r = subprocess.run([tool, "-e", "-v", "-t", str(T), *flags, str(pnm), str(tmp_path / "out.qts")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    f = np.fromfile(tmp_path / "out.qts", dtype=np.uint8)
    assert bytes(f[:4]) == b"QTS1"
    hw = f[4:32].view(np.uint32)
    assert list(hw[:5]) == [w, h, b, dt, T]
    tx, ty = int(hw[5]), int(hw[6])
    assert (tx, ty) == ((w + T - 1) // T, (h + T - 1) // T)
    tab = f[32:32 + 16 * tx * ty].view(np.uint64).reshape(-1, 2)
    mode = 7 if "-b" in flags else 8
    for k in (0, tx - 1, tx * ty - 1):          # an inner tile, the last column (padded), the last tile (padded both ways)
        j, i = divmod(k, tx)
        ys = np.minimum(np.arange(j * T, j * T + T), h - 1)
        xs = np.minimum(np.arange(i * T, i * T + T), w - 1)
        tile = np.ascontiguousarray(img[ys][:, xs])
        want = oracle.encode(tile, dt, mode)
        off, size = int(tab[k, 0]), int(tab[k, 1])
        one = f[off:off + size]
        if "-i" in flags or "-I" in flags:       # the reference's container with the restart table's chunks in front of "DT"
            extra, dt_at = size - len(want), bytes(want).index(b"DT", 11)
            assert extra > 0 and bytes(one[dt_at:dt_at + 2]) == b"ix", "tile %d" % k
            assert bytes(one[:dt_at]) == bytes(want[:dt_at]) and bytes(one[dt_at + extra:]) == bytes(want[dt_at:]), "tile %d" % k
        else:
            assert size == len(want) and np.array_equal(one, want), "tile %d" % k
        r = subprocess.run([tool, "-x", str(tmp_path / "out.qts"), str(k), str(tmp_path / "t.qb3")], capture_output=True, text=True, timeout=60)
        assert r.returncode == 0 and np.array_equal(np.fromfile(tmp_path / "t.qb3", dtype=np.uint8), one)
    r = subprocess.run([tool, "-d", "-v", str(tmp_path / "out.qts"), str(tmp_path / "back.pnm")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    back = (tmp_path / "back.pnm").read_bytes()
    hdr = b"P%c\n%d %d\n%d\n" % (b"56"[b == 3], w, h, 255 if dt == 0 else 65535)
    assert back.startswith(hdr)
    px = np.frombuffer(back[len(hdr):], dtype=np.uint8)
    if dt == 2:
        px = px.view(">u2").astype("<u2").view(np.uint8)
    assert np.array_equal(px, img.view(np.uint8).ravel())


def test_config5_one_rank_batch(qb3, oracle):
    """BASELINE.json configs[4], one rank's share: 32 tiles of 4096 x 4096 x 3 (NOISY3, seeds 1000..1031) through ONE
    qb3x_encode_tiles call and ONE qb3x_decode_tiles call; tiles 1000 and 1001 are the reference's containers byte for
    byte (size and FNV-1a64 of SURVEY.md Appendix C), every tile comes back exactly, with the index and without"""
    import torch
    from qb3_amd import synth, device as qdev
    rows = {a["seed"]: a for a in anchors() if a["cfg"] == "5"}
    n, w = 32, 4096
    imgs = torch.stack([synth.generate(w, w, 3, 0, "NOISY3", 1000 + t) for t in range(n)])
    tc = qdev.TileBatchCoder(w, w, 3, 0, n)
    sizes = tc.encode(imgs)
    for t in (0, 1):
        a = rows[1000 + t]
        host = tc.dst[t * tc.pitch:t * tc.pitch + sizes[t]].cpu().numpy()
        assert sizes[t] == a["size"] and qb3.fnv(host) == a["fnv_stream"]
    assert len(set(sizes)) > 8                  # thirty-two different streams, not one stream thirty-two times
    out = torch.empty_like(imgs)
    for use_index in (True, False):
        out.zero_()
        tc.decode(out, use_index=use_index)
        assert torch.equal(out, imgs)


@pytest.mark.parametrize("mode", [4, 7], ids=["BASE", "BEST"])
@pytest.mark.parametrize("away", [False, True])
@pytest.mark.parametrize("q", [2, 3, 4, 10])
def test_quanta_all_modes(qb3, oracle, q, away, mode):
    """the reference's own matrix: quanta 2/3/4/10 x away x BASE/BEST on 8-bit data (test_qb3.cpp:643-660): same
    container as the CPU path, decode within q/2 of the input"""
    img = oracle.generate(253, 131, 3, 0, "NOISY3", 11)
    want = oracle.encode(img, 0, mode, quanta=q, away=away)
    got = qb3.encode(img, 0, mode, quanta=q, away=away)
    assert np.array_equal(got, want)
    out, dims, _, _ = qb3.decode(got)
    ref, _, _, _ = oracle.decode(want, identity=True)
    assert np.array_equal(out, ref)
    err = np.abs(out.astype(np.int32) - img.ravel().astype(np.int32))
    assert err.max() <= q // 2 + (q % 2 == 0 and not away) * 0 + 1 and (np.abs(out.astype(np.int32) - np.clip(img.ravel().astype(np.int32), 0, (255 // q) * q + q)) <= q).all()


@pytest.mark.parametrize("case", [(2, 5), (2, 1 << 8), (5, 5), (5, 1 << 8), (5, 1 << 24), (7, 5), (7, 1 << 24), (7, 1 << 56), (6, 1 << 56)],
                         ids=lambda c: "t%d-x%d" % c)
def test_common_factor_data_in_best_mode(qb3, oracle, case):
    """the reference's "common factor" inputs (test_qb3.cpp:675-686): values multiplied by 5, 2^8, 2^24, 2^56 on 16-,
    32- and 64-bit types, QB3M_BEST -- same container as the CPU path, exact round trip, device and host flavour"""
    import torch
    from qb3_amd import device as qdev
    dt, mul = case
    base = oracle.generate(192, 160, 1, 0, "NOISY3", 21).astype(np.int64)       # small values: the product fits the type
    img = (base * mul).astype(oracle.NPTYPE[dt])
    for mode in (7, 5):
        want = oracle.encode(img, dt, mode)
        got = qb3.encode(img, dt, mode)
        assert np.array_equal(got, want)
        out, _, _, _ = qb3.decode(got)
        assert np.array_equal(out, img.view(np.uint8).ravel())
    dimg = torch.from_numpy(img.view(np.uint8)).cuda()
    enc = qdev.DeviceEncoder(192, 160, 1, dt, mode=7)
    dst, n, index = enc.encode(dimg)
    assert n == len(want if False else oracle.encode(img, dt, 7)) and np.array_equal(dst[:n].cpu().numpy(), oracle.encode(img, dt, 7))
    dec = qdev.DeviceDecoder(dst, n)
    assert torch.equal(dec.decode(dst, index=index), dimg.reshape(-1)) and torch.equal(dec.decode(dst, index=None), dimg.reshape(-1))


def _cf_rasters(w, h, b, seed):
    """8-bit rasters that exercise every form of a common-factor stream: values scaled by 3 and by 16 (a factor in every
    unit, written once then "same as before"), a handful of distinct values (index form), and a patchwork of the three
    with noise between them (the factor state changes hands inside chunks and across them)"""
    rng = np.random.default_rng(seed)
    def scaled(k, hh=h, ww=w):
        return (rng.integers(0, 256 // k, size=(hh, ww, b), dtype=np.uint8) * k).astype(np.uint8)
    def few(hh=h, ww=w, n=5):
        return rng.integers(0, 256, size=n, dtype=np.uint8)[rng.integers(0, n, size=(hh, ww, b))]
    out = {"scaled3": scaled(3), "scaled16": scaled(16), "few": few()}
    mixed = rng.integers(0, 256, size=(h, w, b), dtype=np.uint8) // 8 + np.arange(w, dtype=np.uint8)[None, :, None]
    mixed[: h // 3] = scaled(6, h // 3)
    mixed[h // 3: h // 2, : w // 2] = few(h // 2 - h // 3, w // 2)
    mixed[h // 2:, w // 2:] = (mixed[h // 2:, w // 2:] // 4) * 4
    out["mixed"] = np.ascontiguousarray(mixed)
    return out


@pytest.mark.parametrize("mode", [1, 5, 7])
@pytest.mark.parametrize("shape", [(64, 48, 3), (509, 259, 3), (1024, 768, 1), (131, 77, 4), (1028, 260, 4), (2048, 520, 3)],
                         ids=lambda s: "%dx%dx%d" % s)
def test_common_factor_8bit_lane_per_block(qb3, oracle, shape, mode):
    """QB3M_BEST family on 8-bit grey / RGB / RGBA through the lane-per-block kernels (k_enc_px_best.hip, k_dec_px_best.hip):
    the container equals the oracle's for data with factors everywhere, index-form data and mixtures; it decodes with the
    out-of-band index (segment entries + a dword per block), from a plain container (index rebuilt), and from a self-indexed
    container alone (a field per block in the table), which the reference's reader steps over
    (reference QB3encode.h:283-361,557-724; QB3decode.h:578-741)"""
    import torch
    from qb3_amd import device as qdev
    w, h, b = shape
    for name, host in _cf_rasters(w, h, b, 11 * w + b).items():
        for cband in ([None, list(range(b))] if b >= 3 else [None]):
            ref = oracle.encode(host, 0, mode, cband=cband)
            img = torch.from_numpy(host).cuda()
            enc = qdev.DeviceEncoder(w, h, b, 0, mode=mode, cband=cband)
            dst, n, index = enc.encode(img)
            got = dst[:n].cpu().numpy()
            assert n == len(ref) and np.array_equal(got, ref), (name, cband, n, len(ref), first_diff(got, ref))
            if ref[10] in (255, 2, 3, 6, 7):
                continue
            dec = qdev.DeviceDecoder(dst, n)
            assert torch.equal(dec.decode(dst, index=index), img.reshape(-1)), (name, cband, "indexed")
            assert torch.equal(dec.decode(dst, index=None), img.reshape(-1)), (name, cband, "plain")
            if cband is None:
                enc2 = qdev.DeviceEncoder(w, h, b, 0, mode=mode, want_index=False, index_chunk=1)
                d2, n2, _ = enc2.encode(img)
                c2 = d2[:n2].cpu().numpy()
                want, _, _, _ = oracle.decode(c2, identity=False)
                assert n2 > n and want is not None and np.array_equal(want, host.ravel()), (name, "the reference's reader and the table")
                assert torch.equal(qdev.DeviceDecoder(d2, n2).decode(d2, index=None), img.reshape(-1)), (name, "container alone")


@pytest.mark.parametrize("device_flavour", [False, True], ids=["host", "device"])
@pytest.mark.parametrize("case", [(0, 3, False, FTL, 3), (0, 3, False, BASE, 1), (1, 10, True, FTL, 4), (2, 5, False, BASE, 3)],
                         ids=lambda c: "t%d-q%d%s-m%d-b%d" % (c[0], c[1], "away" if c[2] else "", c[3], c[4]))
def test_stride_decode_with_quanta(qb3, oracle, case, device_flavour):
    """the reference's "Stride decoding and quanta" row (reference test_qb3.cpp:659-660, check_stride_decode :291-395): a
    quantised container (QV chunk) decoded into a destination whose lines are `stride` values apart -- through qb3_read_data
    on host buffers and through qb3x_decode_device on device buffers.  The pixels equal the oracle's decode of the same
    container, lie within quanta / 2 of the input (the reference's acceptance, :366-375), and nothing is written between the
    lines."""
    import ctypes as C
    dt, q, away, mode, b = case
    w, h = 131, 77
    tsz = oracle.TYPESIZE[dt]
    img = oracle.generate(w, h, b, dt, "NOISY3" if dt == 0 else "LANDSAT16" if dt == 2 else "GRAD", 5)
    ref = oracle.encode(img, dt, mode, quanta=q, away=away)
    got = qb3.encode(img, dt, mode, quanta=q, away=away)
    assert np.array_equal(got, ref)
    want, _, _, _ = oracle.decode(ref, identity=True)
    want = want.view(img.dtype).reshape(h, w * b)
    stride = w * b + 9
    L = qb3.lib
    dims = (C.c_size_t * 3)()
    d = L.qb3_read_start(ref.ctypes.data, ref.size, dims)
    assert d and L.qb3_read_info(d) and L.qb3_get_quanta(d) == q
    L.qb3_set_decoder_stride(d, stride)
    fill = 0x5a
    if device_flavour:
        import torch
        dev_in = torch.from_numpy(ref).cuda()
        dev_out = torch.full((h * stride * tsz,), fill, dtype=torch.uint8, device="cuda")
        n = L.qb3x_decode_device(d, dev_in.data_ptr(), dev_out.data_ptr(), None, None)
        out = dev_out.cpu().numpy().view(img.dtype).reshape(h, stride)
    else:
        out = np.full(h * stride * tsz, fill, np.uint8)
        n = L.qb3_read_data(d, out.ctypes.data)
        out = out.view(img.dtype).reshape(h, stride)
    L.qb3_destroy_decoder(d)
    assert n == img.nbytes
    assert np.array_equal(out[:, :w * b], want), "pixels differ from the oracle's decode"
    assert (out[:, w * b:].view(np.uint8) == fill).all(), "the decoder wrote between the lines"
    err = np.abs(out[:, :w * b].astype(np.int64) - img.reshape(h, w * b).astype(np.int64))
    assert err.max() <= q // 2 + (1 if away and q % 2 == 0 else 0)


@pytest.mark.parametrize("case", [(2, 1 << 8), (4, 1 << 24), (6, 1 << 56)], ids=["u16", "u32", "u64"])
@pytest.mark.parametrize("mode", [BASE, FTL])
def test_large_rung_base(qb3, oracle, case, mode):
    """the reference's "Large rung" rows (reference test_qb3.cpp:689-693: check<uint64_t>(.., 1 << 56, 1, fast),
    check<uint32_t>(.., 1 << 24, ..), check<uint16_t>(.., 1 << 8, ..)): 8-bit image data scaled into the top of the value
    range, so that the rungs sit just below the type's width -- computed codes for every value (QB3encode.h:248-277), the
    rung-switch codes of the wide types.  Encode equals the oracle's; decode with the index, from the plain container and
    from the self-indexed container reproduces the input."""
    import torch
    from qb3_amd import device as qdev
    dt, mul = case
    w, h, b = 260, 132, 3
    base = oracle.generate(w, h, b, 0, "NOISY3", 6).astype(np.uint64)
    img = (base * np.uint64(mul)).astype(oracle.NPTYPE[dt])
    ref = check_encode(qb3, oracle, img, dt, mode)
    out, dims, _, _ = qb3.decode(ref)
    assert dims == (w, h, b) and np.array_equal(out, img.view(np.uint8).ravel())
    dimg = torch.from_numpy(img.view(np.uint8).reshape(-1)).cuda()
    enc = qdev.DeviceEncoder(w, h, b, dt, mode=mode)
    dst, n, index = enc.encode(dimg)
    assert n == len(ref) and np.array_equal(dst[:n].cpu().numpy(), ref)
    dec = qdev.DeviceDecoder(dst, n)
    assert torch.equal(dec.decode(dst, index=index), dimg) and torch.equal(dec.decode(dst, index=None), dimg)
    for level in (1, 2):
        c = qb3.encode(img, dt, mode, index_chunk=level)
        o2, _, _, _ = qb3.decode(c)
        assert np.array_equal(o2, img.view(np.uint8).ravel()), level


@pytest.mark.parametrize("case", [(512, 384, 3, 0, "NOISY3", FTL, 2), (512, 384, 3, 0, "NOISY3", 5, 1), (384, 256, 8, 2, "LANDSAT16", BASE, 2),
                                  (320, 256, 1, 5, "DEM", FTL, 2), (320, 256, 1, 7, "DEM", FTL, 2), (320, 256, 1, 5, "DEM", 5, 2), (1024, 1024, 3, 0, "NOISY3", BASE, 1),
                                  (320, 256, 2, 5, "DEM", FTL, 2), (320, 256, 5, 0, "NOISY3", 5, 2), (320, 256, 3, 7, "DEM", BASE, 2)],      # (the lane-per-unit decoders)
                         ids=lambda c: "%dx%dx%d-t%d-%s-m%d-l%d" % c)
def test_a_damaged_restart_table_costs_time_not_pixels(qb3, oracle, case):
    """The table sits in an ignorable chunk the format does not protect, and the decoder takes positions, rungs, entering
    values and lengths from it.  Every "ix" chunk therefore carries a 16-bit check of its entries (version 3), verified on
    the device before use together with the chunk heads; when the check -- or the decode that relied on the table -- fails,
    the stream is decoded again without the table (the plain walk: what the reference, which skips the chunk, does).  Any
    damaged byte of the table: qb3_read_data and qb3x_decode_device still return the right pixels."""
    import torch
    from qb3_amd import device as qdev
    w, h, b, dt, gen, mode, level = case
    img = oracle.generate(w, h, b, dt, gen, 8)
    cb = None if b in (1, 3, 4) else list(range(b))
    good = qb3.encode(img, dt, mode, cband=cb, index_chunk=level)
    raw = img.view(np.uint8).ravel()
    first = bytes(good).index(b"ix", 11)
    dt_at = bytes(good).index(b"DT", first)          # (entries are binary: the first "DT" behind the chunks' start may be inside one)
    ref = oracle.encode(img, dt, mode, cband=cb)
    table_end = len(good) - (len(ref) - bytes(ref).index(b"DT", 11))
    rng = np.random.default_rng(w + h + dt)
    spots = [first + 12, first + 12 + 3, first + 12 + 6, first + 12 + 7 + b, table_end - 9] + [int(x) for x in rng.integers(first + 12, table_end - 6, 8)]
    for at in spots:
        bad = good.copy()
        bad[at] ^= 1 << int(rng.integers(0, 8))
        out, dims, _, _ = qb3.decode(bad)
        assert dims == (w, h, b) and np.array_equal(out, raw), ("host", at - first)
    dbad = torch.from_numpy(good.copy()).cuda()
    dbad[spots[2]] ^= 0x40
    dbad[spots[-1]] ^= 0x01
    dec = qdev.DeviceDecoder(dbad, len(good))
    if cb is not None:
        qb3.lib.qb3x_set_decoder_compat(dec.p, 0)
    assert torch.equal(dec.decode(dbad, index=None), torch.from_numpy(raw).cuda()), "device flavour"


@pytest.mark.parametrize("switch", ["QB3_EXITS_FROM=0", "QB3_WALK_TAB_KB=4096;QB3_EXITS_FROM=0", "QB3_WIDE_BAND=18", "QB3_WIDE_BAND=17", ""],
                         ids=["exits", "exits-in-many-rounds", "super-windows-parsed-by-the-hopping-lane", "chain-through-the-table", "default-by-stream-length"])
def test_plain_rgb_streams_by_exits(qb3, oracle, switch):
    """plain 8-bit RGB streams: the walk by exits with a rung per band in the state (walk_exitB_kernel, walk_exitB_chain_kernel,
    walk_exit_blocks_kernel) -- odd sizes, FTL and BASE, data of every kind, a truncated stream, a batch of tiles; with table
    memory for four super-windows a round; with the cap on distinct exits set so low that every super-window is parsed by
    the hopping lane; and the chain through the table, which other band counts still take -- and short streams: a super-window costs
    milliseconds whatever the stream's length, so below a measured length (QB3_EXITS_FROM overrides it: 0 = always by exits) the chain
    is the quicker walk.  Pixels exact in every case."""
    import subprocess
    import sys
    code = """
import sys, numpy as np, torch, ctypes as C
sys.path.insert(0, %r)
import qb3_amd
from qb3_amd import device as qdev
from oracle import pyoracle as o
L = qb3_amd.lib
for (w, h, gen, mode) in [(1024, 1024, "NOISY3", 8), (1100, 700, "NOISY3", 4), (509, 259, "GRAD", 8), (2048, 300, "PALETTE", 8), (640, 480, "RANDOM", 8),
                          (768, 512, "FEW", 4), (256, 256, "CONST", 8), (4096, 64, "NOISY3", 0),
                          # common-factor streams (QB3M_CF_H, QB3M_BEST): units with the signal code tabulated apart, super-windows in which a unit
                          # brings its own factor parsed by the hopping lane, the block table of the lane-per-block decoder written by the unit lanes
                          (1024, 1024, "NOISY3", 5), (1100, 700, "NOISY3", 7), (509, 259, "GRAD", 5), (768, 512, "PALETTE", 7), (640, 480, "FEW", 5), (512, 512, "SCALED", 5)]:
    if gen == "SCALED":                         # every value a multiple of three: every unit takes the factor the first ones brought
        img = (o.generate(w, h, 3, 0, "NOISY3", 21).astype(np.uint16) // 3 * 3).astype(np.uint8)
    else:
        img = o.generate(w, h, 3, 0, gen, 21)
    ref = o.encode(img, 0, mode)
    d = torch.from_numpy(ref).cuda()
    dec = qdev.DeviceDecoder(d, len(ref))
    out = dec.decode(d, index=None)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), img.view(np.uint8).ravel()), (w, h, gen, mode)
    got, dims, _, _ = qb3_amd.decode(ref)
    assert np.array_equal(got, img.view(np.uint8).ravel()), (w, h, gen, mode, "host")
img = o.generate(1024, 1024, 3, 0, "NOISY3", 22)
ref = o.encode(img, 0, 8)
import random
rng = random.Random(5)
damaged = [np.ascontiguousarray(ref[:cut]) for cut in (len(ref) // 2, len(ref) - 1000)]      # cut short: an error or clamped pixels, never a hang or a fault
for trial in range(12):                             # ... and smashed: bit flips, runs of one byte value, anywhere behind the header
    s = ref.copy()
    if trial & 1:
        for _ in range(rng.randrange(1, 6)):
            at = rng.randrange(64, len(s)); s[at] ^= 1 << rng.randrange(8)
    else:
        at = rng.randrange(64, len(s)); k = min(len(s) - at, rng.randrange(1, 5000)); s[at:at + k] = rng.choice((0, 255, rng.randrange(256)))
    damaged.append(s)
for s in damaged:
    d = torch.from_numpy(s).cuda()
    dims = (C.c_size_t * 3)()
    q = L.qb3x_read_start_device(d.data_ptr(), len(s), dims, None)
    if q:
        out = torch.zeros(img.nbytes, dtype=torch.uint8, device="cuda")
        L.qb3x_decode_device(q, d.data_ptr(), out.data_ptr(), None, None)
        torch.cuda.synchronize()
        L.qb3_destroy_decoder(q)
d = torch.from_numpy(ref).cuda()                    # the GPU is still well: the intact stream decodes
assert np.array_equal(qdev.DeviceDecoder(d, len(ref)).decode(d, index=None).cpu().numpy(), img.view(np.uint8).ravel())
w, h, n = 512, 384, 3            # (up to four tiles a call go by exits, larger batches by a chain a tile)
imgs = [o.generate(w, h, 3, 0, "NOISY3", 60 + t) for t in range(n)]
refs = [o.encode(im, 0, 8) for im in imgs]
pitch = (max(len(r) for r in refs) + 3) // 4 * 4
buf = np.zeros(n * pitch, dtype=np.uint8)
sizes = (C.c_size_t * n)()
for t, r in enumerate(refs):
    buf[t * pitch:t * pitch + len(r)] = r; sizes[t] = len(r)
dst = torch.from_numpy(buf).cuda()
dims = (C.c_size_t * 3)()
hdr = buf[:64].copy()
q = L.qb3_read_start(hdr.ctypes.data, sizes[0], dims)
assert L.qb3_read_info(q)
out = torch.zeros(n * w * h * 3, dtype=torch.uint8, device="cuda")
assert L.qb3x_decode_tiles(q, dst.data_ptr(), n, pitch, sizes, out.data_ptr(), w * h * 3, None, None) == n
got = out.cpu().numpy()
for t in range(n):
    assert np.array_equal(got[t * w * h * 3:(t + 1) * w * h * 3], imgs[t].view(np.uint8).ravel()), t
print("ok")
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for item in filter(None, switch.split(";")):
        name, _, value = item.partition("=")
        env[name] = value
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("switch", ["", "QB3_WALK_TAB_KB=6144", "QB3_WIDE_BAND=17"], ids=["default", "exits-in-many-rounds", "chain-through-the-table"])
def test_plain_single_band_wide_streams_by_exits(qb3, oracle, switch):
    """plain single-band 32/64-bit streams: the walk by exits of super-windows (walk_exitW_kernel, walk_exit_chain_kernel,
    walk_exit_units_kernel) -- rasters wide enough that the first unit of a block row leaves the band of sixteen rungs (walked
    from the code lengths inside the kernel), data over all rungs, a truncated stream, a batch of tiles; with so little table
    memory that the exits are taken in many rounds; and the chain through the table of round 3's first form, which stays for
    rasters of more than one band.  Pixels exact in every case (reference QB3decode.h:293-412).  A child process each: the
    switches are read once."""
    import subprocess
    import sys
    code = """
import sys, numpy as np, torch, ctypes as C
sys.path.insert(0, %r)
import qb3_amd
from qb3_amd import device as qdev
from oracle import pyoracle as o
L = qb3_amd.lib
for (w, h, dt, gen, mode, want_table) in [(2048, 1024, 5, "DEM", 8, True), (4100, 260, 5, "DEM", 4, True), (1500, 700, 7, "DEM", 8, True), (8192, 64, 5, "DEM", 8, True),
                                          (512, 512, 5, "LANDSAT16", 4, True), (256, 256, 7, "FEW", 8, False), (128, 256, 5, "RANDOM", 8, False), (64, 64, 6, "RUNG63", 8, False),
                                          (640, 480, 4, "TERRACE", 8, False),
                                          # common-factor streams (QB3M_CF_H = 5, QB3M_BEST = 7): units with the signal code parsed inside the walk, super-windows
                                          # in which a unit brings a factor of its own parsed outright by the hopping lane (QB3decode.h:619-716)
                                          (2048, 1024, 5, "DEM", 5, True), (1500, 700, 7, "DEM", 7, True), (1024, 512, 5, "TERRACE", 5, True), (512, 512, 7, "FEW", 7, False),
                                          (1024, 1024, 5, "SCALED", 5, True), (768, 512, 4, "LANDSAT16", 7, True),
                                          # ... of 16- and 8-bit data too (8-bit: the walk's unit lanes write the block table of the lane-per-block decoder)
                                          (1024, 1024, 3, "DEM", 7, True), (700, 500, 2, "LANDSAT16", 5, True), (1024, 768, 0, "NOISY3", 5, True), (515, 259, 1, "GRAD", 7, True)]:
    if gen == "SCALED":                         # every value a multiple of ten: every unit takes the factor the first ones brought
        img = (o.generate(w, h, 1, dt, "DEM", 11).astype(np.int64) // 16 * 10).astype(np.int32)
    else:
        img = o.generate(w, h, 1, dt, gen, 11)
    ref = o.encode(img, dt, mode)
    d = torch.from_numpy(ref).cuda()
    dec = qdev.DeviceDecoder(d, len(ref))
    L.qb3x_profile_enable(1); L.qb3x_profile_reset()
    out = dec.decode(d, index=None)
    torch.cuda.synchronize()
    names = C.create_string_buffer(1024)
    L.qb3x_profile_names(names, 1024)
    L.qb3x_profile_enable(0)
    assert np.array_equal(out.cpu().numpy(), img.view(np.uint8).ravel()), (w, h, dt, gen)
    if want_table and ref[10] != 255:
        assert b"dec_index_table" in names.value, (w, h, dt, gen, names.value)
        if gen == "DEM":        # ... and by hops, not by the hopping lane parsing the super-windows itself (status bit 6)
            assert not (L.qb3x_last_decode_status(dec.p) & 64), (w, h, dt, gen, mode)
# the same stream cut short: an error or clamped pixels as the reference's reader gives, never a hang or a fault
img = o.generate(1024, 1024, 1, 5, "DEM", 12)
ref = o.encode(img, 5, 8)
import random
rng = random.Random(5)
damaged = [np.ascontiguousarray(ref[:cut]) for cut in (len(ref) // 2, len(ref) - 1000)]      # cut short: an error or clamped pixels, never a hang or a fault
for trial in range(12):                             # ... and smashed: bit flips, runs of one byte value, anywhere behind the header
    s = ref.copy()
    if trial & 1:
        for _ in range(rng.randrange(1, 6)):
            at = rng.randrange(64, len(s)); s[at] ^= 1 << rng.randrange(8)
    else:
        at = rng.randrange(64, len(s)); k = min(len(s) - at, rng.randrange(1, 5000)); s[at:at + k] = rng.choice((0, 255, rng.randrange(256)))
    damaged.append(s)
for s in damaged:
    d = torch.from_numpy(s).cuda()
    dims = (C.c_size_t * 3)()
    q = L.qb3x_read_start_device(d.data_ptr(), len(s), dims, None)
    if q:
        out = torch.zeros(img.nbytes, dtype=torch.uint8, device="cuda")
        L.qb3x_decode_device(q, d.data_ptr(), out.data_ptr(), None, None)
        torch.cuda.synchronize()
        L.qb3_destroy_decoder(q)
d = torch.from_numpy(ref).cuda()                    # the GPU is still well: the intact stream decodes
assert np.array_equal(qdev.DeviceDecoder(d, len(ref)).decode(d, index=None).cpu().numpy(), img.view(np.uint8).ravel())
# a batch of plain tiles
w, h, n = 512, 384, 5
imgs = [o.generate(w, h, 1, 5, "DEM", 40 + t) for t in range(n)]
refs = [o.encode(im, 5, 8) for im in imgs]
pitch = (max(len(r) for r in refs) + 3) // 4 * 4
buf = np.zeros(n * pitch, dtype=np.uint8)
sizes = (C.c_size_t * n)()
for t, r in enumerate(refs):
    buf[t * pitch:t * pitch + len(r)] = r; sizes[t] = len(r)
dst = torch.from_numpy(buf).cuda()
dims = (C.c_size_t * 3)()
hdr = buf[:64].copy()
q = L.qb3_read_start(hdr.ctypes.data, sizes[0], dims)
assert L.qb3_read_info(q)
out = torch.zeros(n * w * h * 4, dtype=torch.uint8, device="cuda")
assert L.qb3x_decode_tiles(q, dst.data_ptr(), n, pitch, sizes, out.data_ptr(), w * h * 4, None, None) == n
got = out.cpu().numpy()
for t in range(n):
    assert np.array_equal(got[t * w * h * 4:(t + 1) * w * h * 4], imgs[t].view(np.uint8).ravel()), t
print("ok")
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for item in filter(None, switch.split(";")):
        name, _, value = item.partition("=")
        env[name] = value
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("case", [(256, 256, 1, 5, "DEM", FTL), (300, 200, 1, 7, "DEM", BASE), (260, 132, 3, 4, "NOISY3", BASE), (128, 128, 1, 5, "TERRACE", FTL),
                                  (128, 128, 1, 7, "FEW", FTL), (64, 64, 1, 6, "RUNG63", FTL), (256, 128, 5, 4, "RANDOM", FTL), (1024, 1024, 1, 5, "DEM", FTL),
                                  (1024, 768, 2, 7, "DEM", BASE), (509, 259, 1, 5, "DEM", FTL), (640, 480, 16, 4, "DEM", FTL), (2048, 2048, 1, 5, "LANDSAT16", BASE)],
                         ids=lambda c: "%dx%dx%d-t%d-%s-m%d" % c)
def test_plain_wide_streams_through_the_band_table(qb3, oracle, case):
    """plain (reference-made: no index, no restart table) 32/64-bit FTL/BASE streams: the first index segment is parsed
    outright, the rest walked through a table of unit ends by position for a band of sixteen rungs (walk_tableW_kernel,
    walk_chainW_kernel), the entering values come from a totals pass of the unit-parallel decoder; a stream that leaves the
    band (FEW / RANDOM data ranges over all rungs) falls back to the one-lane parser.  Pixels exact either way; and the
    table path is the one taken where the data allows it (reference QB3decode.h:293-412)."""
    import ctypes as C
    import torch
    from qb3_amd import device as qdev
    w, h, b, dt, gen, mode = case
    img = oracle.generate(w, h, b, dt, gen, 4)
    cb = None if b in (1, 3, 4) else list(range(b))
    ref = oracle.encode(img, dt, mode, cband=cb)
    d = torch.from_numpy(ref).cuda()
    dec = qdev.DeviceDecoder(d, len(ref))
    if cb is not None:
        qb3.lib.qb3x_set_decoder_compat(dec.p, 0)
    L = qb3.lib
    L.qb3x_profile_enable(1); L.qb3x_profile_reset()
    out = dec.decode(d, index=None)
    torch.cuda.synchronize()
    names = C.create_string_buffer(1024)
    L.qb3x_profile_names(names, 1024)
    L.qb3x_profile_enable(0)
    assert np.array_equal(out.cpu().numpy(), img.view(np.uint8).ravel())
    if gen in ("DEM", "LANDSAT16", "NOISY3") and ref[10] != 255:
        assert b"dec_index_table" in names.value, names.value          # the band table carried the stream
    got, dims, _, _ = qb3.decode(ref)                                       # ... and through the host-pointer API
    assert dims == (w, h, b) and np.array_equal(got, img.view(np.uint8).ravel())


def test_handles_come_and_go(qb3, oracle):
    """a container per tile, a handle per container: encoder and decoder handles of changing geometry made, used and destroyed in
    turn -- their device buffers come from and go back to the library's pool (qb3x_trim empties it in between) -- and every
    round trip is exact, whatever a buffer held before (reference calling pattern: cqb3.cpp:405-493, one handle per file)"""
    import torch
    from qb3_amd import device as qdev
    L = qb3.lib
    shapes = [(256, 256, 3, 0, "NOISY3", 8), (1024, 512, 1, 5, "DEM", 8), (64, 64, 3, 0, "GRAD", 4), (512, 512, 8, 2, "LANDSAT16", 4), (256, 256, 3, 0, "NOISY3", 7),
              (1024, 1024, 3, 0, "PALETTE", 8), (128, 96, 1, 7, "DEM", 5), (256, 256, 3, 0, "NOISY3", 8)]
    for rnd in range(3):
        for i, (w, h, b, dt, gen, mode) in enumerate(shapes):
            img = oracle.generate(w, h, b, dt, gen, 30 + i + 10 * rnd)
            cb = None if b in (1, 3, 4) else list(range(b))
            ref = oracle.encode(img, dt, mode, cband=cb)
            got = qb3.encode(img, dt, mode, cband=cb)                   # a fresh encoder handle
            assert np.array_equal(got, ref), (rnd, w, h, b, dt, gen, mode)
            d = torch.from_numpy(ref).cuda()
            dec = qdev.DeviceDecoder(d, len(ref))                       # a fresh decoder handle on the device copy
            if cb is not None:
                L.qb3x_set_decoder_compat(dec.p, 0)
            out = dec.decode(d, index=None)
            torch.cuda.synchronize()
            assert np.array_equal(out.cpu().numpy(), img.view(np.uint8).ravel()), (rnd, w, h, b, dt, gen, mode)
            dec.close()
        if rnd == 1:
            L.qb3x_trim()


# ---------------------------------------------------------------------------------------------------------
# the lane-per-unit decoders (k_dec_pxu.hip): every raster no lane-per-block kernel takes

PXU_CASES = [
    # w, h, bands, dtype, gen, core band map (None: identity)
    (96, 80, 2, 0, "NOISY3", None),
    (131, 77, 2, 0, "PALETTE", [1, 1]),
    (160, 120, 5, 0, "NOISY3", [1, 1, 1, 3, 4]),
    (67, 61, 7, 1, "NOISY3", None),
    (100, 52, 16, 0, "FEW", None),
    (61, 67, 5, 3, "DEM", [0, 0, 2, 2, 4]),
    (256, 36, 7, 2, "LANDSAT16", None),
    (128, 40, 9, 2, "LANDSAT16", [1, 1, 1, 3, 4, 5, 6, 7, 8]),
    (130, 70, 2, 5, "DEM", None),
    (64, 132, 3, 4, "NOISY3", [1, 1, 1]),
    (64, 64, 2, 7, "DEM", [1, 1]),
    (36, 36, 16, 6, "RANDOM", None),
    (48, 40, 5, 6, "RUNG63", None),
    (8, 4, 2, 5, "DEM", None),             # two blocks: a segment that is mostly empty
]
PXU_CF_ONLY = [
    # rasters whose FTL / BASE streams have lane-per-block kernels but whose common-factor streams do not
    (128, 64, 2, 2, "LANDSAT16", None),
    (120, 88, 3, 2, "TERRACE", [1, 1, 1]),
    (256, 64, 4, 3, "DEM", None),
    (96, 48, 8, 2, "FEW", None),
    (64, 36, 12, 3, "PALETTE", None),
]


def _kernels_of(qb3, fn):
    import ctypes as C
    L = qb3.lib
    L.qb3x_profile_enable(1); L.qb3x_profile_reset()
    r = fn()
    buf = C.create_string_buffer(2048)
    L.qb3x_profile_names(buf, 2048)
    L.qb3x_profile_enable(0)
    return r, set(buf.value.decode().split(","))


@pytest.mark.parametrize("mode", [FTL, BASE, BASE_Z, 5, 1])
@pytest.mark.parametrize("case", PXU_CASES + PXU_CF_ONLY, ids=lambda c: "%dx%dx%d-t%d-%s" % c[:5])
def test_lane_per_unit_decoders(qb3, oracle, case, mode):
    """index, table (levels 1 and 2) and plain decode of the rasters of k_dec_pxu.hip, against the oracle's streams; the decode with
    an index runs dec_units and nothing else (the lane-per-segment decoder's name is dec_segments)"""
    import torch
    from qb3_amd import synth, device as qdev
    w, h, b, dt, gen, cb = case
    if case in PXU_CF_ONLY and mode not in (5, 1):
        pytest.skip("this raster's FTL / BASE streams have a lane-per-block kernel")
    if gen == "RUNG63" and mode in (5, 1):
        pytest.skip("64-bit index units: the reference's sentinel defect (SURVEY B-2), which the device encoder does not reproduce")
    img = oracle.generate(w, h, b, dt, gen, 7)
    raw = img.view(np.uint8).ravel()
    cbm = cb if cb is not None else list(range(b))
    ref = oracle.encode(img, dt, mode, cband=cbm)
    assert np.array_equal(qb3.encode(img, dt, mode, cband=cbm), ref)
    if ref[10] == 255:
        pytest.skip("stored raw")
    # plain container: walk, then the parallel decoder
    out, dims, _, _ = qb3.decode(ref)
    assert dims == (w, h, b) and np.array_equal(out, raw), "plain decode"
    # out-of-band index through the device calls
    dimg = torch.from_numpy(img.view(np.uint8).reshape(-1).copy()).cuda()
    enc = qdev.DeviceEncoder(w, h, b, dt, mode=mode, cband=cbm)
    dst, n, index = enc.encode(dimg)
    assert np.array_equal(dst[:n].cpu().numpy(), ref)
    dec = qdev.DeviceDecoder(dst, n)
    got, names = _kernels_of(qb3, lambda: dec.decode(dst, index=index))
    assert torch.equal(got, torch.from_numpy(raw).cuda()), "indexed decode"
    assert "dec_units" in names and "dec_segments" not in names, names
    # self-indexed containers
    for level in (1, 2):
        c = qb3.encode(img, dt, mode, cband=cbm, index_chunk=level)
        assert len(c) > len(ref)
        want, _, _, _ = oracle.decode(c, identity=True)
        assert want is not None and np.array_equal(want, raw), "the reference decoder must step over the chunks"
        dc = torch.from_numpy(c).cuda()
        d2 = qdev.DeviceDecoder(dc, len(c))
        got, names = _kernels_of(qb3, lambda: d2.decode(dc, index=None))
        assert torch.equal(got, torch.from_numpy(raw).cuda()), ("table decode", level)
        assert qb3.lib.qb3x_last_decode_status(d2.p) == 0
        if level == 2 or mode in (5, 1):
            assert names == {"dec_units"}, (level, names)       # from the entries alone: one kernel
    # a damaged field costs time, not pixels
    c = qb3.encode(img, dt, mode, cband=cbm, index_chunk=2).copy()
    at = bytes(c).index(b"ix", 11) + 12 + 6 + b * (1 + img.itemsize * (2 if mode in (5, 1) else 1)) + 1
    c[at] ^= 0x15
    out, _, _, _ = qb3.decode(c)
    assert np.array_equal(out, raw), "decode with a damaged table"


def test_lane_per_unit_stride_and_tiles(qb3, oracle):
    """line strides and batched tiles through the lane-per-unit decoders"""
    import torch
    from qb3_amd import device as qdev
    for (w, h, b, dt, gen, mode) in [(100, 36, 5, 0, "NOISY3", FTL), (64, 44, 2, 5, "DEM", 5), (72, 40, 7, 2, "LANDSAT16", BASE)]:
        img = oracle.generate(w, h, b, dt, gen, 9)
        cbm = list(range(b))
        ref = oracle.encode(img, dt, mode, cband=cbm)
        L = qb3.lib
        buf = np.ascontiguousarray(ref)
        dims = (C_sz() * 3)()
        p = L.qb3_read_start(buf.ctypes.data, buf.size, dims)
        assert p and L.qb3_read_info(p)
        stride = (w * b + 5) * img.itemsize             # bytes; the call takes values (QB3.h:146-148)
        L.qb3_set_decoder_stride(p, w * b + 5)
        out = np.full(stride * h, 0xa5, dtype=np.uint8)
        assert L.qb3_read_data(p, out.ctypes.data) != 0
        L.qb3_destroy_decoder(p)
        rows = out.reshape(h, stride)
        assert np.array_equal(rows[:, :w * b * img.itemsize].ravel(), img.view(np.uint8).ravel())
        assert (rows[:, w * b * img.itemsize:] == 0xa5).all(), "bytes between the lines are not written"
        n = 5
        imgs = torch.stack([torch.from_numpy(oracle.generate(w, h, b, dt, gen, 40 + t).view(np.uint8).reshape(-1).copy()) for t in range(n)]).cuda()
        for ic in (False, 2):
            tc = qdev.TileBatchCoder(w, h, b, dt, n, mode=mode, want_index=not ic, index_chunk=ic)
            tc.encode(imgs)
            o = torch.zeros_like(imgs)
            tc.decode(o, use_index=not ic)
            assert torch.equal(o, imgs), (w, h, b, dt, mode, ic)
            tc.close()


def C_sz():
    import ctypes
    return ctypes.c_size_t


def test_rle0_containers_keep_their_table(qb3, oracle):
    """a self-indexed container whose RLE0 pass wins: the table stays in front of "DT" (it describes the block stream, which the
    decoder has again after the expansion), the bytes behind it are the reference's RLE0 bytes, the reference's reader steps over the
    chunks, and the decode uses the table (status bit 5 clear, one decoder kernel behind the expansion)"""
    import torch
    from qb3_amd import device as qdev
    won = 0
    for (w, h, b, dt, gen, mode) in [(512, 256, 3, 0, "CONST", 6), (256, 256, 3, 4, "NOISY3", 7), (300, 100, 1, 5, "TERRACE", 7), (256, 128, 5, 2, "CONST", 3),
                                     (200, 120, 2, 0, "GRAD", 2)]:
        img = oracle.generate(w, h, b, dt, gen, 3)
        raw = img.view(np.uint8).ravel()
        cb = None if b in (1, 3, 4) else list(range(b))
        ref = oracle.encode(img, dt, mode, cband=cb)
        if ref[10] not in (2, 3, 6, 7):
            continue
        won += 1
        for level in (1, 2):
            c = qb3.encode(img, dt, mode, cband=cb, index_chunk=level)
            dt_at = bytes(ref).index(b"DT", 11)
            extra = len(c) - len(ref)
            assert extra > 0 and bytes(c[:dt_at]) == bytes(ref[:dt_at]) and bytes(c[dt_at + extra:]) == bytes(ref[dt_at:]), (w, h, b, dt, mode, level)
            want, _, _, _ = oracle.decode(c, identity=True)
            assert want is not None and np.array_equal(want, raw)
            out, _, _, _ = qb3.decode(c)
            assert np.array_equal(out, raw)
            dc = torch.from_numpy(c).cuda()
            d = qdev.DeviceDecoder(dc, len(c))
            assert qb3.lib.qb3x_decoder_table_entries(d.p) > 0
            got, names = _kernels_of(qb3, lambda: d.decode(dc, index=None))
            assert torch.equal(got, torch.from_numpy(raw).cuda())
            assert qb3.lib.qb3x_last_decode_status(d.p) == 0, "the table was dropped"
            assert "dec_index_serial" not in names or level == 1, names
    assert won >= 2


@pytest.mark.parametrize("switch", ["", "QB3_WALK_TAB_KB=8192", "QB3_EXITS_FROM=0"], ids=["default", "chain-in-many-rounds", "two-band-rasters-by-exits-whatever-their-size"])
def test_plain_streams_of_several_bands_by_the_chain(qb3, oracle, switch):
    """plain (reference-made) streams of several bands through walk_tableN_kernel / walk_chainN_kernel (k_dec_walk_chain.hip): 8-bit
    rasters of 5 and 16 bands and 16-bit rasters of odd band counts in FTL / BASE (8-bit rasters of TWO bands, plain and common factor, go
    by exits like RGB -- walk_exitB_kernel<2>: 298 positions x 64 rung pairs --, 16-bit rasters of two bands in FTL / BASE too -- <2, false, 4>:
    556 positions x 256 rung pairs -- and by the chain when a batch holds more than four tiles); common-factor streams of several bands of 8- and
    16-bit data (8-bit RGBA with the lane-per-block decoder's block table, the others with the lane-per-unit decoder's dword per unit):
    signal units parsed by the walking lane from the window's stream words, the factors in force left at the segment starts once a
    unit has brought one (SCALED: every unit takes a factor brought early; PALETTE, FEW: index units and factors of their own all
    over).  With table memory for a few windows a round (the walk's state -- rungs, factors -- crosses rounds); truncated and
    smashed streams: an error or clamped pixels, never a hang or a fault; a batch of tiles.  The walk is named in the profile."""
    import subprocess
    import sys
    code = """
import sys, numpy as np, torch, ctypes as C
sys.path.insert(0, %r)
import qb3_amd
from qb3_amd import device as qdev
from oracle import pyoracle as o
L = qb3_amd.lib
def kernels_of(fn):
    L.qb3x_profile_enable(1); L.qb3x_profile_reset()
    r = fn(); torch.cuda.synchronize()
    buf = C.create_string_buffer(2048); L.qb3x_profile_names(buf, 2048); L.qb3x_profile_enable(0)
    return r, set(buf.value.decode().split(","))
cases = [(640, 480, 2, 0, "NOISY3", 8), (512, 300, 2, 2, "LANDSAT16", 4), (260, 200, 2, 3, "DEM", 8), (1024, 700, 2, 2, "PALETTE", 0), (509, 259, 5, 0, "NOISY3", 4), (300, 200, 16, 0, "FEW", 8), (768, 300, 2, 1, "PALETTE", 0),
         (512, 260, 5, 2, "LANDSAT16", 4), (260, 512, 7, 3, "DEM", 8), (128, 96, 15, 2, "RANDOM", 4),
         (1024, 512, 4, 0, "NOISY3", 5), (509, 259, 4, 0, "PALETTE", 7), (640, 480, 4, 0, "SCALED", 5), (512, 512, 2, 0, "NOISY3", 5), (400, 300, 5, 0, "FEW", 1),
         (512, 384, 8, 2, "LANDSAT16", 5), (256, 300, 3, 2, "TERRACE", 5), (300, 256, 4, 3, "SCALED", 5), (200, 120, 2, 2, "PALETTE", 1), (160, 100, 9, 2, "FEW", 5)]
for (w, h, b, dt, gen, mode) in cases:
    if gen == "SCALED":                         # every value a multiple of three: every unit takes the factor the first ones brought
        base = o.generate(w, h, b, dt, "NOISY3", 21)
        img = (base.astype(np.int64) // 3 * 3).astype(base.dtype)
    else:
        img = o.generate(w, h, b, dt, gen, 21)
    cb = None if b in (1, 3, 4) else list(range(b))
    ref = o.encode(img, dt, mode, cband=cb)
    if ref[10] == 255:
        continue
    d = torch.from_numpy(ref).cuda()
    dec = qdev.DeviceDecoder(d, len(ref))
    out, names = kernels_of(lambda: dec.decode(d, index=None))
    assert np.array_equal(out.cpu().numpy(), img.view(np.uint8).ravel()), (w, h, b, dt, gen, mode)
    if ref[10] in (0, 1, 4, 5, 8):              # (not under RLE0: the expansion is another path to the same walk)
        assert "dec_index_table" in names, (w, h, b, dt, gen, mode, names)
    assert L.qb3x_last_decode_status(dec.p) == 0, (w, h, b, dt, gen, mode)
    got, dims, _, _ = qb3_amd.decode(ref)
    assert np.array_equal(got, img.view(np.uint8).ravel()), (w, h, b, dt, gen, mode, "host")
import random
rng = random.Random(7)
for (w, h, b, dt, gen, mode) in [(768, 512, 4, 0, "NOISY3", 5), (512, 300, 5, 2, "LANDSAT16", 5), (640, 400, 2, 0, "NOISY3", 8), (640, 400, 2, 2, "LANDSAT16", 4)]:
    img = o.generate(w, h, b, dt, gen, 22)
    cb = None if b in (1, 3, 4) else list(range(b))
    ref = o.encode(img, dt, mode, cband=cb)
    damaged = [np.ascontiguousarray(ref[:cut]) for cut in (len(ref) // 2, len(ref) - 1000)]
    for trial in range(10):
        s = ref.copy()
        if trial & 1:
            for _ in range(rng.randrange(1, 6)):
                at = rng.randrange(64, len(s)); s[at] ^= 1 << rng.randrange(8)
        else:
            at = rng.randrange(64, len(s)); k = min(len(s) - at, rng.randrange(1, 5000)); s[at:at + k] = rng.choice((0, 255, rng.randrange(256)))
        damaged.append(s)
    for s in damaged:
        d = torch.from_numpy(s).cuda()
        dims = (C.c_size_t * 3)()
        q = L.qb3x_read_start_device(d.data_ptr(), len(s), dims, None)
        if q:
            out = torch.zeros(img.nbytes, dtype=torch.uint8, device="cuda")
            L.qb3x_decode_device(q, d.data_ptr(), out.data_ptr(), None, None)
            torch.cuda.synchronize()
            L.qb3_destroy_decoder(q)
    d = torch.from_numpy(ref).cuda()                    # the GPU is still well: the intact stream decodes
    assert np.array_equal(qdev.DeviceDecoder(d, len(ref)).decode(d, index=None).cpu().numpy(), img.view(np.uint8).ravel())
    cut = np.ascontiguousarray(ref[:len(ref) // 3])     # a truncated stream reads as zeros behind its end, like the reference's (bitstream.h:36)
    want, _, _, _ = o.decode(cut, identity=True)
    if want is not None:
        got, _, _, _ = qb3_amd.decode(cut)
        assert np.array_equal(got, want), (w, h, b, dt, gen, mode, "truncated")
for (w, h, b, dt, gen, mode, n) in [(256, 192, 4, 0, "NOISY3", 5, 3), (200, 100, 5, 2, "LANDSAT16", 4, 6), (320, 200, 2, 0, "NOISY3", 8, 3), (320, 200, 2, 0, "NOISY3", 5, 6)]:
    imgs = [o.generate(w, h, b, dt, gen, 60 + t) for t in range(n)]
    cb = None if b in (1, 3, 4) else list(range(b))
    refs = [o.encode(im, dt, mode, cband=cb) for im in imgs]
    pitch = (max(len(r) for r in refs) + 3) // 4 * 4
    buf = np.zeros(n * pitch, dtype=np.uint8)
    sizes = (C.c_size_t * n)()
    for t, r in enumerate(refs):
        buf[t * pitch:t * pitch + len(r)] = r; sizes[t] = len(r)
    dst = torch.from_numpy(buf).cuda()
    dims = (C.c_size_t * 3)()
    hdr = buf[:64].copy()
    q = L.qb3_read_start(hdr.ctypes.data, sizes[0], dims)
    assert L.qb3_read_info(q)
    raw = imgs[0].nbytes
    out = torch.zeros(n * raw, dtype=torch.uint8, device="cuda")
    assert L.qb3x_decode_tiles(q, dst.data_ptr(), n, pitch, sizes, out.data_ptr(), raw, None, None) == n
    got = out.cpu().numpy()
    for t in range(n):
        assert np.array_equal(got[t * raw:(t + 1) * raw], imgs[t].view(np.uint8).ravel()), (w, h, b, dt, t)
    L.qb3_destroy_decoder(q)
print("ok")
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for item in filter(None, switch.split(";")):
        name, _, value = item.partition("=")
        env[name] = value
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_streams_cut_short_decode_like_the_reference(qb3, oracle):
    """a stream that ends early is not an error in the reference: its reader gives zeros behind the end (bitstream.h:36) and only more
    than 7 unused bits fail (QB3decode.h:411,569,740).  Every raster family -- whichever walk its plain stream takes first (exits,
    chains, lanes) -- ends at the one-lane parser for such a stream and returns the oracle's pixels; bytes behind the stream's end in
    the library's buffer do not leak into them (a second decode through the same handle pool, after a longer stream)"""
    cases = [(768, 512, 3, 0, "NOISY3", 8), (768, 512, 4, 0, "NOISY3", 8), (512, 512, 1, 0, "NOISY3", 4), (640, 400, 2, 0, "NOISY3", 8), (400, 300, 5, 0, "NOISY3", 4),
             (512, 512, 1, 2, "DEM", 4), (512, 300, 2, 2, "LANDSAT16", 4), (512, 300, 8, 2, "LANDSAT16", 4), (512, 300, 5, 2, "LANDSAT16", 4), (256, 256, 1, 5, "DEM", 8), (256, 256, 2, 5, "DEM", 8),
             (256, 256, 1, 7, "DEM", 8), (768, 512, 3, 0, "NOISY3", 5), (768, 512, 4, 0, "NOISY3", 5), (768, 512, 1, 0, "NOISY3", 5), (400, 300, 5, 0, "NOISY3", 5),
             (512, 300, 5, 2, "LANDSAT16", 5), (512, 300, 8, 2, "LANDSAT16", 1), (256, 256, 1, 5, "DEM", 5), (200, 200, 3, 4, "NOISY3", 5)]
    for (w, h, b, dt, gen, mode) in cases:
        img = oracle.generate(w, h, b, dt, gen, 22)
        cb = None if b in (1, 3, 4) else list(range(b))
        ref = oracle.encode(img, dt, mode, cband=cb)
        out, _, _, _ = qb3.decode(ref)                  # (a full-length stream first: what it leaves in the library's buffers must not show below)
        assert np.array_equal(out, img.view(np.uint8).ravel())
        for frac in (3, 2):
            cut = np.ascontiguousarray(ref[:len(ref) // frac])
            want, _, _, _ = oracle.decode(cut, identity=True)
            if want is None:
                with pytest.raises(RuntimeError):
                    qb3.decode(cut)
                continue
            got, _, _, _ = qb3.decode(cut)
            assert np.array_equal(got, want), (w, h, b, dt, gen, mode, frac)


def test_rle0_passes_run_only_for_streams_with_zero_runs(qb3, oracle):
    """The RLE0 modes count, while the chunks are concatenated, the positions at which four zero bytes start -- exactly, the positions at
    chunk boundaries on the finished dwords (finish_seam) -- and a stream that has none goes without the byte passes over it; one that has
    them gets the passes and the reference's decision.  (Counted per chunk as the chunk sees a shared dword, a noise raster of 4 113
    chunks came out with 319 runs it does not have and paid for the size pass.)  The container is the oracle's either way."""
    import torch
    from qb3_amd import synth, device as qdev
    for (w, h, b, dt, gen, expect_pass) in [(1024, 1024, 3, 0, "NOISY3", False), (1024, 1024, 1, 0, "CONST", None), (1024, 1024, 3, 0, "TERRACE", None), (1024, 1024, 8, 2, "LANDSAT16", None), (515, 389, 1, 3, "DEM", None)]:
        img = synth.generate(w, h, b, dt, gen, 3)
        ref = oracle.encode(img.cpu().numpy().view(oracle.NPTYPE[dt]), dt, 7, fix_b2=True)
        enc = qdev.DeviceEncoder(w, h, b, dt, mode=7)
        enc.encode(img)
        qdev.profile_reset(); qdev.profile_enable(1)
        dst, n, _ = enc.encode(img)
        torch.cuda.synchronize()
        qdev.profile_enable(False)
        names = set(qdev.profile_report())
        got = dst[:n].cpu().numpy()
        assert n == len(ref) and np.array_equal(got, ref), (w, h, b, dt, gen, first_diff(got, ref))
        if expect_pass is not None:
            assert any(k.startswith("rle0") for k in names) == expect_pass, (gen, sorted(names))


def test_rle0_decision_with_short_zero_runs_at_chunk_boundaries(qb3, oracle):
    """Whether an RLE0 mode's byte passes run at all hangs on the count of zero runs, and a stream of noise with ONE short run of zero
    bytes is the close call: RLE0 wins it by a byte or two, or not at all.  Flat patches of a few blocks (a handful of zero bytes in the
    stream) are laid over the blocks around the encoder's chunk boundaries (255 blocks a chunk), where the run's bytes lie in dwords two
    chunks share: the container must be the oracle's -- mode byte, size and bytes -- every time."""
    import torch
    from qb3_amd import synth, device as qdev
    w, h, b, dt = 1024, 256, 3, 0
    base = synth.generate(w, h, b, dt, "NOISY3", 11)
    enc = qdev.DeviceEncoder(w, h, b, dt, mode=7)
    rng = np.random.default_rng(5)
    wins = 0
    for trial in range(48):
        k = int(rng.integers(1, (w // 4) * (h // 4) // 255))            # a chunk boundary: in front of block 255 * k
        first = 255 * k - int(rng.integers(1, 8)); nblk = int(rng.integers(4, 12))
        img = base.clone()
        for g in range(first, first + nblk):
            by, bx = divmod(g, w // 4)
            if by < h // 4: img[4 * by:4 * by + 4, 4 * bx:4 * bx + 4, :] = 77
        ref = oracle.encode(img.cpu().numpy().view(oracle.NPTYPE[dt]), dt, 7, fix_b2=True)
        dst, n, _ = enc.encode(img)
        got = dst[:n].cpu().numpy()
        assert n == len(ref) and np.array_equal(got, ref), (trial, k, first, nblk, int(got[10]), int(ref[10]), first_diff(got, ref))
        wins += int(ref[10]) == 7
    assert 0 < wins, "no trial made RLE0 win: the patches are too short to test the decision"


@pytest.mark.parametrize("case", [(4096, 4096, 3, 7, "DEM", 8), (4096, 4096, 2, 7, "DEM", 8), (4096, 4096, 2, 7, "DEM", 0), (4096, 4096, 1, 7, "DEM", 8)],
                         ids=lambda c: "%dx%dx%d-t%d-%s-m%d" % c)
def test_wide_decode_is_repeatable(qb3, case):
    """The same container and index must decode to the same pixels every time, whatever ran on the device before.  The 64-bit
    lane-per-unit decoder read, in a few calls out of a hundred and only right behind an encode, the last value of a block row's first
    unit (rung 18, stream position a multiple of 32) one bit late: same inputs, status 0, the next call right again.
    Its values now come out of a window without a branch (wide_values_lds, qb3_wide.h); this is the flow that showed the fault."""
    import torch
    from qb3_amd import synth, device as qdev
    w, h, b, dt, gen, mode = case
    img = synth.generate(w, h, b, dt, gen, 3)
    raw = img.reshape(-1).view(torch.uint8)
    enc = qdev.DeviceEncoder(w, h, b, dt, mode=mode, index_chunk=2)
    out = torch.empty(raw.numel(), dtype=torch.uint8, device=img.device)
    first = None
    for rep in range(120):
        dst, n, index = enc.encode(img)
        cur = dst[:n].clone()
        if first is None: first = cur
        assert torch.equal(cur, first), f"run {rep}: the container differs"
        dec = qdev.DeviceDecoder(dst, n)
        for name, ix in (("index", index), ("table", None)):
            out.zero_()
            dec.decode(dst, out=out, index=ix)
            assert torch.equal(out, raw), f"run {rep}, by the {name}: {int((out != raw).sum())} bytes differ"
        dec.close()


def test_large_lane_per_unit_rasters_through_the_host_calls(qb3, oracle):
    """rasters of 64 MB and more of the lane-per-unit shapes through qb3_encode / qb3_read_data (the encode side codes them strip by
    strip): the plain container is the oracle's, the self-indexed one the oracle's plus table chunks, both decode exactly"""
    for (w, h, b, dt, gen, mode) in [(4096, 4096, 2, 5, "DEM", 8), (4096, 2048, 5, 2, "LANDSAT16", 4), (4096, 4096, 5, 0, "NOISY3", 5)]:
        img = oracle.generate(w, h, b, dt, gen, 5)
        cb = list(range(b))
        ref = oracle.encode(img, dt, mode, cband=cb)
        for level in (0, 2):
            got = qb3.encode(img, dt, mode, cband=cb, index_chunk=level)
            if level == 0:
                assert np.array_equal(got, ref), (w, h, b, dt)
            else:
                dt_at, extra = bytes(ref).index(b"DT", 11), len(got) - len(ref)
                assert extra > 0 and np.array_equal(np.concatenate([got[:dt_at], got[dt_at + extra:]]), ref), (w, h, b, dt)
            out, _, _, _ = qb3.decode(got)
            assert np.array_equal(out, img.view(np.uint8).ravel()), (w, h, b, dt, level)
