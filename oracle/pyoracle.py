"""oracle/pyoracle.py -- ctypes binding of the CPU restatement (oracle/libqb3oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
Nothing under qb3_amd/ may import this module.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libqb3oracle.so")

GEN = {"GRAD": 0, "NOISY3": 1, "LANDSAT16": 2, "DEM": 3, "TERRACE": 4, "FEW": 5, "PALETTE": 6, "RANDOM": 7,
       "RUNG63": 8, "CONST": 9}
NPTYPE = (np.uint8, np.int8, np.uint16, np.int16, np.uint32, np.int32, np.uint64, np.int64)
TYPESIZE = (1, 1, 2, 2, 4, 4, 8, 8)


def build():
    """Compile the oracle if the shared object is missing or older than its sources."""
    srcs = [os.path.join(_HERE, f) for f in ("qb3o.c", "qb3o_tables.c", "qb3o_gen.c", "qb3o.h", "qb3o_bits.h",
                                             "qb3o_gen.h", "qb3o_codec.inc")]
    if os.path.exists(_SO) and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in srcs):
        return
    subprocess.run(["make", "-C", _HERE, "libqb3oracle.so"], check=True, capture_output=True)


build()
lib = C.CDLL(_SO)
_vp, _sz, _u64 = C.c_void_p, C.c_size_t, C.c_uint64
for name, res, args in [
    ("qb3o_encoder_new", _vp, [_sz, _sz, _sz, C.c_int]),
    ("qb3o_decoder_new", _vp, [_vp, _sz, C.POINTER(_sz)]),
    ("qb3o_free", None, [_vp]),
    ("qb3o_encoder_reset", None, [_vp]),
    ("qb3o_set_coreband", C.c_int, [_vp, _sz, C.POINTER(_sz)]),
    ("qb3o_set_quanta", C.c_int, [_vp, _u64, C.c_int]),
    ("qb3o_set_mode", C.c_int, [_vp, C.c_int]),
    ("qb3o_set_stride", None, [_vp, _sz]),
    ("qb3o_set_fix_b2", None, [_vp, C.c_int]),
    ("qb3o_set_order", None, [_vp, _u64]),
    ("qb3o_get_error", C.c_int, [_vp]),
    ("qb3o_get_encoder_mode", C.c_int, [_vp]),
    ("qb3o_get_band_state", None, [_vp, C.POINTER(_u64)]),
    ("qb3o_max_encoded_size", _sz, [_vp]),
    ("qb3o_encode", _sz, [_vp, _vp, _vp]),
    ("qb3o_encode_raw", _u64, [_vp, _vp, _vp]),
    ("qb3o_read_info", C.c_int, [_vp]),
    ("qb3o_decoded_size", _sz, [_vp]),
    ("qb3o_read_data", _sz, [_vp, _vp]),
    ("qb3o_decoder_set_stride", None, [_vp, _sz]),
    ("qb3o_decoder_set_identity", None, [_vp, C.c_int]),
    ("qb3o_decoder_mode", C.c_int, [_vp]),
    ("qb3o_decoder_type", C.c_int, [_vp]),
    ("qb3o_decoder_error", C.c_int, [_vp]),
    ("qb3o_decoder_quanta", _u64, [_vp]),
    ("qb3o_decoder_order", _u64, [_vp]),
    ("qb3o_decoder_coreband", C.c_int, [_vp, C.POINTER(_sz)]),
    ("qb3o_generate", None, [_vp, _sz, _sz, _sz, C.c_int, C.c_int, _u64]),
    ("qb3o_fnv1a64", _u64, [_vp, _sz]),
    ("qb3o_rle0", _sz, [_vp, _sz, _vp]),
    ("qb3o_rle0_size", _sz, [_vp, _sz]),
    ("qb3o_derle0", C.c_int64, [_vp, _sz, _vp, _sz]),
    ("qb3o_derle0_size", _sz, [_vp, _sz]),
]:
    f = getattr(lib, name)
    f.restype, f.argtypes = res, args


def _p(a):
    return a.ctypes.data_as(_vp)


def generate(w, h, bands, dtype, gen, seed):
    """Synthetic raster of SURVEY.md section 8(d); returns an array of shape (h, w, bands)."""
    a = np.empty((h, w, bands), dtype=NPTYPE[dtype])
    lib.qb3o_generate(_p(a), w, h, bands, TYPESIZE[dtype], GEN[gen] if isinstance(gen, str) else gen, seed)
    return a


def fnv(a):
    a = np.ascontiguousarray(a)
    return "%016x" % lib.qb3o_fnv1a64(_p(a), a.nbytes)


class Encoder:
    """The oracle's encoder handle (same statefulness as the reference's encs)."""

    def __init__(self, w, h, bands, dtype):
        self.p = lib.qb3o_encoder_new(w, h, bands, dtype)
        if not self.p:
            raise ValueError("bad encoder parameters")
        self.bands = bands

    def __del__(self):
        if getattr(self, "p", None):
            lib.qb3o_free(self.p)
            self.p = None

    def set_mode(self, m):
        return lib.qb3o_set_mode(self.p, m)

    def set_coreband(self, cb):
        arr = (_sz * self.bands)(*cb)
        lib.qb3o_set_coreband(self.p, self.bands, arr)
        return list(arr)

    def set_stride(self, s):
        lib.qb3o_set_stride(self.p, s)

    def set_quanta(self, q, away=False):
        return bool(lib.qb3o_set_quanta(self.p, q, int(away)))

    def reset(self):
        lib.qb3o_encoder_reset(self.p)

    def max_size(self):
        return lib.qb3o_max_encoded_size(self.p)

    def band_state(self):
        st = (_u64 * (3 * self.bands))()
        lib.qb3o_get_band_state(self.p, st)
        return [tuple(st[3 * c:3 * c + 3]) for c in range(self.bands)]

    def encode(self, img):
        src = np.ascontiguousarray(img)
        dst = np.empty(self.max_size(), dtype=np.uint8)
        n = lib.qb3o_encode(self.p, _p(src), _p(dst))
        if n == 0:
            raise RuntimeError("oracle encode failed, error %d" % lib.qb3o_get_error(self.p))
        return dst[:n].copy()


def encode(img, dtype, mode=8, cband=None, stride=0, quanta=1, away=False, fix_b2=False, order=0):
    h, w, b = img.shape
    e = Encoder(w, h, b, dtype)
    e.set_mode(mode)
    if order:
        lib.qb3o_set_order(e.p, order)
    if fix_b2:
        lib.qb3o_set_fix_b2(e.p, 1)
    if cband is not None:
        e.set_coreband(cband)
    if stride:
        e.set_stride(stride)
    if quanta > 1:
        e.set_quanta(quanta, away)
    return e.encode(img)


def decode(stream, identity=False, stride=0):
    """Returns (decoded bytes as uint8 array or None on failure, (w, h, bands), dtype, mode)."""
    buf = np.ascontiguousarray(stream, dtype=np.uint8)
    dims = (_sz * 3)()
    p = lib.qb3o_decoder_new(_p(buf), buf.size, dims)
    if not p:
        raise ValueError("oracle read_start rejected the stream")
    try:
        if identity:
            lib.qb3o_decoder_set_identity(p, 1)
        if not lib.qb3o_read_info(p):
            raise ValueError("oracle read_info failed")
        if stride:
            lib.qb3o_decoder_set_stride(p, stride)
        out = np.zeros(lib.qb3o_decoded_size(p) if not stride else stride * dims[1] * TYPESIZE[lib.qb3o_decoder_type(p)], dtype=np.uint8)
        n = lib.qb3o_read_data(p, _p(out))
        return (out if n else None), tuple(dims), lib.qb3o_decoder_type(p), lib.qb3o_decoder_mode(p)
    finally:
        lib.qb3o_free(p)
