"""qb3_amd -- Python access to the MI355X-native QB3 library (qb3_amd/libQB3.so) through its C ABI.

The product is the shared library (sources in qb3_amd/csrc, headers in include/); this module only binds
the C entry points with ctypes so that tests and bench.py can call them.  Nothing here encodes or decodes
by itself and nothing falls back to a CPU implementation: if the library is missing, import fails.

Reference interface mirrored: QB3lib/QB3.h:85-162 (names, argument order, return conventions).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("QB3_LIB_PATH") or os.path.join(_HERE, "libQB3.so")     # (QB3_LIB_PATH: a diagnostic build of the library, scratch/variant.sh)

if not os.path.exists(LIB_PATH):
    raise ImportError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                      "or `make -C qb3_amd/csrc` (there is no CPU fallback)")

# One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64 and libQB3.so is linked against the
# system one (same soname).  Whichever is loaded first serves both, and torch does not see the GPU when the
# system copy wins -- so when torch is installed, load it first.  The library itself does not need torch.
try:
    import torch as _torch  # noqa: F401
except ImportError:         # pure C/ctypes use
    _torch = None

lib = C.CDLL(LIB_PATH)

# enum values as in include/QB3.h
QB3_U8, QB3_I8, QB3_U16, QB3_I16, QB3_U32, QB3_I32, QB3_U64, QB3_I64 = range(8)
QB3M_BASE_Z, QB3M_CF, QB3M_RLE, QB3M_CF_RLE, QB3M_BASE_H, QB3M_CF_H, QB3M_RLE_H, QB3M_CF_RLE_H, QB3M_FTL = range(9)
QB3M_DEFAULT, QB3M_BASE, QB3M_BEST, QB3M_STORED, QB3M_INVALID = 8, 4, 7, 255, -1
QB3X_REF_CBAND0 = 1
TYPESIZE = (1, 1, 2, 2, 4, 4, 8, 8)

_vp, _sz, _u64 = C.c_void_p, C.c_size_t, C.c_uint64
_PROTOS = {
    # name: (restype, argtypes)            -- include/QB3.h
    "qb3_create_encoder": (_vp, [_sz, _sz, _sz, C.c_int]),
    "qb3_destroy_encoder": (None, [_vp]),
    "qb3_reset_encoder": (None, [_vp]),
    "qb3_set_encoder_coreband": (C.c_bool, [_vp, _sz, C.POINTER(_sz)]),
    "qb3_set_encoder_quanta": (C.c_bool, [_vp, _u64, C.c_bool]),
    "qb3_max_encoded_size": (_sz, [_vp]),
    "qb3_set_encoder_mode": (C.c_int, [_vp, C.c_int]),
    "qb3_set_encoder_stride": (None, [_vp, _sz]),
    "qb3_encode": (_sz, [_vp, _vp, _vp]),
    "qb3_get_encoder_state": (C.c_int, [_vp]),
    "qb3_read_start": (_vp, [_vp, _sz, C.POINTER(_sz)]),
    "qb3_read_info": (C.c_bool, [_vp]),
    "qb3_read_data": (_sz, [_vp, _vp]),
    "qb3_destroy_decoder": (None, [_vp]),
    "qb3_decoded_size": (_sz, [_vp]),
    "qb3_get_type": (C.c_int, [_vp]),
    "qb3_set_decoder_stride": (None, [_vp, _sz]),
    "qb3_get_mode": (C.c_int, [_vp]),
    "qb3_get_quanta": (_u64, [_vp]),
    "qb3_get_order": (_u64, [_vp]),
    "qb3_get_coreband": (C.c_bool, [_vp, C.POINTER(_sz)]),
    # include/qb3x.h
    "qb3x_device_count": (C.c_int, []),
    "qb3x_trim": (None, []),
    "qb3x_last_decode_status": (C.c_uint, [_vp]),
    "qb3x_index_size": (_sz, [_vp]),
    "qb3x_set_encoder_index_chunk": (None, [_vp, C.c_int]),
    "qb3x_decoder_index_size": (_sz, [_vp]),
    "qb3x_encode_device": (_sz, [_vp, _vp, _vp, _vp, _vp]),
    "qb3x_decode_device": (_sz, [_vp, _vp, _vp, _vp, _vp]),
    "qb3x_encode_tiles": (_sz, [_vp, _vp, _sz, _sz, _vp, _sz, _vp, C.POINTER(_sz), _vp]),
    "qb3x_decode_tiles": (_sz, [_vp, _vp, _sz, _sz, C.POINTER(_sz), _vp, _sz, _vp, _vp]),
    "qb3x_decode_tile_ok": (C.c_int, [_vp, _sz]),
    "qb3x_read_start": (_vp, [_vp, _sz, _sz, C.POINTER(_sz)]),
    "qb3x_read_start_device": (_vp, [_vp, _sz, C.POINTER(_sz), _vp]),
    "qb3x_header_size_bound": (_sz, [_vp, _sz]),
    "qb3x_decoder_table_entries": (_sz, [_vp]),
    "qb3x_set_decoder_compat": (None, [_vp, C.c_uint]),
    "qb3_create_decoder": (_vp, [_vp, _sz, C.POINTER(_sz)]),
    "qb3_decode": (_sz, [_vp, _vp]),
    "qb3x_last_error": (C.c_char_p, []),
    "qb3x_fnv1a64": (_u64, [_vp, _sz, _u64]),
    "qb3x_rle0_device": (_sz, [_vp, _sz, _vp, _sz, C.c_int, _vp]),
    "qb3x_profile_enable": (None, [C.c_int]),
    "qb3x_profile_reset": (None, []),
    "qb3x_profile_get": (C.c_int, [C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "qb3x_profile_names": (C.c_int, [C.c_char_p, _sz]),
}
for _name, (_res, _args) in _PROTOS.items():
    _f = getattr(lib, _name)        # AttributeError here = the library does not export a declared symbol
    _f.restype, _f.argtypes = _res, _args

EXPORTED = tuple(_PROTOS)


def last_error():
    return lib.qb3x_last_error().decode()


def fnv(*parts):
    """FNV-1a64 (hex) of the concatenation of host arrays."""
    import numpy as np
    h = 0
    for a in parts:
        a = np.ascontiguousarray(a)
        if a.nbytes:
            h = lib.qb3x_fnv1a64(a.ctypes.data_as(_vp), a.nbytes, h)
    return "%016x" % (h or 0xcbf29ce484222325)


def _np_ptr(a):
    return a.ctypes.data_as(_vp)


def encode(img, dtype, mode=QB3M_FTL, cband=None, stride=0, quanta=1, away=False, index_chunk=False):
    """qb3_create_encoder .. qb3_encode on a numpy image of shape (h, w, bands); returns the container bytes."""
    import numpy as np
    h, w, b = img.shape
    p = lib.qb3_create_encoder(w, h, b, dtype)
    if not p:
        raise ValueError("qb3_create_encoder refused the parameters")
    try:
        lib.qb3_set_encoder_mode(p, mode)
        if index_chunk:
            lib.qb3x_set_encoder_index_chunk(p, int(index_chunk))      # 1: restart table; 2: with block lengths
        if cband is not None:
            arr = (_sz * b)(*cband)
            lib.qb3_set_encoder_coreband(p, b, arr)
        if stride:
            lib.qb3_set_encoder_stride(p, stride)
        if quanta > 1:
            lib.qb3_set_encoder_quanta(p, quanta, away)
        dst = np.empty(lib.qb3_max_encoded_size(p), dtype=np.uint8)
        src = np.ascontiguousarray(img)
        n = lib.qb3_encode(p, _np_ptr(src), _np_ptr(dst))
        if n == 0:
            raise RuntimeError(f"qb3_encode failed, state {lib.qb3_get_encoder_state(p)}: {last_error()}")
        return dst[:n].copy()
    finally:
        lib.qb3_destroy_encoder(p)


def decode(stream, compat=0):
    """qb3_read_start .. qb3_read_data; returns (flat uint8 array of decoded bytes, (w, h, bands), dtype, mode)."""
    import numpy as np
    buf = np.ascontiguousarray(stream, dtype=np.uint8)
    dims = (_sz * 3)()
    p = lib.qb3_read_start(_np_ptr(buf), buf.size, dims)
    if not p:
        raise ValueError("qb3_read_start rejected the stream")
    try:
        if not lib.qb3_read_info(p):
            raise ValueError("qb3_read_info failed")
        if compat:
            lib.qb3x_set_decoder_compat(p, compat)
        out = np.empty(lib.qb3_decoded_size(p), dtype=np.uint8)
        n = lib.qb3_read_data(p, _np_ptr(out))
        if n == 0:
            raise RuntimeError(f"qb3_read_data failed: {last_error()}")
        return out[:n], tuple(dims), lib.qb3_get_type(p), lib.qb3_get_mode(p)
    finally:
        lib.qb3_destroy_decoder(p)
