"""qb3_amd/device.py -- device-resident encode/decode through the qb3x_ C entry points, on torch tensors.

torch only supplies device memory and the stream handle; every byte of work happens inside libQB3.so.
The handles follow the reference's usage pattern (cqb3.cpp:405-493 for encode, :276-323 for decode):
create, set mode / band map, encode; read_start, read_info, read_data.
"""
import ctypes as C

import numpy as np
import torch

from . import lib, last_error, TYPESIZE, QB3M_FTL, _sz

_vp = C.c_void_p


def _stream_ptr():
    return _vp(torch.cuda.current_stream().cuda_stream)


class DeviceEncoder:
    """One encoder handle bound to the current device; output and index buffers are reused across calls."""

    def __init__(self, w, h, bands, dtype, mode=QB3M_FTL, cband=None, want_index=True, index_chunk=False):
        self.w, self.h, self.bands, self.dtype = w, h, bands, dtype
        self.p = lib.qb3_create_encoder(w, h, bands, dtype)
        if not self.p:
            raise ValueError("qb3_create_encoder refused the parameters")
        self.mode = lib.qb3_set_encoder_mode(self.p, mode)
        if index_chunk:             # self-indexing container (qb3x.h): one more chunk, up to 64 KB
            lib.qb3x_set_encoder_index_chunk(self.p, int(index_chunk))   # 1: restart table; 2: with block lengths
        if cband is not None:
            arr = (_sz * bands)(*cband)
            lib.qb3_set_encoder_coreband(self.p, bands, arr)
        self.max_size = lib.qb3_max_encoded_size(self.p)
        self.raw_bytes = w * h * bands * TYPESIZE[dtype]
        self.index_bytes = lib.qb3x_index_size(self.p) if want_index else 0
        self.dst = None
        self.index = None

    def close(self):
        if self.p and lib is not None:      # `lib` may already be gone at interpreter shutdown
            lib.qb3_destroy_encoder(self.p)
            self.p = None

    __del__ = close

    def encode(self, src, dst=None, index=None):
        """src: device tensor holding the image bytes.  Returns (dst uint8 tensor, container size, index tensor)."""
        assert src.is_cuda and src.is_contiguous() and src.numel() * src.element_size() >= self.raw_bytes
        if dst is None:
            if self.dst is None:
                self.dst = torch.empty((self.max_size + 3) // 4 * 4, dtype=torch.uint8, device=src.device)
            dst = self.dst
        if index is None and self.index_bytes:
            if self.index is None:
                self.index = torch.empty(self.index_bytes, dtype=torch.uint8, device=src.device)
            index = self.index
        lib.qb3_reset_encoder(self.p)               # independent images: do not carry the band state
        lib.qb3_set_encoder_mode(self.p, self.mode)  # a STORED fallback leaves the handle's mode at 255
        n = lib.qb3x_encode_device(self.p, _vp(src.data_ptr()), _vp(dst.data_ptr()),
                                   _vp(index.data_ptr()) if index is not None else None, _stream_ptr())
        if n == 0:
            raise RuntimeError(f"qb3x_encode_device failed (state {lib.qb3_get_encoder_state(self.p)}): {last_error()}")
        return dst, n, index


class DeviceDecoder:
    """Decoder for one device-resident container.  `container` is the device tensor holding it -- the library then reads
    the few header bytes it needs itself (qb3x_read_start_device: two small copies, whatever the size of a restart table) --
    or a host array holding the container's head up to its "DT" mark (qb3x_read_start)."""

    def __init__(self, container, nbytes):
        nbytes = int(nbytes)
        dims = (_sz * 3)()
        if torch.is_tensor(container):
            self.hdr = None
            self.p = lib.qb3x_read_start_device(_vp(container.data_ptr()), nbytes, dims, _stream_ptr())
            if not self.p:
                raise ValueError("not a QB3 container (or its header does not parse)")
        else:
            self.hdr = np.ascontiguousarray(container, dtype=np.uint8)
            self.p = lib.qb3x_read_start(self.hdr.ctypes.data_as(_vp), min(self.hdr.size, nbytes), nbytes, dims)
            if not self.p:
                raise ValueError("qb3x_read_start rejected the stream")
            if not lib.qb3_read_info(self.p):
                lib.qb3_destroy_decoder(self.p)
                self.p = None
                raise ValueError("qb3_read_info failed (or the host copy ends before the container's DT mark)")
        self.w, self.h, self.bands = dims[0], dims[1], dims[2]
        self.out_bytes = lib.qb3_decoded_size(self.p)
        self.nbytes = nbytes

    def close(self):
        if self.p and lib is not None:
            lib.qb3_destroy_decoder(self.p)
            self.p = None

    __del__ = close

    def decode(self, d_stream, out=None, index=None):
        assert d_stream.is_cuda
        if out is None:
            out = torch.empty(self.out_bytes, dtype=torch.uint8, device=d_stream.device)
        n = lib.qb3x_decode_device(self.p, _vp(d_stream.data_ptr()), _vp(out.data_ptr()),
                                   _vp(index.data_ptr()) if index is not None else None, _stream_ptr())
        if n == 0:
            raise RuntimeError(f"qb3x_decode_device failed: {last_error()}")
        return out


class TileBatchCoder:
    """qb3x_encode_tiles / qb3x_decode_tiles on torch tensors: n independent tiles of one geometry per call, tile i of
    the input at i * raw_bytes, its container at i * pitch of `dst`, its out-of-band index at i * index_bytes."""

    def __init__(self, w, h, bands, dtype, n, mode=QB3M_FTL, device="cuda", want_index=True, index_chunk=False):
        """index_chunk: every tile's container carries its own restart table (include/qb3x.h), so that decode(use_index=False)
        needs nothing but the containers"""
        self.w, self.h, self.bands, self.dtype, self.n, self.mode = w, h, bands, dtype, n, mode
        self.p = lib.qb3_create_encoder(w, h, bands, dtype)
        if not self.p:
            raise ValueError("qb3_create_encoder refused the parameters")
        lib.qb3_set_encoder_mode(self.p, mode)
        lib.qb3x_set_encoder_index_chunk(self.p, int(index_chunk))
        self.raw_bytes = w * h * bands * TYPESIZE[dtype]
        self.pitch = (lib.qb3_max_encoded_size(self.p) + 3) // 4 * 4
        self.index_bytes = lib.qb3x_index_size(self.p) if want_index else 0
        self.dst = torch.empty(n * self.pitch, dtype=torch.uint8, device=device)
        self.index = torch.empty(max(1, n * self.index_bytes), dtype=torch.uint8, device=device) if want_index else None
        self.sizes = (_sz * n)()
        self.d = None

    def close(self):
        if lib is None:
            return
        if self.p:
            lib.qb3_destroy_encoder(self.p)
            self.p = None
        if self.d:
            lib.qb3_destroy_decoder(self.d)
            self.d = None

    __del__ = close

    def encode(self, imgs, first=0, count=None):
        """imgs: device tensor holding the tiles back to back; codes tiles [first, first + count) of it.  Returns their sizes."""
        count = self.n - first if count is None else count
        assert imgs.is_cuda and imgs.is_contiguous() and imgs.numel() * imgs.element_size() >= (first + count) * self.raw_bytes
        sizes = (_sz * count)()
        k = lib.qb3x_encode_tiles(self.p, _vp(imgs.data_ptr() + first * self.raw_bytes), count, self.raw_bytes,
                                  _vp(self.dst.data_ptr() + first * self.pitch), self.pitch,
                                  _vp(self.index.data_ptr() + first * self.index_bytes) if self.index is not None else None,
                                  sizes, _stream_ptr())
        if k != count:
            raise RuntimeError(f"qb3x_encode_tiles coded {k} of {count} tiles: {last_error()}")
        for i in range(count):
            self.sizes[first + i] = sizes[i]
        return list(sizes)

    def decode(self, out, use_index=True):
        """decodes the n containers made by encode() into `out` (n * raw_bytes)."""
        if self.d is None:
            dims = (_sz * 3)()
            self.d = lib.qb3x_read_start_device(_vp(self.dst.data_ptr()), int(self.sizes[0]), dims, _stream_ptr())
            if not self.d:
                raise ValueError("tile 0 does not parse")
        k = lib.qb3x_decode_tiles(self.d, _vp(self.dst.data_ptr()), self.n, self.pitch, self.sizes, _vp(out.data_ptr()), self.raw_bytes,
                                  _vp(self.index.data_ptr()) if (use_index and self.index is not None) else None, _stream_ptr())
        if k != self.n:
            raise RuntimeError(f"qb3x_decode_tiles decoded {k} of {self.n} tiles: {last_error()}")
        return out


def profile_enable(on=True):
    """True / 1: every kernel; 2: skip the microsecond kernels (less event traffic in a timed loop); 3: the coding kernels (`*_units`) alone; False / 0: off."""
    lib.qb3x_profile_enable(int(on))


def profile_reset():
    lib.qb3x_profile_reset()


def profile_report():
    """{kernel name: (total ms, launches)} measured with HIP events on the launch stream."""
    buf = C.create_string_buffer(1024)
    lib.qb3x_profile_names(buf, 1024)
    out = {}
    for name in filter(None, buf.value.decode().split(",")):
        ms, cnt = C.c_double(), C.c_uint64()
        if lib.qb3x_profile_get(name.encode(), C.byref(ms), C.byref(cnt)):
            out[name] = (ms.value, cnt.value)
    return out
