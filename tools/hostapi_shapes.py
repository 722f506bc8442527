"""tools/hostapi_shapes.py -- qb3_encode / qb3_read_data wall times (host buffers, the link included; buffers allocated and touched
beforehand) of large rasters of several shapes.  A measuring aid."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import qb3_amd
from oracle import pyoracle as o
L = qb3_amd.lib
for (w, h, b, dt, gen, mode) in [(8192, 8192, 2, 5, "DEM", 8), (8192, 8192, 7, 2, "LANDSAT16", 4), (8192, 8192, 8, 2, "LANDSAT16", 5), (8192, 8192, 5, 0, "NOISY3", 8), (16384, 16384, 3, 0, "NOISY3", 8)]:
    img = np.ascontiguousarray(o.generate(w, h, b, dt, gen, 5))
    p = L.qb3_create_encoder(w, h, b, dt)
    L.qb3_set_encoder_mode(p, mode)
    L.qb3x_set_encoder_index_chunk(p, 2)
    arr = (C.c_size_t * b)(*range(b))
    if b not in (1, 3, 4): L.qb3_set_encoder_coreband(p, b, arr)
    dst = np.zeros(L.qb3_max_encoded_size(p), dtype=np.uint8)
    te = 1e9
    for _ in range(4):
        L.qb3_reset_encoder(p); L.qb3_set_encoder_mode(p, mode)
        t0 = time.perf_counter(); n = L.qb3_encode(p, img.ctypes.data, dst.ctypes.data); te = min(te, time.perf_counter() - t0)
    assert n
    L.qb3_destroy_encoder(p)
    dims = (C.c_size_t * 3)()
    d = L.qb3_read_start(dst.ctypes.data, n, dims)
    assert d and L.qb3_read_info(d)
    out = np.zeros(img.nbytes, dtype=np.uint8)
    td = 1e9
    for _ in range(4):
        t0 = time.perf_counter(); got = L.qb3_read_data(d, out.ctypes.data); td = min(td, time.perf_counter() - t0)
    assert got == out.size and np.array_equal(out, img.view(np.uint8).ravel())
    L.qb3_destroy_decoder(d)
    print((w, h, b, dt, gen, mode), "raw MB", img.nbytes >> 20, "container MB", n >> 20, "qb3_encode ms %.1f  qb3_read_data ms %.1f" % (te * 1e3, td * 1e3), flush=True)
    del img, dst, out
