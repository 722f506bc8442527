"""C-ABI drop-in boundary, CPU side: the shared library loads, exports every symbol include/*.h declares,
and its host logic (handles, setters, header writer/parser, STORED path, error conventions) behaves like the
reference's (QB3encode.cpp:26-134, QB3decode.cpp:36-264), checked against the oracle's restatement.
No block coding happens here -- that needs the GPU and has no CPU fallback, which is also asserted."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    names = []
    for hdr in ("QB3.h", "qb3x.h"):
        text = open(os.path.join(ROOT, "include", hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names += re.findall(r"\b(qb3x?_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_every_declared_symbol_is_exported(qb3):
    names = declared_functions()
    assert len(names) >= 21 + 8
    for n in names:
        assert hasattr(qb3.lib, n), f"{n} is declared in include/ but not exported by libQB3.so"
    # the reference's 21 entry points (QB3.h:85-162)
    for n in ("qb3_create_encoder qb3_destroy_encoder qb3_reset_encoder qb3_set_encoder_coreband qb3_set_encoder_quanta "
              "qb3_max_encoded_size qb3_set_encoder_mode qb3_set_encoder_stride qb3_encode qb3_get_encoder_state "
              "qb3_read_start qb3_read_info qb3_read_data qb3_destroy_decoder qb3_decoded_size qb3_get_type "
              "qb3_set_decoder_stride qb3_get_mode qb3_get_quanta qb3_get_order qb3_get_coreband").split():
        assert n in names


def test_binding_covers_all_declarations(qb3):
    assert set(declared_functions()) <= set(qb3.EXPORTED)


def test_create_encoder_validation(qb3):
    L = qb3.lib
    for bad in ((0, 4, 1, 0), (4, 0, 1, 0), (65537, 4, 1, 0), (4, 4, 0, 0), (4, 4, 17, 0), (4, 4, 1, 8)):
        assert not L.qb3_create_encoder(*bad)
    p = L.qb3_create_encoder(65536, 65536, 16, 7)
    assert p
    L.qb3_destroy_encoder(p)


@pytest.mark.parametrize("w,h,b,dt", [(512, 512, 3, 0), (509, 515, 3, 0), (8192, 8192, 8, 2), (4096, 4096, 1, 7), (16384, 16384, 3, 0), (5, 7, 16, 5)])
def test_max_encoded_size_matches(qb3, oracle, w, h, b, dt):
    p = qb3.lib.qb3_create_encoder(w, h, b, dt)
    e = oracle.Encoder(w, h, b, dt)
    assert qb3.lib.qb3_max_encoded_size(p) == e.max_size()
    qb3.lib.qb3_destroy_encoder(p)
    if (w, h, b, dt) == (16384, 16384, 3, 0):
        assert e.max_size() == 912262144


@pytest.mark.parametrize("w,h,b,dt,mode,lens", [(512, 512, 3, 0, 8, True), (509, 259, 1, 0, 4, True), (640, 384, 4, 0, 0, True), (16384, 16384, 3, 0, 8, True),
                                                  (512, 512, 3, 0, 7, False), (256, 256, 5, 0, 8, False), (256, 256, 3, 2, 8, True), (256, 128, 5, 2, 8, False), (256, 256, 1, 7, 5, False),
                                                  (8192, 8192, 8, 2, 4, True), (300, 200, 4, 3, 8, True), (256, 256, 16, 2, 8, False), (700, 300, 1, 2, 8, True), (300, 200, 3, 2, 4, True), (320, 240, 6, 3, 8, True)])
def test_room_for_the_restart_table(qb3, w, h, b, dt, mode, lens):
    """qb3_max_encoded_size grows by the room for the table's chunks while qb3x_set_encoder_index_chunk is on (host logic, no
    GPU).  The bound does not depend on the mode -- the reference's callers size their buffer BEFORE qb3_set_encoder_mode
    (reference cqb3.cpp:405-464) -- so it is the room of the largest table any mode writes for the raster: FTL/BASE streams,
    level 1 -- an entry per segment of 6 + bands * (1 + size) bytes; level 2 -- 80 more bytes an entry where the 8-bit
    lane-per-block decoder applies, 160 more for 16-bit rasters of four or eight bands; 8-bit common-factor streams of 1/3/4
    bands -- an entry per 64 blocks of 6 + 3 * bands + 3 * 64 bytes at either level; chunks of at most 65535 bytes, each with a
    12-byte head and a 4-byte pad chunk"""
    L = qb3.lib
    grow = {}
    for m in (mode, 8, 5, 4):
        p = L.qb3_create_encoder(w, h, b, dt)
        base = L.qb3_max_encoded_size(p)            # (sized before the mode is set, as the reference's callers do)
        L.qb3_set_encoder_mode(p, m)
        assert L.qb3_max_encoded_size(p) == base
        L.qb3x_set_encoder_index_chunk(p, 1)
        one = L.qb3_max_encoded_size(p) - base
        L.qb3x_set_encoder_index_chunk(p, 2)
        two = L.qb3_max_encoded_size(p) - base
        L.qb3x_set_encoder_index_chunk(p, 0)
        assert L.qb3_max_encoded_size(p) == base
        L.qb3_destroy_encoder(p)
        grow[m] = (one, two)
    assert len(set(grow.values())) == 1, grow     # the same room whatever the mode
    one, two = grow[mode]
    assert 0 < one <= two
    nblocks = ((w + 3) // 4) * ((h + 3) // 4)

    def room(nseg, entry):
        per_chunk = (65535 - 12) // entry
        return nseg * entry + ((nseg + per_chunk - 1) // per_chunk) * 16
    if dt <= 1 and b in (1, 3, 4):                  # the common-factor table of 8-bit grey / RGB / RGBA is the largest at either level
        assert one == two == room((nblocks + 63) // 64, 6 + 3 * b + 3 * 64)
    elif lens:
        bg16 = b if b <= 4 else (4 if b % 4 == 0 else 2)    # 16-bit: bands a lane of the decoder's wave owns
        per_seg = 64 // (b // bg16)
        nseg = (nblocks + per_seg - 1) // per_seg
        fixed = 6 + b * (1 + 2)
        assert one >= room(nseg, fixed) and two >= room(nseg, fixed + (80 if b == 1 else 160))


@pytest.mark.parametrize("w,h,b,dt,mode", [(4096, 4096, 1, 5, 8), (520, 300, 1, 7, 4), (160, 120, 5, 4, 8)])
def test_room_for_unit_lengths_of_wide_types(qb3, w, h, b, dt, mode):
    """32/64-bit rasters: a level 2 table of an FTL/BASE stream carries twelve bits per unit on top of the level 1 table; the room
    is the largest any mode needs (the common-factor modes' tables have no lengths but closer entries) (host logic, no GPU)"""
    L = qb3.lib
    p = L.qb3_create_encoder(w, h, b, dt)
    L.qb3_set_encoder_mode(p, mode)
    base = L.qb3_max_encoded_size(p)
    L.qb3x_set_encoder_index_chunk(p, 1)
    one = L.qb3_max_encoded_size(p) - base
    L.qb3x_set_encoder_index_chunk(p, 2)
    two = L.qb3_max_encoded_size(p) - base
    L.qb3_destroy_encoder(p)
    units = ((w + 3) // 4) * ((h + 3) // 4) * b
    assert 0 < one <= two and units * 12 // 8 <= two <= units * 12 // 8 + units * 4 + 4096


def test_setters_match_oracle(qb3, oracle):
    L = qb3.lib
    p = L.qb3_create_encoder(16, 16, 5, 2)
    e = oracle.Encoder(16, 16, 5, 2)
    for m in (8, 9, -1, 300, 4, 7, 2, 0, 8):
        assert L.qb3_set_encoder_mode(p, m) == e.set_mode(m)
    for cb in ([1, 1, 1, 3, 4], [4, 4, 4, 4, 4], [9, 0, 1, 2, 3], [1, 2, 3, 4, 0]):
        arr = (C.c_size_t * 5)(*cb)
        assert L.qb3_set_encoder_coreband(p, 5, arr)
        assert list(arr) == e.set_coreband(cb)
    assert not L.qb3_set_encoder_coreband(p, 4, (C.c_size_t * 4)(0, 0, 0, 0))
    for dt in range(8):
        q = L.qb3_create_encoder(8, 8, 1, dt)
        o = oracle.Encoder(8, 8, 1, dt)
        for v in (0, 1, 2, 127, 128, 255, 256, 32767, 32768, 65535, 65536, 2**31 - 1, 2**31, 2**32, 2**63 - 1, 2**63):
            assert L.qb3_set_encoder_quanta(q, v, False) == o.set_quanta(v), (dt, v)
        L.qb3_destroy_encoder(q)
    L.qb3_destroy_encoder(p)


def test_tiny_images_are_stored_without_gpu(qb3, oracle):
    """w*h <= 16 never reaches the block coder (reference QB3encode.cpp:490): host only, must equal the oracle"""
    for (w, h, b, dt) in ((4, 4, 3, 0), (1, 1, 1, 7), (16, 1, 2, 3), (2, 8, 4, 5)):
        img = oracle.generate(w, h, b, dt, "RANDOM", 3)
        ref = oracle.encode(img, dt, 8)
        got = qb3.encode(img, dt, 8)
        assert np.array_equal(got, ref) and got[10] == 255
        out, dims, t, mode = qb3.decode(got)
        assert dims == (w, h, b) and t == dt and mode == 255
        assert np.array_equal(out, img.view(np.uint8).ravel())


@pytest.mark.parametrize("mode", range(9))
def test_header_parse_matches_oracle(qb3, oracle, mode):
    L = qb3.lib
    for (w, h, b, dt, q) in ((33, 21, 3, 0, 1), (16, 16, 8, 2, 1), (20, 20, 1, 5, 7), (16, 16, 4, 1, 300 if False else 3)):
        img = oracle.generate(w, h, b, dt, "NOISY3", 4)
        s = oracle.encode(img, dt, mode, quanta=q)
        dims = (C.c_size_t * 3)()
        p = L.qb3_read_start(s.ctypes.data, s.size, dims)
        assert p and tuple(dims) == (w, h, b)
        assert L.qb3_get_mode(p) == -1 and L.qb3_get_quanta(p) == 0 and L.qb3_get_order(p) == 0   # before read_info
        cb = (C.c_size_t * 16)()
        assert not L.qb3_get_coreband(p, cb)
        assert L.qb3_read_info(p)
        od = oracle.lib.qb3o_decoder_new(s.ctypes.data, s.size, (C.c_size_t * 3)())
        assert oracle.lib.qb3o_read_info(od)
        assert L.qb3_get_mode(p) == oracle.lib.qb3o_decoder_mode(od) == s[10]
        assert L.qb3_get_type(p) == dt
        assert L.qb3_get_quanta(p) == oracle.lib.qb3o_decoder_quanta(od)
        assert L.qb3_get_order(p) == oracle.lib.qb3o_decoder_order(od)
        assert L.qb3_decoded_size(p) == img.nbytes
        ocb = (C.c_size_t * 16)()
        assert L.qb3_get_coreband(p, cb) and oracle.lib.qb3o_decoder_coreband(od, ocb)
        assert list(cb)[:b] == list(ocb)[:b]
        assert not L.qb3_read_info(p)               # wrong stage the second time (QB3decode.cpp:178)
        oracle.lib.qb3o_free(od)
        L.qb3_destroy_decoder(p)


def test_read_start_rejections(qb3, oracle):
    L = qb3.lib
    img = oracle.generate(16, 16, 3, 0, "NOISY3", 1)
    s = oracle.encode(img, 0, 8)
    dims = (C.c_size_t * 3)()
    assert not L.qb3_read_start(s.ctypes.data, 14, dims)            # shorter than 15 bytes
    assert not L.qb3_read_start(s.ctypes.data, s.size, None)
    for off, val in ((0, 0x51 ^ 1), (3, 0x81), (8, 16), (9, 8), (10, 9), (11, 0x80 | ord("C"))):
        t = s.copy()
        t[off] = val
        assert not L.qb3_read_start(t.ctypes.data, t.size, dims), (off, val)
    t = s.copy()
    t[11:13] = np.frombuffer(b"XY", np.uint8)                        # unknown upper-case chunk -> QB3E_UNKN
    p = L.qb3_read_start(t.ctypes.data, t.size, dims)
    assert p and not L.qb3_read_info(p)
    assert L.qb3_read_data(p, t.ctypes.data) == 0
    L.qb3_destroy_decoder(p)


def test_block_coding_fails_loudly_without_gpu(qb3, oracle):
    """No CPU fallback: on a box without a HIP device qb3_encode / qb3_read_data return 0 and say why."""
    if qb3.lib.qb3x_device_count() > 0:
        pytest.skip("a GPU is present")
    img = oracle.generate(32, 32, 3, 0, "NOISY3", 1)
    p = qb3.lib.qb3_create_encoder(32, 32, 3, 0)
    dst = np.zeros(qb3.lib.qb3_max_encoded_size(p), np.uint8)
    assert qb3.lib.qb3_encode(p, img.ctypes.data, dst.ctypes.data) == 0
    assert qb3.lib.qb3_get_encoder_state(p) == 255                   # QB3E_LIBERR
    assert "no usable HIP device" in qb3.last_error()
    qb3.lib.qb3_destroy_encoder(p)
    s = oracle.encode(img, 0, 8)
    with pytest.raises(RuntimeError):
        qb3.decode(s)


def test_headers_compile_as_c_and_cpp_and_link(tmp_path):
    """include/QB3.h is the drop-in header: a C caller and a C++ caller that use every reference entry point must compile
    against it and link against qb3_amd/libQB3.so (nothing is run: no GPU here)"""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    inc, libdir = os.path.join(root, "include"), os.path.join(root, "qb3_amd")
    src = r'''
#include "QB3.h"
#include "qb3x.h"
#include <stddef.h>
int main(void) {
    size_t dims[3], map[3] = {1, 1, 2};
    unsigned char in[48] = {0}, out[2048];
    encsp e = qb3_create_encoder(4, 4, 3, QB3_U8);
    if (!e) return 1;
    qb3_set_encoder_mode(e, QB3M_FTL);
    qb3_set_encoder_coreband(e, 3, map);
    qb3_set_encoder_quanta(e, 1, 0);
    qb3_set_encoder_stride(e, 0);
    size_t n = qb3_max_encoded_size(e) <= sizeof(out) ? qb3_encode(e, in, out) : 0;
    int st = qb3_get_encoder_state(e);
    qb3_reset_encoder(e);
    qb3_destroy_encoder(e);
    decsp d = n ? qb3_read_start(out, n, dims) : 0;
    if (d) {
        qb3_read_info(d);
        (void)qb3_get_type(d); (void)qb3_get_mode(d); (void)qb3_get_quanta(d); (void)qb3_get_order(d);
        (void)qb3_get_coreband(d, map); (void)qb3_decoded_size(d);
        qb3_set_decoder_stride(d, 0);
        qb3_read_data(d, in);
        qb3_destroy_decoder(d);
    }
    (void)qb3x_device_count(); (void)qb3x_last_error();
    return st;
}
'''
    for compiler, name, std in (("gcc", "caller.c", "-std=c99"), ("g++", "caller.cpp", "-std=c++11")):
        path = tmp_path / name
        path.write_text(src)
        exe = tmp_path / (name + ".out")
        r = subprocess.run([compiler, std, "-Wall", "-Werror", "-I", inc, str(path), "-L", libdir, "-lQB3",
                            "-Wl,-rpath," + libdir, "-o", str(exe)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def test_pool_and_status_entry_points_without_a_device(qb3):
    """qb3x_trim and qb3x_last_decode_status (include/qb3x.h, no counterpart in the reference) are harmless without a GPU and
    with no handle: nothing pooled, nothing to report"""
    qb3.lib.qb3x_trim()
    qb3.lib.qb3x_trim()
    assert qb3.lib.qb3x_last_decode_status(None) == 0


def test_header_only_handle_refuses_the_host_pointer_call(qb3, oracle):
    """A handle made by qb3x_read_start from a copy of the container's head (the stream itself elsewhere: in device memory)
    holds fewer bytes than the container has; qb3_read_data on it must not read the stream past that copy:
    QB3E_EINV, nothing touched -- decided before any device is asked for, so also here without a GPU."""
    L = qb3.lib
    img = oracle.generate(64, 64, 3, 0, "NOISY3", 1)
    s = oracle.encode(img, 0, 8)
    head = s[:64].copy()                            # header + "DT" + the first stream bytes
    dims = (C.c_size_t * 3)()
    p = L.qb3x_read_start(head.ctypes.data, head.size, s.size, dims)
    assert p and L.qb3_read_info(p)
    out = np.full(64 * 64 * 3, 0xa5, np.uint8)
    assert L.qb3_read_data(p, out.ctypes.data) == 0
    assert (out == 0xa5).all()
    L.qb3_destroy_decoder(p)
    # ... and for a raw-stored container, whose "decode" is a host copy of s_size bytes
    e = oracle.Encoder(4, 4, 1, 0)
    tiny = e.encode(np.arange(16, dtype=np.uint8).reshape(4, 4, 1))
    assert tiny[10] == 255 and tiny.size == 13 + 16
    p = L.qb3x_read_start(tiny.ctypes.data, 20, tiny.size, dims)
    assert p and L.qb3_read_info(p)
    out = np.full(16, 0xa5, np.uint8)
    assert L.qb3_read_data(p, out.ctypes.data) == 0 and (out == 0xa5).all()
    L.qb3_destroy_decoder(p)
    # the whole container on the host: the stored copy works (no GPU needed)
    p = L.qb3_read_start(tiny.ctypes.data, tiny.size, dims)
    assert p and L.qb3_read_info(p)
    assert L.qb3_read_data(p, out.ctypes.data) == 16 and (out == np.arange(16)).all()
    L.qb3_destroy_decoder(p)


def _with_table(stream, entries_per_chunk, nchunks, entry_bytes, blocks, junk=None):
    """a container with a (fake but regular) version 2 restart table in front of "DT": nchunks "ix" + "zz" pairs"""
    at = 11
    while bytes(stream[at:at + 2]) in (b"CB", b"QV", b"SC"):
        at += 4 + int(stream[at + 2]) + 256 * int(stream[at + 3])
    assert bytes(stream[at:at + 2]) == b"DT"
    chunks = b""
    for c in range(nchunks):
        ln = 12 + entries_per_chunk * entry_bytes
        head = b"ix" + ln.to_bytes(2, "little") + bytes([2, 0, 0, 0]) + blocks.to_bytes(4, "little")
        body = bytes(entries_per_chunk * entry_bytes)
        if junk is not None and c == junk:
            head = b"iy" + head[2:]                 # not a table chunk: an ignorable chunk of the same length
        chunks += head + body + b"zz\x04\x00"
    return np.frombuffer(bytes(stream[:at]) + chunks + bytes(stream[at:]), np.uint8).copy()


def test_table_chunks_are_walked_one_by_one_on_the_host(qb3, oracle):
    """With the whole container on the host qb3_read_info walks every chunk head in front of "DT" like the reference's parser
    (QB3decode.cpp:176-264) instead of stepping over a regular table in one go: a foreign chunk in the middle of
    the run of table chunks means there is no usable table -- and is found; the stream still parses."""
    L = qb3.lib
    w = h = 256
    img = oracle.generate(w, h, 1, 0, "NOISY3", 1)           # (one band: no CB chunk, "DT" right behind the fixed header)
    s = oracle.encode(img, 0, 8)
    nseg = (w // 4) * (h // 4) // 64
    E = 6 + 1 * 2
    per = 16
    assert nseg % per == 0
    dims = (C.c_size_t * 3)()
    good = _with_table(s, per, nseg // per, E, 64)
    bad = _with_table(s, per, nseg // per, E, 64, junk=1)
    for t, usable in ((good, True), (bad, False)):
        p = L.qb3_read_start(t.ctypes.data, t.size, dims)
        assert p and L.qb3_read_info(p), "the reference's parser steps over ignorable chunks"
        assert L.qb3_get_mode(p) == 8
        assert L.qb3x_decoder_table_entries(p) == (nseg if usable else 0)
        L.qb3_destroy_decoder(p)
    # a handle that has only the head (the table's middle is not on the host) may step over a regular table: it cannot see the junk
    p = L.qb3x_read_start(bad.ctypes.data, 64, bad.size, dims)
    assert p
    L.qb3_read_info(p)
    L.qb3_destroy_decoder(p)
