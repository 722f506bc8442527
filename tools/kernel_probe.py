#!/usr/bin/env python3
"""tools/kernel_probe.py W H BANDS DTYPE GEN MODE [steps] -- encode / decode kernel times (the library's HIP events) of one synthetic
raster through the device-pointer calls, with the out-of-band index and from the container's table.  A measuring aid."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import qb3_amd
    from qb3_amd import synth, device as qdev
    w, h, b, dt = (int(v) for v in sys.argv[1:5])
    gen, mode = sys.argv[5], int(sys.argv[6])
    steps = int(sys.argv[7]) if len(sys.argv) > 7 else 5
    dev = torch.device("cuda", 0)
    img = synth.generate(w, h, b, dt, gen, 3, device=dev)
    raw = img.reshape(-1).view(torch.uint8)
    enc = qdev.DeviceEncoder(w, h, b, dt, mode=mode, index_chunk=2)
    dst, n, index = enc.encode(img)
    dec = qdev.DeviceDecoder(dst, n)
    out = torch.empty(raw.numel(), dtype=torch.uint8, device=dev)
    for ix in (index, None):
        out.zero_()
        dec.decode(dst, out=out, index=ix)
        assert torch.equal(out, raw), "decode(encode(x)) != x"
    for name, fn in (("encode", lambda: enc.encode(img)), ("decode_index", lambda: dec.decode(dst, out=out, index=index)), ("decode_table", lambda: dec.decode(dst, out=out, index=None))):
        fn()
        qdev.profile_reset(); qdev.profile_enable(1)
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        qdev.profile_enable(False)
        rep = qdev.profile_report()
        print(name, {k: round(ms / max(c, 1), 4) for k, (ms, c) in sorted(rep.items())}, flush=True)
    print("container", int(n), "raw", raw.numel(), "ratio", round(int(n) / raw.numel(), 4))
    if os.environ.get("PROBE_PLAIN"):           # the same raster as the reference writes it (no table), decoded from the stream alone
        import time
        penc = qdev.DeviceEncoder(w, h, b, dt, mode=mode)
        pdst, pn, _ = penc.encode(img)
        pdec = qdev.DeviceDecoder(pdst, pn)
        out.zero_()
        pdec.decode(pdst, out=out, index=None)
        assert torch.equal(out, raw), "plain decode != x"
        qdev.profile_reset(); qdev.profile_enable(1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pdec.decode(pdst, out=out, index=None)
        torch.cuda.synchronize()
        dt_s = time.perf_counter() - t0
        qdev.profile_enable(False)
        print("decode_plain", {k: round(ms / max(c, 1), 3) for k, (ms, c) in sorted(qdev.profile_report().items())}, "wall ms", round(dt_s * 1e3, 2), "MPixel/s", round(w * h / dt_s / 1e6, 1),
              "status", qb3_amd.lib.qb3x_last_decode_status(pdec.p))


if __name__ == "__main__":
    main()
