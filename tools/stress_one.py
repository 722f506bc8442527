"""tools/stress_one.py W H B DT GEN MODE N -- the flow of tools/stress_shapes.py on one raster, with the failing iterations named"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import qb3_amd
from qb3_amd import synth, device as qdev
w, h, b, dt = (int(v) for v in sys.argv[1:5]); gen, mode, N = sys.argv[5], int(sys.argv[6]), int(sys.argv[7])
dev = torch.device("cuda", 0)
img = synth.generate(w, h, b, dt, gen, 3, device=dev)
raw = img.reshape(-1).view(torch.uint8)
enc = qdev.DeviceEncoder(w, h, b, dt, mode=mode, index_chunk=2)
out = torch.empty(raw.numel(), dtype=torch.uint8, device=dev)
isz = img.element_size() * b
first = None
for it in range(N):
    dst, n, index = enc.encode(img)
    cur = dst[:n].clone()
    if first is None: first = cur; index0 = None
    elif not torch.equal(cur, first):
        d = (cur != first).nonzero().flatten()
        print("iteration", it, "container differs", d.numel(), flush=True)
    dec = qdev.DeviceDecoder(dst, n)
    for name, ix in (("index", index), ("table", None)):
        out.zero_()
        sys.stderr.write("it %d %s\n" % (it, name)); sys.stderr.flush()
        dec.decode(dst, out=out, index=ix)
        if not torch.equal(out, raw):
            d = (out != raw).nonzero().flatten()
            px = d // isz
            print("iteration", it, name, "ndiff", d.numel(), "pixels", int(px[0]), "..", int(px[-1]), "rows", int(px[0]) // w, int(px[-1]) // w, "x", int(px[0]) % w, int(px[-1]) % w, flush=True)
            sys.stderr.write("   ^ BAD\n")
            out.zero_(); dec.decode(dst, out=out, index=ix)
            print("   again:", "ok" if torch.equal(out, raw) else "bad again", "; index now == index at iteration 0:", bool(torch.equal(index, index0)), flush=True)
    dec.close()
    if index0 is None: index0 = index.clone()
print("done", N)
