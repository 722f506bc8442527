"""tools/stress_shapes.py [iterations] -- repeated encode / decode of rasters of the lane-per-unit shapes: is the container the same every time, does it decode (with
the index, from its table) to the input every time.  A debugging aid for intermittent failures."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import qb3_amd
from qb3_amd import synth, device as qdev
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda", 0)
SHAPES = [(4096, 4096, 2, 7, "DEM", 8), (4096, 4096, 3, 7, "DEM", 0), (4096, 4096, 2, 5, "DEM", 8), (4096, 4096, 5, 0, "NOISY3", 8), (4096, 4096, 7, 2, "LANDSAT16", 4),
          (4096, 4096, 3, 5, "DEM", 7), (4096, 4096, 3, 7, "DEM", 7), (4096, 4096, 5, 0, "NOISY3", 5),
          # the lane-per-block families
          (8192, 8192, 3, 0, "NOISY3", 8), (8192, 8192, 3, 0, "NOISY3", 7), (4096, 4096, 8, 2, "LANDSAT16", 4), (4096, 4096, 8, 2, "LANDSAT16", 5),
          (4096, 4096, 1, 5, "DEM", 8), (4096, 4096, 1, 7, "DEM", 7), (4096, 4096, 1, 3, "DEM", 7), (4096, 4096, 4, 0, "NOISY3", 7)]
for (w, h, b, dt, gen, mode) in SHAPES:
    img = synth.generate(w, h, b, dt, gen, 3, device=dev)
    raw = img.reshape(-1).view(torch.uint8)
    enc = qdev.DeviceEncoder(w, h, b, dt, mode=mode, index_chunk=2)
    out = torch.empty(raw.numel(), dtype=torch.uint8, device=dev)
    first = None
    bad_enc = bad_ix = bad_tab = 0
    for it in range(N):
        dst, n, index = enc.encode(img)
        cur = dst[:n].clone()
        if first is None: first = cur
        elif cur.numel() != first.numel() or not torch.equal(cur, first): bad_enc += 1
        dec = qdev.DeviceDecoder(dst, n)
        out.zero_(); dec.decode(dst, out=out, index=index)
        if not torch.equal(out, raw):
            bad_ix += 1
            if bad_ix == 1:
                d = (out != raw).nonzero().flatten()
                print("   index decode: first diff byte", int(d[0]), "ndiff", d.numel(), "last", int(d[-1]), "status", qb3_amd.lib.qb3x_last_decode_status(dec.p), flush=True)
        out.zero_(); dec.decode(dst, out=out, index=None)
        if not torch.equal(out, raw):
            bad_tab += 1
            if bad_tab == 1:
                d = (out != raw).nonzero().flatten()
                print("   table decode: first diff byte", int(d[0]), "ndiff", d.numel(), "last", int(d[-1]), "status", qb3_amd.lib.qb3x_last_decode_status(dec.p), flush=True)
        dec.close()
    print((w, h, b, dt, gen, mode), "iterations", N, "containers that differ from the first", bad_enc, "bad decodes: index", bad_ix, "table", bad_tab, flush=True)
