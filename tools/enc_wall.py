"""tools/enc_wall.py W H B DT GEN MODE [N] -- elapsed time of a device-resident encode call (events on the caller's stream around N calls), and of
encode + decode from the container; what the per-kernel sums of kernel_probe.py cannot say once kernels of one call overlap"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qb3_amd import synth, device as qdev
w, h, b, dt = (int(v) for v in sys.argv[1:5]); gen, mode = sys.argv[5], int(sys.argv[6]); N = int(sys.argv[7]) if len(sys.argv) > 7 else 20
dev = torch.device("cuda", 0)
img = synth.generate(w, h, b, dt, gen, 3, device=dev)
raw = img.reshape(-1).view(torch.uint8)
enc = qdev.DeviceEncoder(w, h, b, dt, mode=mode, index_chunk=2)
dst, n, index = enc.encode(img)
first = dst[:n].clone()
dec = qdev.DeviceDecoder(dst, n)
out = torch.empty(raw.numel(), dtype=torch.uint8, device=dev)
dec.decode(dst, out=out)
assert torch.equal(out, raw), "round trip"
for what in ("encode", "encode+decode"):
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(N):
            dst, n, index = enc.encode(img)
            if what != "encode": dec.decode(dst, out=out)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        print(what, "%.3f ms a call" % ((t1 - t0) * 1e3 / N), flush=True)
assert torch.equal(dst[:n], first), "container changed"
assert torch.equal(out, raw), "round trip"
print("ok", int(n), "bytes")
