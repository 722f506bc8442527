"""tools/check_seal.py W H B -- recompute the check of every chunk of the restart table of a level 2 container on the host (debug aid)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from qb3_amd import synth, device as qdev
w, h, b = (int(v) for v in sys.argv[1:4])
img = synth.generate(w, h, b, 0, "NOISY3", 3, device=torch.device("cuda", 0))
enc = qdev.DeviceEncoder(w, h, b, 0, mode=8, index_chunk=2)
dst, n, index = enc.encode(img)
c = dst[:min(int(n), 1 << 24)].cpu().numpy()
p = 11
while True:
    tag = bytes(c[p:p + 2])
    if tag == b"ix":
        ln = int(c[p + 2]) | int(c[p + 3]) << 8
        ent = c[p + 12:p + ln].astype(np.uint64)
        i = np.arange(len(ent), dtype=np.uint64)
        s = int((((ent + 1) * ((i * 0x9e3779b1 + 1) & 0xffffffff)) & 0xffffffff).sum() & 0xffffffff)
        f = (s ^ (s >> 16)) & 0xffff
        print("chunk at", p, "len", ln, "stored", int(c[p + 6]) | int(c[p + 7]) << 8, "computed", f)
        p += ln
    elif tag == b"zz": p += 4
    elif tag == b"DT": break
    else:
        ln = int(c[p + 2]) | int(c[p + 3]) << 8; print("chunk", tag, ln); p += 4 + ln
    if p + 12 > len(c): break
