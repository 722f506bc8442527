"""qb3_amd/synth.py -- the synthetic rasters of SURVEY.md section 8(d), generated on the device with torch.

Same definitions as the anchor table, so a raster made here can be checked against the
published stream size / FNV of the reference without going through the host:

    r = splitmix64(seed + idx),  idx = (y*W + x)*bands + c
    GRAD = x + y + 17c            NOISY3 = GRAD + (r & 7)       LANDSAT16 = 7000 + 3x + 2y + 301c + (r & 63)
    DEM = 37(x+y) - 50000 + (r&63)   TERRACE = 1000*(x//16 + y//16 - 100)   RANDOM = r
    FEW = ((r mod 6) << (bits-6)) - (1 << (bits-4))      PALETTE = (splitmix64(77 + r mod 5) >> (66-bits)) | 1

torch is used for device memory and arithmetic only (int64 wraps like uint64; logical shifts are masked).
"""
import torch

TORCH_DTYPE = (torch.uint8, torch.int8, torch.int16, torch.int16, torch.int32, torch.int32, torch.int64, torch.int64)
TYPESIZE = (1, 1, 2, 2, 4, 4, 8, 8)


def _s64(v):
    v &= (1 << 64) - 1
    return v - (1 << 64) if v >> 63 else v


def _lsr(z, k):
    return (z >> k) & ((1 << (64 - k)) - 1)


def splitmix64(x):
    """x: int64 tensor holding uint64 bit patterns."""
    z = x + _s64(0x9E3779B97F4A7C15)
    z = (z ^ _lsr(z, 30)) * _s64(0xBF58476D1CE4E5B9)
    z = (z ^ _lsr(z, 27)) * _s64(0x94D049BB133111EB)
    return z ^ _lsr(z, 31)


def _umod(r, m):
    """r mod m for uint64 bit patterns held in int64 (m small)."""
    return ((_lsr(r, 1) % m) * 2 + (r & 1)) % m


def generate(w, h, bands, dtype, gen, seed, device="cuda", rows_per_chunk=None):
    """Returns a contiguous tensor of shape (h, w, bands) whose BYTES equal the generator's output truncated to
    the value width (the torch dtype is the same-width signed type where torch lacks the unsigned one)."""
    tsz = TYPESIZE[dtype]
    out = torch.empty((h, w, bands), dtype=TORCH_DTYPE[dtype], device=device)
    if rows_per_chunk is None:
        rows_per_chunk = max(1, (1 << 25) // (w * bands))
    xs = torch.arange(w, dtype=torch.int64, device=device).view(1, w, 1)
    cs = torch.arange(bands, dtype=torch.int64, device=device).view(1, 1, bands)
    for y0 in range(0, h, rows_per_chunk):
        y1 = min(h, y0 + rows_per_chunk)
        ys = torch.arange(y0, y1, dtype=torch.int64, device=device).view(-1, 1, 1)
        idx = (ys * w + xs) * bands + cs
        r = splitmix64(idx + _s64(seed)) if gen not in ("GRAD", "TERRACE", "CONST") else None
        if gen == "GRAD":
            v = xs + ys + 17 * cs
        elif gen == "NOISY3":
            v = xs + ys + 17 * cs + (r & 7)
        elif gen == "LANDSAT16":
            v = 7000 + 3 * xs + 2 * ys + 301 * cs + (r & 63)
        elif gen == "DEM":
            v = 37 * (xs + ys) - 50000 + (r & 63) + 0 * cs
        elif gen == "TERRACE":
            v = 1000 * (xs // 16 + ys // 16 - 100) + 0 * cs
        elif gen == "FEW":
            bits = 8 * tsz
            v = (_umod(r, 6) << (bits - 6)) - (1 << (bits - 4))
        elif gen == "PALETTE":
            bits = 8 * tsz
            v = _lsr(splitmix64(77 + _umod(r, 5)), 66 - bits) | 1
        elif gen == "RANDOM":
            v = r
        elif gen == "CONST":
            v = torch.full_like(idx, 42)
        else:
            raise ValueError(gen)
        if tsz == 8:
            out[y0:y1] = v
        else:       # truncate to the value width, reinterpret as the storage dtype
            m = (1 << (8 * tsz)) - 1
            v = v & m
            if TORCH_DTYPE[dtype] != torch.uint8:
                half = 1 << (8 * tsz - 1)
                v = torch.where(v >= half, v - (1 << (8 * tsz)), v)
            out[y0:y1] = v.to(TORCH_DTYPE[dtype])
    return out
