#!/usr/bin/env python3
"""tools/hostapi_bench.py -- the reference's own entry points on HOST buffers (qb3_encode / qb3_read_data, reference QB3.h:110,141;
upload and download included) on the headline raster (16384 x 16384 x 3 uint8 NOISY3 seed 2): best of N, and the container
compared byte for byte with the one the device-pointer call writes.  What bench.py prints as `host_api_ms`, alone."""
import argparse
import sys
import os
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=16384)
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--mode", type=int, default=8)
    ap.add_argument("--levels", default="0,2")
    args = ap.parse_args()
    import numpy as np
    import torch
    import qb3_amd
    from qb3_amd import synth, device as qdev
    L = qb3_amd.lib
    W = H = args.size
    dev = torch.device("cuda", 0)
    img = synth.generate(W, H, 3, qb3_amd.QB3_U8, "NOISY3", 2, device=dev)
    host = np.ascontiguousarray(img.cpu().numpy()).reshape(H, W, 3)
    for level in [int(v) for v in args.levels.split(",")]:
        enc = qdev.DeviceEncoder(W, H, 3, qb3_amd.QB3_U8, mode=args.mode, want_index=False, index_chunk=level)
        dst_d, n_d, _ = enc.encode(img)
        want = dst_d[:n_d].cpu().numpy()
        del enc, dst_d
        p = L.qb3_create_encoder(W, H, 3, qb3_amd.QB3_U8)
        L.qb3_set_encoder_mode(p, args.mode)
        if level:
            L.qb3x_set_encoder_index_chunk(p, level)
        dst = np.empty(L.qb3_max_encoded_size(p), dtype=np.uint8)
        t_enc = []
        for _ in range(args.reps):
            L.qb3_reset_encoder(p)
            L.qb3_set_encoder_mode(p, args.mode)
            dst[:64] = 0
            t0 = time.perf_counter()
            n = L.qb3_encode(p, host.ctypes.data, dst.ctypes.data)
            t_enc.append(time.perf_counter() - t0)
        L.qb3_destroy_encoder(p)
        same = bool(n == want.size and np.array_equal(dst[:n], want))
        back = np.empty(W * H * 3, dtype=np.uint8)
        t_dec = []
        m = 0
        for _ in range(args.reps):
            dims = (qb3_amd._sz * 3)()
            d = L.qb3_read_start(dst.ctypes.data, n, dims)
            ok = d and L.qb3_read_info(d)
            back[:64] = 0
            t0 = time.perf_counter()
            m = L.qb3_read_data(d, back.ctypes.data) if ok else 0
            t_dec.append(time.perf_counter() - t0)
            L.qb3_destroy_decoder(d)
            if level == 0 and args.size > 8192:
                break                                   # (a plain container of this size: the walk, a third of a second)
        exact = bool(m == back.size and np.array_equal(back, host.ravel()))
        print(f"level {level}: qb3_encode ms {[round(t * 1e3, 2) for t in t_enc]} container {n} same_as_device_call {same}; "
              f"qb3_read_data ms {[round(t * 1e3, 2) for t in t_dec]} exact {exact}", flush=True)
        if not (same and exact):
            sys.exit(1)


if __name__ == "__main__":
    main()
