#!/usr/bin/env python3
"""bench.py -- QB3M_FTL encode+decode throughput of the MI355X-native QB3 library.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A step = one pass of the hot path over this rank's synthetic raster, resident in HBM before the clock
starts: qb3x_encode_device (container + out-of-band index) followed by qb3x_decode_device of that
container with that index.  With N > 1 every rank codes its own raster: tiles shard with NO data-path
collective, so the timed region holds only the coding path and scaling is weak.
The one exchange the workload has -- collecting the finished containers on rank 0, RCCL send/recv over
xGMI -- is done once after the timed region and reported on its own (`gather`): it is bound by the
peers' links into rank 0 (435 MB per peer), not by anything this library does.  Rank 0 prints ONE JSON line.

The workload at N = 1 is BASELINE.json configs[1]: 16384x16384, 3-band uint8, NOISY3 seed 2, QB3M_FTL.
`roofline` prices the dominant kernel against the HBM rate with the algorithmic bytes of SURVEY.md
section 8(d): bands*sizeof(T)*(1+rho) bytes per pixel (raw once + stream once), timed live with HIP events
the library records on the launch stream.  `cpu_baseline` times oracle/ (the CPU restatement, a "port")
on one host core over a bounded sample of the same raster.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
ANCHOR_C2 = 434747055       # reference stream size for 16384x16384x3 NOISY3 seed 2


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, default=16384, help="raster edge in pixels (default: BASELINE configs[1])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse on one GPU)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the QB3 block codec has no CPU fallback")
    if args.backend != "nccl":
        local_rank %= torch.cuda.device_count()         # rehearsal: ranks may share a GPU
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    import qb3_amd
    from qb3_amd import synth, device as qdev, tiles

    W = H = args.size
    bands, dtype = 3, qb3_amd.QB3_U8
    img = synth.generate(W, H, bands, dtype, "NOISY3", 2 + rank, device=dev)
    raw_bytes = img.numel() * img.element_size()
    enc = qdev.DeviceEncoder(W, H, bands, dtype, mode=qb3_amd.QB3M_FTL)
    out = torch.empty(raw_bytes, dtype=torch.uint8, device=dev)
    dec_cache = {}

    def step(check=False):
        dst, n, index = enc.encode(img)
        key = n
        if key not in dec_cache:            # header parse is host work done once per distinct container
            dec_cache.clear()
            dec_cache[key] = qdev.DeviceDecoder(dst[:64].cpu().numpy(), n)
        dec_cache[key].decode(dst, out=out, index=index)
        if check and not torch.equal(out.view(torch.uint8), img.view(torch.uint8).reshape(-1)):
            sys.exit("bench.py: decode(encode(x)) != x -- refusing to report a number")
        return n

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    nbytes = 0
    for i in range(max(1, args.warmup)):
        nbytes = step(check=(i == 0))
    if rank == 0 and args.size == 16384 and nbytes != ANCHOR_C2:
        sys.exit(f"bench.py: stream is {nbytes} bytes, the reference produces {ANCHOR_C2} -- not bit-identical")

    # HIP events around the long kernels only inside the timed region (level 2): an event pair costs more than the
    # microsecond kernels take; those are timed in two extra, untimed steps afterwards
    qdev.profile_reset()
    qdev.profile_enable(2)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    qdev.profile_enable(False)
    timed_kernels = set(qdev.profile_report())
    qdev.profile_enable(1)
    prof_small = {}
    before = qdev.profile_report()
    for _ in range(2):
        step()
    fence()
    qdev.profile_enable(False)
    for name, (ms, cnt) in qdev.profile_report().items():
        if name not in timed_kernels:
            prof_small[name] = (ms, cnt)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    prof = dict(before)             # the long kernels: exactly what the timed region saw
    prof.update(prof_small)         # the microsecond kernels: from the two untimed steps

    # ---- the workload's only exchange: containers to rank 0 (variable-size gather), timed on its own
    gather = None
    if world > 1:
        try:
            dst, n, _ = enc.encode(img)
            fence()
            g0 = time.perf_counter()
            bufs, size_lists = tiles.gather_streams(dst, [n], root=0)
            fence()
            gdt = time.perf_counter() - g0
            if rank == 0:
                total = sum(sum(sl) for sl in size_lists)
                ok = all(int(b.numel()) == sum(sl) for b, sl in zip(bufs, size_lists)) and bytes(bufs[-1][:4].cpu().numpy()) == b"QB3\x80"
                gather = {"ms": round(gdt * 1e3, 3), "bytes_at_root": total, "GBps_into_root": round((total - n) / gdt / 1e9, 1),
                          "containers_intact": bool(ok), "backend": "nccl (RCCL) send/recv" if args.backend == "nccl" else args.backend + " (rehearsal)"}
        except Exception as e:      # never lose the coding numbers to a transport problem
            gather = {"error": repr(e)[:200]}

    # ---- per-kernel rates (rank 0's kernels; every rank runs the same launches)
    stream_bytes = nbytes
    algo_bytes = raw_bytes + stream_bytes                       # read raw + write stream (or the reverse)
    kernels = {}
    for name, (ms, cnt) in prof.items():
        avg = ms / max(cnt, 1)
        kernels[name] = {"avg_ms": round(avg, 4), "launches": int(cnt), "GBps_algorithmic": round(algo_bytes / avg / 1e6, 1) if avg > 0 else None,
                         "in_timed_region": name in timed_kernels}
    enc_ms = sum(kernels[k]["avg_ms"] for k in ("enc_units", "enc_scan", "enc_concat", "enc_seams") if k in kernels)
    dec_ms = sum(kernels[k]["avg_ms"] for k in ("dec_index_serial", "dec_segments", "dec_units") if k in kernels)
    dom = max(kernels, key=lambda k: kernels[k]["avg_ms"]) if kernels else None
    traffic = valu = None
    try:
        if args.size != 16384:
            raise OSError("PMC traffic was collected for the 16384 workload only")
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            rec = json.load(f).get(dom, {})
        traffic = rec.get("hbm_bytes_per_launch")
        valu = rec.get("valu")
    except (OSError, ValueError):
        pass
    roofline = None
    if dom:
        achieved = algo_bytes / (kernels[dom]["avg_ms"] * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "algorithmic_bytes_per_launch": algo_bytes,
                    # where the time goes besides HBM (rocprofv3 SQ counters of the same workload, profiles/): the share of the
                    # kernel's duration in which the SIMDs issue vector instructions and the LDS pipe is busy -- integer bit
                    # packing, no MFMA; the rest is memory latency the resident waves do not cover
                    "valu": valu}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        line = {
            "metric": "MPixel/s encode+decode (QB3M_FTL, 8-bit 3-band)",
            "value": round(world * W * H / (dt / args.steps) / 1e6, 1),
            "unit": "MPixel/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{W}x{H}x3 uint8 NOISY3 (seed 2+rank) per GPU; QB3M_FTL qb3x_encode_device + indexed qb3x_decode_device"
                                   + ("; containers gathered on rank 0 with RCCL after the timed region (see gather)" if world > 1 else ""),
                       "stream_bytes": stream_bytes, "ratio": round(stream_bytes / raw_bytes, 4),
                       "bit_identical_to_reference": bool(args.size == 16384)},
            "encode_MPixel_s_kernels": round(W * H / enc_ms / 1e3, 1) if enc_ms else None,
            "decode_MPixel_s_kernels": round(W * H / dec_ms / 1e3, 1) if dec_ms else None,
            "kernels": kernels,
            "roofline": roofline,
            "cpu_baseline": cpu,
            "gather": gather,
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(target_s=12.0):
    """The CPU restatement (oracle/, bit-identical to the reference on the anchor table) on ONE host core:
    encode + decode of a 4096x4096x3 NOISY3 tile, repeated for about target_s seconds."""
    from oracle import pyoracle as o
    w = h = 4096
    img = o.generate(w, h, 3, 0, "NOISY3", 2)
    t_enc = t_dec = 0.0
    reps = 0
    t_start = time.perf_counter()
    while reps < 2 or time.perf_counter() - t_start < target_s:
        t0 = time.perf_counter()
        s = o.encode(img, 0, 8)
        t1 = time.perf_counter()
        out, _, _, _ = o.decode(s)
        t2 = time.perf_counter()
        t_enc += t1 - t0
        t_dec += t2 - t1
        reps += 1
        if out is None:
            raise RuntimeError("oracle failed to decode its own stream")
    px = w * h * reps
    return {"value": round(px / (t_enc + t_dec) / 1e6, 1), "unit": "MPixel/s", "cores": 1, "kind": "port",
            "sample": f"{reps} x (encode + decode) of a 4096x4096x3 uint8 NOISY3 tile, QB3M_FTL, single thread",
            "encode_MPixel_s": round(px / t_enc / 1e6, 1), "decode_MPixel_s": round(px / t_dec / 1e6, 1)}


if __name__ == "__main__":
    main()
