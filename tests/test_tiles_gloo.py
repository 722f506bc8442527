"""Multi-GPU path on CPU: 2 ranks over gloo shard a tile set, each 'encodes' its tiles and rank 0 gathers the
variable-size containers with qb3_amd.tiles.gather_streams -- the same code the nccl (RCCL) path runs.
The encoder is injected: on CPU the containers come from the oracle (test infrastructure)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NTILES = 5


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pyoracle as o
    from qb3_amd import tiles
    first, count = tiles.shard_range(NTILES, rank, world)
    streams = [o.encode(o.generate(32, 24, 3, 0, "NOISY3", 1000 + t), 0, 8) for t in range(first, first + count)]
    sizes = [len(s) for s in streams]
    payload = torch.from_numpy(np.concatenate(streams)) if streams else torch.zeros(0, dtype=torch.uint8)
    bufs, size_lists = tiles.gather_streams(payload, sizes, root=0)
    if rank == 0:
        got = []
        for b, sl in zip(bufs, size_lists):
            off = 0
            for n in sl:
                got.append(b[off:off + n].numpy().copy())
                off += n
        q.put(got)
    dist.barrier()
    dist.destroy_process_group()


def _worker_pitched(rank, world, port, q, ntiles):
    """the layout qb3x_encode_tiles leaves (containers at a fixed pitch) through start_gather, two batches in flight"""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pyoracle as o
    from qb3_amd import tiles
    first, count = tiles.shard_range(ntiles, rank, world)
    pitch = 4096
    pending = []
    half = (count + 1) // 2
    for bi, (lo, hi) in enumerate(((0, half), (half, count))):           # two batches; a rank with no tiles still takes part
        streams = [o.encode(o.generate(32, 24, 3, 0, "NOISY3", 1000 + t), 0, 8) for t in range(first + lo, first + hi)]
        dst = torch.zeros(max(1, len(streams)) * pitch, dtype=torch.uint8)
        for i, s in enumerate(streams):
            dst[i * pitch:i * pitch + len(s)] = torch.from_numpy(s)
        # (one message per peer and batch; the second batch also tells the bound of the tile count: one all-gather instead of two)
        pending.append(tiles.start_gather(dst, pitch, [len(s) for s in streams], root=0, max_tiles=None if bi == 0 else ntiles))
    got = {}
    for b, pg in enumerate(pending):
        assert len(pg.reqs) <= (world - 1 if rank == 0 else 1), "more than one message per peer and batch"
        bufs, size_lists = pg.wait()
        if rank == 0:
            for r, (buf, sl, ol) in enumerate(zip(bufs, size_lists, pg.offset_lists)):
                f, c = tiles.shard_range(ntiles, r, world)
                h = (c + 1) // 2
                base = f + (0 if b == 0 else h)
                if r != 0 and sl:
                    assert ol == [sum(sl[:i]) for i in range(len(sl))] and buf.numel() >= sum(sl)     # packed back to back
                for i, n in enumerate(sl):
                    got[base + i] = buf[ol[i]:ol[i] + n].numpy().copy()
    if rank == 0:
        q.put([got[t] for t in sorted(got)])
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_is_a_partition():
    from qb3_amd import tiles
    for n in (0, 1, 5, 8, 256, 257):
        for world in (1, 2, 3, 8):
            spans = [tiles.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == n
            for (f0, c0), (f1, _) in zip(spans, spans[1:]):
                assert f0 + c0 == f1
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1


def test_gather_streams_two_ranks(oracle):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert len(got) == NTILES
    for t, s in enumerate(got):
        ref = oracle.encode(oracle.generate(32, 24, 3, 0, "NOISY3", 1000 + t), 0, 8)
        assert np.array_equal(s, ref), f"tile {t} changed in transit"


@pytest.mark.parametrize("world,ntiles", [(2, 5), (3, 2)], ids=["2ranks-5tiles", "3ranks-2tiles-one-rank-empty"])
def test_start_gather_pitched_batches(oracle, world, ntiles):
    """uneven shards, a rank without tiles, two batches in flight at once"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000 + world
    procs = [ctx.Process(target=_worker_pitched, args=(r, world, port, q, ntiles)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert len(got) == ntiles
    for t, s in enumerate(got):
        ref = oracle.encode(oracle.generate(32, 24, 3, 0, "NOISY3", 1000 + t), 0, 8)
        assert np.array_equal(s, ref), f"tile {t} changed in transit"
