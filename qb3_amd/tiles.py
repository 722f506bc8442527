"""qb3_amd/tiles.py -- independent tiles across the GPUs of one node.

The path shards by image tile (every tile is its own QB3 stream): no collective while
coding.  The only exchange is the variable-size gather of the finished containers to one rank, done with
point-to-point sends (RCCL send/recv on the `nccl` backend; `gloo` on CPU in the tests), one message per
peer and batch, sized to the bytes actually produced -- over xGMI each peer owns its own link to the root, so the
gather is bound by the per-link rate, not by a ring.
"""
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world):
    """Contiguous, balanced split of n_items; returns (first, count) for `rank`."""
    base, extra = divmod(n_items, world)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def gather_streams(payload, sizes, root=0, group=None):
    """Gather variable-size byte payloads on `root`.

    payload: 1-D uint8 tensor holding this rank's containers back to back (only the first sum(sizes) bytes count)
    sizes:   list[int], container sizes of this rank's tiles, in tile order
    Returns on root: (list of per-rank uint8 tensors, list of per-rank size lists); elsewhere (None, None).
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if payload.is_cuda and dist.get_backend(group) != "nccl":
        payload = payload.cpu()         # gloo moves host memory (CPU tests, single-GPU rehearsals); RCCL moves HBM to HBM
    device = payload.device
    counts = torch.tensor([len(sizes)], dtype=torch.int64, device=device)
    all_counts = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(all_counts, counts, group=group)
    maxn = int(max(int(c.item()) for c in all_counts))
    mine = torch.zeros(maxn, dtype=torch.int64, device=device)
    if sizes:
        mine[:len(sizes)] = torch.tensor(sizes, dtype=torch.int64, device=device)
    all_sizes = [torch.zeros(maxn, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(all_sizes, mine, group=group)
    size_lists = [[int(v) for v in s[:int(c.item())].tolist()] for s, c in zip(all_sizes, all_counts)]
    total = sum(sizes)
    # one batched group of point-to-point operations: every peer sends exactly the bytes it produced
    ops, bufs = [], None
    if rank == root:
        bufs = []
        for r in range(world):
            nbytes = sum(size_lists[r])
            if r == root:
                bufs.append(payload[:total])
                continue
            b = torch.empty(nbytes, dtype=torch.uint8, device=device)
            bufs.append(b)
            if nbytes:
                ops.append(dist.P2POp(dist.irecv, b, r, group))
    elif total:
        ops.append(dist.P2POp(dist.isend, payload[:total].contiguous(), root, group))
    if ops:
        for q in dist.batch_isend_irecv(ops):
            q.wait()
    return (bufs, size_lists) if rank == root else (None, None)


class PendingGather:
    """A gather in flight (start_gather): wait() returns, on the root, ([per-rank buffer], [per-rank size list]) with tile
    t of rank r at buffer[r][offsets[r][t] : offsets[r][t] + sizes[r][t]] (`offset_lists`: the peers' tiles arrive packed
    back to back, the root's own stay where the encoder put them, at t * pitch); elsewhere (None, None)."""

    def __init__(self, reqs, bufs, size_lists, offset_lists, keep):
        self.reqs, self.bufs, self.size_lists, self.offset_lists, self._keep = reqs, bufs, size_lists, offset_lists, keep

    def wait(self):
        for q in self.reqs:
            q.wait()
        self.reqs = []
        self._keep = []
        return self.bufs, self.size_lists


def _exchange_sizes(sizes, device, group, max_tiles):
    """every rank's list of container sizes.  max_tiles (an upper bound of any rank's tile count, known to all): ONE small
    all-gather of [count, sizes...]; without it the counts go first (two)."""
    world = dist.get_world_size(group)
    if max_tiles is None:
        counts = torch.tensor([len(sizes)], dtype=torch.int64, device=device)
        all_counts = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
        dist.all_gather(all_counts, counts, group=group)
        max_tiles = max(1, int(max(int(c.item()) for c in all_counts)))
    if len(sizes) > max_tiles:
        raise ValueError("start_gather: more tiles than max_tiles")
    mine = torch.zeros(max_tiles + 1, dtype=torch.int64, device=device)
    mine[0] = len(sizes)
    if sizes:
        mine[1:1 + len(sizes)] = torch.tensor([int(v) for v in sizes], dtype=torch.int64, device=device)
    everyone = [torch.zeros(max_tiles + 1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(everyone, mine, group=group)
    out = []
    for v in everyone:
        v = v.tolist()
        out.append([int(x) for x in v[1:1 + int(v[0])]])
    return out


def start_gather(dst, pitch, sizes, root=0, group=None, recv_bufs=None, send_buf=None, max_tiles=None):
    """Starts the gather of this rank's tile containers -- tile t at dst[t * pitch : t * pitch + sizes[t]], the layout
    qb3x_encode_tiles leaves -- to `root` and returns at once.  ONE point-to-point message per peer and call: the sender
    packs its containers back to back (device-to-device copies, a fraction of the time the link takes) and sends that one
    span, sized to the bytes produced; the sizes -- exchanged first, in one small all-gather when max_tiles is given --
    are the offset table the root needs to take the span apart.  The transfer (RCCL send/recv on its own stream over
    xGMI, every peer on its own link to the root) runs beside the coding of the next batch.
    recv_bufs: on the root, optional list of per-rank uint8 tensors to receive into (reused step after step); the root's
    own entry is ignored (its tiles stay in dst).  send_buf: on a peer, optional uint8 tensor to pack into."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    keep = []
    if dst.is_cuda and dist.get_backend(group) != "nccl":
        dst = dst.cpu()                 # gloo moves host memory (CPU tests, single-GPU rehearsals); RCCL moves HBM to HBM
        recv_bufs = send_buf = None
    device = dst.device
    size_lists = _exchange_sizes(list(sizes), device, group, max_tiles)
    ops, bufs, offset_lists = [], None, None
    if rank == root:
        bufs, offset_lists = [], []
        for r in range(world):
            if r == root:
                bufs.append(dst)
                offset_lists.append([t * pitch for t in range(len(size_lists[r]))])
                continue
            offs, total = [], 0
            for n in size_lists[r]:
                offs.append(total)
                total += n
            offset_lists.append(offs)
            b = recv_bufs[r] if recv_bufs is not None and recv_bufs[r] is not None and recv_bufs[r].numel() >= total \
                else torch.empty(total, dtype=torch.uint8, device=device)
            bufs.append(b)
            if total:
                ops.append(dist.P2POp(dist.irecv, b[:total], r, group))
    else:
        total = sum(int(n) for n in sizes)
        if total:
            pack = send_buf if send_buf is not None and send_buf.numel() >= total else torch.empty(total, dtype=torch.uint8, device=device)
            off = 0
            for t, n in enumerate(sizes):
                n = int(n)
                if n:
                    pack[off:off + n].copy_(dst[t * pitch:t * pitch + n])
                    off += n
            keep.append(pack)
            ops.append(dist.P2POp(dist.isend, pack[:total], root, group))
    reqs = dist.batch_isend_irecv(ops) if ops else []
    return PendingGather(reqs, bufs, size_lists if rank == root else None, offset_lists, keep)
