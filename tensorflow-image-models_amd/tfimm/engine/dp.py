"""Data-parallel forward over the GPUs of one node: one process per GPU, weights replicated, the
batch split in contiguous shards, and ONE exchange step -- an all-gather of the fp32 logits
(RCCL over xGMI through ``torch.distributed``'s ``nccl`` backend).

Images are independent at inference (BatchNorm uses moving statistics: every norm call in the
reference passes ``training=training``, e.g. resnet.py:270,275,281), so there is no other
collective on the path.  The helpers are backend-agnostic so the sharding / gather logic is
covered by world-size-2 ``gloo`` tests on CPU (tests/test_distributed.py).
"""
from typing import Callable, Tuple


def shard_bounds(batch: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of ``rank``; the first ``batch % world`` ranks get one extra image."""
    q, r = divmod(batch, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def all_gather_rows(local, batch: int, dist=None):
    """Gather per-rank row blocks ``local`` (shard_bounds order) into the full ``(batch, ...)`` tensor on
    every rank.  Equal shards use one ``all_gather_into_tensor`` (a single ring pass over xGMI);
    ragged shards are padded to the largest shard first."""
    import torch
    if dist is None:
        import torch.distributed as dist
    world = dist.get_world_size()
    if world == 1:
        return local
    rank = dist.get_rank()
    sizes = [shard_bounds(batch, world, r)[1] - shard_bounds(batch, world, r)[0] for r in range(world)]
    assert local.shape[0] == sizes[rank], (local.shape, sizes, rank)
    mx = max(sizes)
    if local.shape[0] != mx:
        pad = torch.zeros((mx - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], dim=0)
    dev = local.device
    if local.is_cuda and dist.get_backend() == "gloo":
        local = local.cpu()           # gloo moves bytes over host sockets (ranks may share one GPU); RCCL gathers in HBM
    out = torch.empty((world * mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous())
    out = out.to(dev)
    if all(s == mx for s in sizes):
        return out
    return torch.cat([out[r * mx:r * mx + sizes[r]] for r in range(world)], dim=0)


def dp_forward(forward: Callable, x, dist=None):
    """``forward`` maps a local image batch to local logits (a torch tensor).  ``x`` is the GLOBAL
    batch (every rank holds it, or at least its own shard's rows are valid): each rank runs its
    shard and all ranks return the full ``(B, nb_classes)`` logits."""
    if dist is None:
        import torch.distributed as dist
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        return forward(x)
    lo, hi = shard_bounds(x.shape[0], world, dist.get_rank())
    return all_gather_rows(forward(x[lo:hi]), x.shape[0], dist)


class PipelinedGather:
    """The logits exchange of step i running UNDER step i + 1 (the all-gather is the only collective).

    Issued the plain way -- ``all_gather_into_tensor`` on the launch stream after every forward -- the exchange is on the
    critical path: the next forward's first kernel waits for a 1-MB ring pass that costs a launch, a rendezvous and a drain
    (measured at world = 1: +0.2 .. 0.3 ms per 3.6-ms ResNet-50 step, profiles/r04_bench_rccl_world1.json).  Here the
    collective is asynchronous (``async_op=True``: torch enqueues it on the communicator's own stream behind an event of the
    launch stream) and double-buffered: ``submit(logits)`` copies the rows into send slot ``i % depth`` on the launch stream
    and starts the gather into receive slot ``i % depth``; the launch stream only waits for a slot's PREVIOUS gather when
    the slot comes round again, ``depth`` steps later, when it has long finished.  ``result(slot)`` / ``drain()`` make the
    launch stream wait for what is still in flight.  Ragged shards are not handled here (``all_gather_rows`` pads).
    With the ``gloo`` backend and device tensors the rows go through host memory synchronously (ranks sharing a GPU).
    """

    def __init__(self, rows: int, cols: int, dtype, device, dist=None, depth: int = 2):
        import torch
        if dist is None:
            import torch.distributed as dist
        self.dist, self.depth = dist, depth
        self.world = dist.get_world_size()
        self.host = torch.device(device).type == "cuda" and dist.get_backend() == "gloo"
        self.send = [torch.empty(rows, cols, dtype=dtype, device=device) for _ in range(depth)]
        self.recv = [torch.empty(self.world * rows, cols, dtype=dtype, device=device) for _ in range(depth)]
        self.work = [None] * depth
        self.step = 0

    def submit(self, local) -> int:
        """Start the exchange of ``local`` (this rank's rows); returns the slot that will hold the gathered rows."""
        import torch
        k = self.step % self.depth
        self.step += 1
        if self.work[k] is not None:          # the slot's previous exchange (depth steps ago): normally long complete
            self.work[k].wait()
            self.work[k] = None
        self.send[k].copy_(local, non_blocking=True)
        if self.host:
            lh = self.send[k].cpu()
            gh = torch.empty(self.recv[k].shape, dtype=lh.dtype)
            self.dist.all_gather_into_tensor(gh, lh)
            self.recv[k].copy_(gh)
        else:
            self.work[k] = self.dist.all_gather_into_tensor(self.recv[k], self.send[k], async_op=True)
        return k

    def result(self, slot: int):
        """The gathered rows of ``slot`` (the current stream waits for its exchange)."""
        if self.work[slot] is not None:
            self.work[slot].wait()
            self.work[slot] = None
        return self.recv[slot]

    def last(self):
        """Gathered rows of the most recent ``submit``."""
        assert self.step > 0, "nothing submitted"
        return self.result((self.step - 1) % self.depth)

    def drain(self):
        for k in range(self.depth):
            if self.work[k] is not None:
                self.work[k].wait()
                self.work[k] = None
