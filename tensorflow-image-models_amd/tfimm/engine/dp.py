"""Data-parallel forward over the GPUs of one node: one process per GPU, weights replicated, the
batch split in contiguous shards, and ONE exchange step -- an all-gather of the fp32 logits
(RCCL over xGMI through ``torch.distributed``'s ``nccl`` backend).

Images are independent at inference (BatchNorm uses moving statistics: every norm call in the
reference passes ``training=training``, e.g. resnet.py:270,275,281), so there is no other
collective on the path.  The helpers are backend-agnostic so the sharding / gather logic is
covered by world-size-2 ``gloo`` tests on CPU (tests/test_distributed.py).
"""
from typing import Callable, Tuple

# TFIMM_DP_EXCHANGE=capi: the bench / dp helpers exchange logits through CapiComm (direct ncclAllGather behind the C ABI)
# instead of torch.distributed's collective; default: torch.distributed (the path the driver's N > 1 runs have always taken)


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
    launch stream wait for what is still in flight.  Ragged shards (``batch`` given and not a multiple of the world size:
    ``shard_bounds``): ``rows`` is the LARGEST shard, a shorter shard's rows are followed by zero rows in the send slot and
    ``result`` returns the ``batch`` valid rows only.
    With the ``gloo`` backend and device tensors the rows go through host memory synchronously (ranks sharing a GPU).
    ``comm`` (a ``CapiComm``): the exchange is the C ABI's ``tfimm_hip_dp_all_gather_logits`` -- a direct ``ncclAllGather``
    on a side stream of this object's own -- instead of ``torch.distributed``.
    """

    def __init__(self, rows: int, cols: int, dtype, device, dist=None, depth: int = 2, batch: int = None, comm=None):
        import torch
        if dist is None:
            import torch.distributed as dist
        self.dist, self.depth, self.comm = dist, depth, comm
        self.world = comm.world if comm is not None else dist.get_world_size()
        rank = comm.rank if comm is not None else dist.get_rank()
        self.host = comm is None and torch.device(device).type == "cuda" and dist.get_backend() == "gloo"
        self.sizes = None
        if batch is not None and batch != rows * self.world:
            self.sizes = [shard_bounds(batch, self.world, r)[1] - shard_bounds(batch, self.world, r)[0] for r in range(self.world)]
            assert rows == max(self.sizes), (rows, self.sizes)
        self.rows, self.mine = rows, (rows if self.sizes is None else self.sizes[rank])
        self.send = [torch.zeros(rows, cols, dtype=dtype, device=device) for _ in range(depth)]     # (zero: the padding rows stay zero)
        self.recv = [torch.empty(self.world * rows, cols, dtype=dtype, device=device) for _ in range(depth)]
        self.work = [None] * depth
        self.step = 0
        if comm is not None:
            assert dtype == torch.float32 and torch.device(device).type == "cuda", "the C exchange moves fp32 device rows"
            self.side = torch.cuda.Stream(device=device)
            self.done = [None] * depth

    def submit(self, local) -> int:
        """Start the exchange of ``local`` (this rank's rows); returns the slot that will hold the gathered rows."""
        import torch
        k = self.step % self.depth
        self.step += 1
        assert local.shape[0] == self.mine, (local.shape, self.mine)
        if self.comm is not None:
            cur = torch.cuda.current_stream()
            if self.done[k] is not None:      # the slot's previous exchange
                cur.wait_event(self.done[k])
            self.send[k][:self.mine].copy_(local, non_blocking=True)
            ready = torch.cuda.Event()
            ready.record(cur)
            self.side.wait_event(ready)       # the ring pass runs on the side stream, under the next step's kernels
            self.comm.all_gather(self.send[k], self.recv[k], self.side)
            self.done[k] = torch.cuda.Event()
            self.done[k].record(self.side)
            return k
        if self.work[k] is not None:          # the slot's previous exchange (depth steps ago): normally long complete
            self.work[k].wait()
            self.work[k] = None
        self.send[k][:self.mine].copy_(local, non_blocking=True)
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
        import torch
        if self.comm is not None:
            if self.done[slot] is not None:
                torch.cuda.current_stream().wait_event(self.done[slot])
        elif self.work[slot] is not None:
            self.work[slot].wait()
            self.work[slot] = None
        if self.sizes is None:
            return self.recv[slot]
        return torch.cat([self.recv[slot][r * self.rows:r * self.rows + n] for r, n in enumerate(self.sizes)], dim=0)

    def last(self):
        """Gathered rows of the most recent ``submit``."""
        assert self.step > 0, "nothing submitted"
        return self.result((self.step - 1) % self.depth)

    def drain(self):
        import torch
        for k in range(self.depth):
            if self.comm is not None:
                if self.done[k] is not None:
                    torch.cuda.current_stream().wait_event(self.done[k])
            elif self.work[k] is not None:
                self.work[k].wait()
                self.work[k] = None


class CapiComm:
    """The exchange step through the C ABI (include/tfimm_hip_dp.h, libtfimm_hip_dp.so): ``tfimm_hip_dp_create`` (one RCCL
    communicator per process / GPU) and ``tfimm_hip_dp_all_gather_logits`` -- a direct ``ncclAllGather`` call site, the one a
    host without Python uses (tools/capi/dp_host.cpp).  The 128-byte RCCL id is made by rank 0 and carried to the other
    ranks by whatever rendezvous the host already has: here one ``torch.distributed`` broadcast (any backend)."""

    def __init__(self, dist=None, device=None):
        import ctypes as C

        import torch
        from . import ffi
        self.lib = ffi.dp_lib()
        world, rank = (dist.get_world_size(), dist.get_rank()) if dist is not None else (1, 0)
        device = torch.cuda.current_device() if device is None else device
        idbuf = (C.c_uint8 * 128)()
        if rank == 0:
            self._check(self.lib.tfimm_hip_dp_unique_id(idbuf, 128), "dp_unique_id")
        if world > 1:
            on = "cuda" if dist.get_backend() == "nccl" else "cpu"
            t = torch.tensor(list(bytes(idbuf)), dtype=torch.uint8, device=on)
            dist.broadcast(t, src=0)
            idbuf = (C.c_uint8 * 128)(*t.cpu().tolist())
        self.h = C.c_void_p()
        self._check(self.lib.tfimm_hip_dp_create(C.byref(self.h), idbuf, 128, world, rank, device), "dp_create")
        self.world, self.rank = world, rank

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed (rc={rc}): {self.lib.tfimm_hip_dp_last_error().decode('utf-8', 'replace')}")

    def all_gather(self, local, gathered, stream=None):
        """gathered[r * rows : (r + 1) * rows] = rank r's ``local`` (fp32 device rows), enqueued on ``stream``."""
        import ctypes as C

        import torch
        assert local.dtype == torch.float32 and local.is_cuda and local.is_contiguous() and gathered.is_contiguous()
        assert gathered.shape[0] == self.world * local.shape[0] and gathered.shape[1:] == local.shape[1:]
        st = (stream or torch.cuda.current_stream()).cuda_stream
        cols = local[0].numel()
        self._check(self.lib.tfimm_hip_dp_all_gather_logits(self.h, local.data_ptr(), gathered.data_ptr(), local.shape[0], cols,
                                                           C.c_void_p(st)), "dp_all_gather_logits")
        return gathered

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.lib.tfimm_hip_dp_destroy(self.h)
            self.h = None
