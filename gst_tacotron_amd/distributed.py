"""Utterance-sharded multi-GPU inference: one process per GPU, one gather at the end.

The reference is single-device (no tf.distribute / NCCL anywhere); the path shards
naturally because utterances are independent (BatchNorm is in inference mode, no cross-utterance
op).  Each rank decodes a contiguous block of utterances with a full weight replica and the ONLY
exchange is the final gather of the output tensors to rank 0 -- ``torch.distributed`` backend
"nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.
"""
import os

import torch
import torch.distributed as dist


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_process_group(backend=None, device_index=None, force=False):
    """Idempotent init from the torchrun environment (MASTER_ADDR / MASTER_PORT / RANK / WORLD_SIZE).  With the "nccl"
    (= RCCL) backend the process group is bound to this rank's GPU (``device_id``), so the communicator is created eagerly
    on the right device instead of lazily on whichever device is current at the first collective.  A world of one needs no
    process group and gets none unless ``force`` (the single-GPU RCCL test: communicator, gather and barrier on hardware)."""
    rank, local_rank, world = env_rank_world()
    if (world > 1 or force) and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.device_count() > 0 else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", local_rank if device_index is None else device_index)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def spawn_local_ranks(n_ranks, argv, extra_env=None):
    """Start ``n_ranks`` fresh rank processes of ``argv`` on this node (one per GPU) with the torchrun environment
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT) and wait for them; returns the first non-zero
    exit status (0 if all succeed).  The CALLER must not have initialised the GPU: children are separate processes created
    by fork+exec from a parent that only waits (a process that has touched HIP must never exec another program)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n_ranks), "LOCAL_WORLD_SIZE": str(n_ranks),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL over xGMI needs it on this driver
        if extra_env:
            env.update(extra_env)
        procs.append(subprocess.Popen(list(argv), env=env))
    status = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                try:
                    rc = p.wait(timeout=0.5)
                except subprocess.TimeoutExpired:
                    continue
                pending.remove(p)
                if rc != 0 and status == 0:
                    status = rc
                    for q in pending:           # one rank failed: the others would wait in a collective forever
                        q.terminate()
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return status


def shard_bounds(n_items, rank, world):
    """Contiguous block [lo, hi) of ``n_items`` utterances owned by ``rank`` (sizes differ by <= 1)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_inputs(inputs, rank, world):
    """Slice every per-utterance array of an inference pattern dict along axis 0."""
    n = len(inputs["tokens"])
    lo, hi = shard_bounds(n, rank, world)
    return {k: (v[lo:hi] if v is not None else None) for k, v in inputs.items()}


class PendingGather:
    """Handle of a gather started with ``gather_to_root(..., async_op=True)``: the collective runs on the backend's own
    stream while the caller enqueues the next batch; ``result()`` makes the current stream wait for it and returns what the
    synchronous call returns (the concatenated tensor on ``dst``, None elsewhere)."""

    def __init__(self, work, bufs, sizes, send, local):
        self._work, self._bufs, self._sizes, self._send, self._local = work, bufs, sizes, send, local

    def result(self):
        if self._work is None:
            return self._local
        self._work.wait()
        if self._bufs is None:
            return None
        return torch.cat([b[: hi - lo] for b, (lo, hi) in zip(self._bufs, self._sizes)], dim=0)


def gather_to_root(local, n_total=None, dst=0, async_op=False, force=False):
    """Gather per-rank ``[B_local, ...]`` tensors to ``dst``; returns the concatenated tensor on ``dst``
    and None elsewhere.  Ragged shards (batch not divisible by world size) are padded to the largest
    shard for the collective and trimmed after it.  ``async_op=True`` returns a ``PendingGather`` instead, so that the
    next batch's kernels are enqueued behind this batch's compute, not behind its gather.  A world of one returns ``local``
    without a collective unless ``force``."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return PendingGather(None, None, None, None, local) if async_op else local
    world, rank = dist.get_world_size(), dist.get_rank()
    if n_total is None:
        cnt = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
        dist.all_reduce(cnt)
        n_total = int(cnt.item())
    sizes = [shard_bounds(n_total, r, world) for r in range(world)]
    bmax = max(hi - lo for lo, hi in sizes)
    send = local
    if local.shape[0] < bmax:
        pad = torch.zeros((bmax - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        send = torch.cat([local, pad], dim=0)
    send = send.contiguous()
    bufs = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
    if async_op:
        return PendingGather(dist.gather(send, bufs, dst=dst, async_op=True), bufs, sizes, send, local)
    dist.gather(send, bufs, dst=dst)
    if rank != dst:
        return None
    return torch.cat([b[: hi - lo] for b, (lo, hi) in zip(bufs, sizes)], dim=0)


def run_steps(n_steps, one_step, overlap_gather=False, first=0):
    """The multi-GPU step loop of ``bench.py``: ``one_step(i)`` enqueues batch ``i`` on this rank's stream and returns the
    ``PendingGather`` of its output; returns the last batch's gathered result.

    ``overlap_gather=False`` (the default for a world of more than one rank): batch ``i``'s gather is CLAIMED -- the compute stream
    is made to wait for it -- before batch ``i + 1`` is enqueued.  The decode loop of a batch is one persistent launch of 256
    workgroups that must all be resident (one per CU, 157 KB of LDS each); a collective's receive kernel still resident on rank 0
    when that launch starts -- waiting for a late peer -- would keep one workgroup out until the bounded hand-off waits give up
    (caught and repeated on the launch forms, but then every number is the slow path's).  With the claim in front, the launch is
    stream-ordered behind the gather's completion, so the two can never be co-resident: safe by construction, at the cost of the
    gather's ~0.1 ms per 11 ms batch that the overlapped form hides.
    ``overlap_gather=True``: batch ``i``'s gather is claimed after batch ``i + 1`` has been enqueued (the collective runs on the
    backend's stream beside the next batch's compute) -- the faster form where measurement shows the launches do not collide."""
    last = pending = None
    for i in range(n_steps):
        nxt = one_step(first + i)
        if not overlap_gather:
            last = nxt.result()
            continue
        if pending is not None:
            pending.result()
        pending = nxt
    if overlap_gather and pending is not None:
        last = pending.result()
    return last
