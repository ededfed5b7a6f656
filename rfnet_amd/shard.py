"""Batch sharding of the operator hot path over the GPUs of one node.

Every kernel's outermost loop is the batch index and no sample reads another's data (the
reference: `for (int i=blockIdx.x;i<b;i+=gridDim.x)`, e.g. tf_nndistance_g.cu:7), so the path
shards by contiguous batch ranges with NO data-path collective: rank r of R owns samples
[r*B/R, (r+1)*B/R).  The only exchange is the loss reduction -- an all-gather of per-sample
losses (B x 4 bytes) -- plus an optional scatter of inputs from rank 0.  One process per GPU,
`torch.distributed` backend "nccl" (= RCCL over xGMI on ROCm); "gloo" on CPU for the tests.
The reference itself is single-GPU (vv_recon.py:32) and has no counterpart.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_* (torchrun's env).
    Returns (rank, world, local_rank).  No-op for WORLD_SIZE=1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if torch.cuda.is_available():
        torch.cuda.set_device(local % torch.cuda.device_count())
    force = os.environ.get("RF_FORCE_PG") == "1"  # testing: exercise the RCCL path on one GPU
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":  # bind the communicator to this rank's GPU up front (no device guessing)
            kw["device_id"] = torch.device("cuda", torch.cuda.current_device())
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def shard_bounds(batch, rank, world):
    """Contiguous, balanced [lo, hi) of `batch` samples for `rank` (first B % R ranks get one more)."""
    base, rem = divmod(batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(tensors, rank, world):
    """Slice every (B, ...) tensor to this rank's batch range (views, no copy)."""
    out = []
    for t in tensors:
        lo, hi = shard_bounds(t.shape[0], rank, world)
        out.append(t[lo:hi])
    return out


def scatter_from_rank0(full, rank, world, device=None, group=None):
    """Optional input distribution: rank 0 holds `full` (B, ...); every rank receives its shard.
    Uses dist.scatter on equal shards (B % R == 0), else a broadcast + local slice."""
    if world == 1 and not (dist.is_available() and dist.is_initialized()):
        return full
    meta = [None]
    if rank == 0:
        meta = [(tuple(full.shape), full.dtype)]
    dist.broadcast_object_list(meta, src=0, group=group)
    shape, dtype = meta[0]
    if device is not None:
        dev = torch.device(device)
    elif dist.get_backend(group) == "nccl":
        # RCCL moves device buffers only: every rank receives on its current GPU
        dev = torch.device("cuda", torch.cuda.current_device())
    else:
        dev = full.device if rank == 0 else torch.device("cpu")
    if rank == 0 and full.device != dev:
        # the usual case: rank 0 loaded the data on the host.  The collective needs the source where the
        # receive buffers are (RCCL moves device buffers only; gloo takes CPU tensors)
        full = full.to(dev, non_blocking=True)
    B = shape[0]
    if B % world == 0:
        mine = torch.empty((B // world,) + tuple(shape[1:]), dtype=dtype, device=dev)
        chunks = list(full.contiguous().chunk(world, 0)) if rank == 0 else None
        dist.scatter(mine, chunks, src=0, group=group)
        return mine
    buf = full.contiguous() if rank == 0 else torch.empty(shape, dtype=dtype, device=dev)
    dist.broadcast(buf, src=0, group=group)
    lo, hi = shard_bounds(B, rank, world)
    return buf[lo:hi].clone()


def all_gather_per_sample(local, batch, rank, world, group=None):
    """Loss reduction: gather each rank's per-sample vector (its shard of B) into the full (B,...)
    tensor on every rank.  Messages are ~1 KB: latency-bound, one direct all-gather.

    Autograd: the collective itself carries no history, so the other ranks' pieces are constants;
    THIS rank's piece is `local` itself, history intact.  A loss written on the gathered vector
    therefore back-propagates into this rank's samples only -- exactly this rank's share of the
    global gradient -- and `allreduce_gradients(params, average=False)` sums the shares.  (The usual
    data-parallel recipe, mean of the LOCAL per-sample losses + `allreduce_gradients(average=True)`,
    needs no gather at all; the gather is then only for reporting.)"""
    if world == 1:
        return local
    sizes = [shard_bounds(batch, r, world) for r in range(world)]
    mx = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local.detach()
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    pieces = [bufs[r][: hi - lo] for r, (lo, hi) in enumerate(sizes)]
    pieces[rank] = local
    return torch.cat(pieces, 0)


def allreduce_gradients(params, group=None, average=True, bucket_bytes=32 << 20):
    """Data-parallel gradient reduction for a training step on the sharded batch (SURVEY.md 8(e):
    "a full training step would add a DP gradient all-reduce of the model's few-M parameters";
    the reference is single-GPU, vv_recon.py:32, and has no counterpart).

    Every rank has back-propagated its own batch shard; the `.grad` of every parameter is summed
    over the ranks (and divided by the world size with `average`).  Gradients are packed into
    flat buckets of at most `bucket_bytes` per dtype -- RFNet's 3.8 M fp32 parameters are ONE
    15 MB message: over xGMI's point-to-point links a ring all-reduce is per-link bound
    (~153 GB/s), so few large messages beat many small ones -- reduced in place by RCCL
    (`nccl` backend; `gloo` in the CPU tests) and unpacked.  Parameters whose grad is None on
    this rank (unreached by the loss -- the same set on every rank, the graph being identical)
    are skipped.  Returns the number of collectives issued."""
    if not (dist.is_available() and dist.is_initialized()):
        return 0
    world = dist.get_world_size(group)
    if world == 1:
        return 0
    grads = [p.grad for p in params if p.grad is not None]
    ncoll = 0
    by_type = {}
    for g in grads:
        by_type.setdefault((g.dtype, g.device), []).append(g)
    for (_, _), gs in by_type.items():
        bucket, size = [], 0
        def flush():
            nonlocal bucket, size, ncoll
            if not bucket:
                return
            flat = torch.cat([g.reshape(-1) for g in bucket])
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
            if average:
                flat.div_(world)
            off = 0
            for g in bucket:
                g.copy_(flat[off:off + g.numel()].view_as(g))
                off += g.numel()
            ncoll += 1
            bucket, size = [], 0
        for g in gs:
            nbytes = g.numel() * g.element_size()
            if bucket and size + nbytes > bucket_bytes:
                flush()
            bucket.append(g)
            size += nbytes
        flush()
    return ncoll


def broadcast_parameters(params, src=0, group=None):
    """Make every rank start from rank `src`'s weights (one flat broadcast per dtype)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    by_type = {}
    for p in params:
        by_type.setdefault((p.dtype, p.device), []).append(p)
    for ps in by_type.values():
        flat = torch.cat([p.detach().reshape(-1) for p in ps])
        dist.broadcast(flat, src=src, group=group)
        off = 0
        with torch.no_grad():
            for p in ps:
                p.copy_(flat[off:off + p.numel()].view_as(p))
                off += p.numel()


def sharded_per_sample(op, tensors, batch=None, group=None):
    """Apply `op(*shards) -> (b_local, ...)` per-sample results on this rank's batch shard of
    `tensors` (each (B, ...), identical on every rank) and return the gathered (B, ...) result.

    Example (Chamfer loss as vv_recon.py:381-385 `chamfer_big`):
        op = lambda a, c: chamfer_per_sample(a, c)
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    B = tensors[0].shape[0] if batch is None else batch
    local = op(*shard_batch(tensors, rank, world))
    return all_gather_per_sample(local, B, rank, world, group)


def chamfer_per_sample(xyz1, xyz2):
    """Per-sample Chamfer loss as `chamfer_big` (vv_recon.py:381-385):
    (mean sqrt(dist1) + mean sqrt(dist2)) / 2, one value per batch element."""
    from .tf_ops.CD.tf_nndistance import nn_distance
    d1, _, d2, _ = nn_distance(xyz1, xyz2)
    return (torch.sqrt(d1).mean(1) + torch.sqrt(d2).mean(1)) / 2


def emd_per_sample(xyz1, xyz2):
    """Per-sample EMD as `earth_mover` (vv_recon.py:392-399): match_cost / num_points.  On the PINNED route
    (rf_earth_mover_mode, RF_EMD_SWEPT): sample i's value is bit-identical whatever the shard it lands in, as the
    reference's per-sample kernel loop makes it (tf_approxmatch.cu:13) -- so the gathered loss vector does not depend on
    how the batch was cut over the ranks (SURVEY 8(d) C5 "cross-GPU loss equality")."""
    from .pc_distance.tf_approxmatch import earth_mover_cost
    return earth_mover_cost(xyz1, xyz2, mode="swept") / float(xyz1.shape[1])
