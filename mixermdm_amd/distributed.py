"""Batch sharding of the sampling loop across the GPUs of one node (one process per GPU, torch.distributed).

Every motion is an independent DDIM trajectory: the loop itself needs NO collective.  The only
exchanges are, once per job, a broadcast of the packed weights (+ conditioning / x_T when rank 0 owns the request)
and a gather of the finished motions.  Backend "nccl" is RCCL over xGMI on ROCm; "gloo" is used by the CPU tests.
"""
import torch
import torch.distributed as dist


def shard_range(total, world, rank):
    """Contiguous shard [lo, hi) of `total` motions for `rank`; earlier ranks take the remainder (sizes differ by <= 1)."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def broadcast_state_dict(sd, shapes, src=0, device=None):
    """One collective for all weights: rank `src` packs its state dict into a single fp32 vector, every rank receives it
    and unpacks views.  `shapes` (name -> shape, same order on all ranks) comes from mixermdm_amd.synthetic.mixer_shapes
    or a checkpoint's metadata.  Returns name -> tensor views of the packed vector on `device`."""
    total = sum(int(torch.Size(s).numel()) for s in shapes.values())
    flat = torch.empty(total, dtype=torch.float32, device=device)
    if not dist.is_initialized() or dist.get_rank() == src:
        flat.copy_(torch.cat([sd[k].reshape(-1).to(torch.float32) for k in shapes]))
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(flat, src)
    out, off = {}, 0
    for k, shp in shapes.items():
        n = int(torch.Size(shp).numel())
        out[k] = flat[off:off + n].view(shp)
        off += n
    return out


def scatter_requests(cond, x_T, src=0, device=None):
    """Rank `src` holds the whole request (cond [B, C], x_T [B, T, F]); every rank gets its shard.  Implemented as one
    broadcast of each tensor + a local slice (B*T*F*4 bytes is ~10 MB at B=16: negligible next to a 1000-step loop)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    meta = torch.zeros(4, dtype=torch.int64, device=device)
    if rank == src:
        meta = torch.tensor([cond.shape[0], cond.shape[1], x_T.shape[1], x_T.shape[2]], dtype=torch.int64, device=device)
    dist.broadcast(meta, src)
    B, Cc, T, Fd = [int(v) for v in meta.tolist()]
    c = cond.to(device) if rank == src else torch.empty(B, Cc, device=device)
    x = x_T.to(device) if rank == src else torch.empty(B, T, Fd, device=device)
    dist.broadcast(c, src)
    dist.broadcast(x, src)
    lo, hi = shard_range(B, world, rank)
    return c[lo:hi].contiguous(), x[lo:hi].contiguous(), (lo, hi, B)


def gather_motions(local, total, dst=None):
    """The [total, T, F] result, rows in shard order (uneven shards are padded for the collective).
    dst=None: every rank receives it (one all-gather).  dst=r: only rank r does (one gather: what a caller that owns the request needs --
    at BASELINE configs[3] an all-gather would deliver 161 MB to each of 8 ranks of which 7 throw it away); the other ranks return None."""
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [shard_range(total, world, r) for r in range(world)]
    mx = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros(mx, *local.shape[1:], dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    if dst is None:
        bufs = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(bufs, pad)
    else:
        bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
        dist.gather(pad, bufs, dst=dst)
        if rank != dst:
            return None
    return torch.cat([b[:hi - lo] for b, (lo, hi) in zip(bufs, sizes)], dim=0)
