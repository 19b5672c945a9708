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


def _gather_history(lists, lo, hi, total, dst, device=None):
    """History side outputs of a shard -- a list (one entry per kept step) of [2b, T, C] tensors whose rows are the shard's cond rows followed
    by its uncond rows -- gathered into the global [2B, T, C] layout: all cond rows in shard order, then all uncond rows."""
    b = hi - lo if lists else 0          # a rank with an empty shard (or a model that kept nothing) still takes part in every collective below
    st = torch.stack(lists, dim=1) if b else None                               # [2b, S, T, C]
    world = dist.get_world_size()
    # the shape of an entry must be known on ranks without rows too: taken from the first rank that owns some
    meta = torch.tensor([len(lists), st.shape[2], st.shape[3], 1] if b else [0, 0, 0, 0], dtype=torch.int64, device=st.device if b else device)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta)
    have = [m for m in metas if int(m[3]) == 1]
    if not have:
        return []
    S, T, Cc, _ = [int(v) for v in have[0].tolist()]
    halves = []
    for h in range(2):
        part = st[h * b:(h + 1) * b].reshape(b, S * T, Cc) if b else torch.zeros(0, S * T, Cc, device=device)
        g = gather_motions(part, total, dst=dst)
        halves.append(None if g is None else g.reshape(total, S, T, Cc))
    if halves[0] is None:
        return None
    full = torch.cat(halves, dim=0)                                             # [2B, S, T, C]
    return list(full.unbind(1))


def sample_sharded(model, batch, fn="forward_test", owner=0, gather_to=None, histories=False):
    """One sampling request spread over the ranks of the initialised process group: ``model.forward_test(batch)`` (or ``forward``) with the
    B motions of the request split into contiguous shards (shard_range), one per GPU, and no collective inside the denoising loop.

    Rank `owner` holds the request (the other ranks may pass batch=None): it encodes the text (or takes batch["cond"]), draws x_T when the
    batch has none (the reference draws it inside the sampler, gaussian_diffusion.py:1798-1802: every rank must see the SAME noise, so it is
    drawn once), and both are broadcast and sliced (scatter_requests).  Every rank runs the reference-API call on its shard; the finished
    motions are gathered in shard order to every rank (gather_to=None) or to rank `gather_to` only (the others get None).  histories=True
    also gathers the influence / out1 / out2 / out_influenced lists into the reference's [2B, T, C] row layout.  Without a process group
    this is the plain call (with a group of one rank the collectives still run: the RCCL path can be exercised on one GPU).  Results are
    bitwise those of the unsharded call: no kernel mixes batch rows (tests)."""
    if not dist.is_initialized():
        return getattr(model, fn)(batch)
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = model.device
    cond = x_T = None
    if rank == owner:
        cond = model.generate_cond(batch)
        T = int(batch["motion_lens"][0])
        x_T = batch["x_T"] if batch.get("x_T") is not None else torch.randn(cond.shape[0], T, 2 * model.nfeats, device=dev)
    cond_s, x_s, (lo, hi, B) = scatter_requests(cond, x_T, src=owner, device=dev)
    T = x_s.shape[1]
    out = None
    if hi > lo:
        out = getattr(model, fn)({"cond": cond_s, "x_T": x_s, "motion_lens": torch.full((hi - lo, 1), T, dtype=torch.long)})
    local = out["output"] if out is not None else torch.zeros(0, T, x_s.shape[2], device=dev)
    res = {"output": gather_motions(local, B, dst=gather_to)}
    names = ("influence_i1", "influence_i2") if fn == "forward_test" else ("influence_i1", "influence_i2", "out1", "out2", "out_influenced")
    for nm in names:
        res[nm] = _gather_history(out[nm] if out is not None else [], lo, hi, B, gather_to, dev) if histories else []
    if gather_to is not None and rank != gather_to:
        return None
    return res


def shard_items(n_items, world=None, rank=None):
    """Indices of the evaluation items rank `rank` generates: round-robin, so that long and short motions spread evenly."""
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    return list(range(rank, n_items, world))


def gather_items(local, n_items, device=None):
    """Every rank's {item index: result} dictionaries merged into one list in item order on every rank (host objects: the evaluation
    harness keeps numpy arrays and strings, src/evaluation/datasets.py:117-163).  Under the RCCL backend the pickled objects are staged
    through the CURRENT device: `device` (the model's) is made current first, so every rank stages on its own GPU whatever the caller's
    current device was."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [local[i] for i in range(n_items)]
    if device is not None and torch.device(device).type == "cuda" and dist.get_backend() == "nccl":
        torch.cuda.set_device(torch.device(device))
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, local)
    merged = {}
    for p in parts:
        merged.update(p)
    return [merged[i] for i in range(n_items)]
