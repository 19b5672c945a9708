"""CPU, world_size 2, gloo: the N>1 host path (weight broadcast, request sharding, result gather).  The loop itself has no
collective; the per-rank compute is replaced here by a deterministic per-row function so the gathered result can be
checked against the unsharded one (the GPU counterpart is tests/test_gpu_sampler.py::test_batch_rows_are_independent)."""
import os
import socket
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mixermdm_amd.distributed import shard_range, broadcast_state_dict, scatter_requests, gather_motions
from mixermdm_amd.synthetic import mixer_shapes, synthetic_state_dict

DIMS = dict(d_latent=16, d_ff=32, d_layers=1, m_latent=16, m_ff=32, m_layers=1)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _row_fn(cond, x):          # stands in for an independent per-motion trajectory
    return x * 2.0 + cond[:, :1, None] - 1.0


def _worker(rank, world, port, batches, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shapes = mixer_shapes(**DIMS)
    sd = synthetic_state_dict(seed=0, **DIMS) if rank == 0 else None
    got = broadcast_state_dict(sd, shapes, src=0)
    ref = synthetic_state_dict(seed=0, **DIMS)
    assert set(got) == set(ref) and all(torch.equal(got[k], ref[k]) for k in ref)
    for B in batches:
        g = torch.Generator().manual_seed(5)
        cond, xT = (torch.randn(B, 8 * 4, generator=g), torch.randn(B, 6, 524, generator=g)) if rank == 0 else (None, None)
        c, x, (lo, hi, total) = scatter_requests(cond, xT, src=0)
        assert total == B and (lo, hi) == shard_range(B, world, rank) and c.shape[0] == hi - lo
        full = gather_motions(_row_fn(c, x), total)
        torch.save(full, os.path.join(out_dir, f"B{B}_r{rank}.pt"))
        only1 = gather_motions(_row_fn(c, x), total, dst=1)          # rank-1-only gather: the owner gets the same rows, the others nothing
        assert (only1 is None) == (rank != 1) and (only1 is None or torch.equal(only1, full))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_covers_everything():
    for total in [0, 1, 5, 16, 17, 256]:
        for world in [1, 2, 3, 8]:
            rs = [shard_range(total, world, r) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(rs, rs[1:]))
            assert max(h - l for l, h in rs) - min(h - l for l, h in rs) <= 1


def test_two_rank_broadcast_shard_gather(tmp_path):
    batches = [4, 5, 1]          # even, ragged, and fewer motions than ranks (one empty shard)
    mp.spawn(_worker, args=(2, _free_port(), batches, str(tmp_path)), nprocs=2, join=True)
    for B in batches:
        g = torch.Generator().manual_seed(5)
        cond, xT = torch.randn(B, 8 * 4, generator=g), torch.randn(B, 6, 524, generator=g)
        ref = _row_fn(cond, xT)
        for r in range(2):
            assert torch.equal(torch.load(os.path.join(tmp_path, f"B{B}_r{r}.pt")), ref)


def test_bench_starts_its_own_ranks_and_rejects_a_wrong_world():
    """`python bench.py --gpus N` without a launcher starts N ranks itself; `--dry-run` stops after the rendezvous, so the
    launcher path runs here without a GPU.  A world that differs from --gpus is an error, not a silently different benchmark."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run"],
                       capture_output=True, text=True, timeout=240, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line == {"dry_run": True, "n_gpus": 2, "max_rank_seen": 1}
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True, text=True, timeout=240, env=env, cwd=root)
    assert r.returncode != 0 and "gpus" in (r.stderr + r.stdout)
