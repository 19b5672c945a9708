"""CPU, world_size 2, gloo: the N>1 host path (weight broadcast, request sharding, result gather).  The loop itself has no
collective; the per-rank compute is replaced here by a deterministic per-row function so the gathered result can be
checked against the unsharded one (the GPU counterpart is tests/test_gpu_sampler.py::test_batch_rows_are_independent)."""
import os
import socket
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mixermdm_amd.distributed import shard_range, broadcast_state_dict, scatter_requests, gather_motions, sample_sharded, shard_items
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


class _StubModel:
    """The facade's calling convention (mixermdm_amd.models.MixerMDM: .device, .nfeats, generate_cond, forward / forward_test returning the
    reference's dictionary) with the denoising loop replaced by a deterministic per-row function: what sample_sharded moves around."""
    nfeats = 262
    device = torch.device("cpu")
    steps = 3

    def generate_cond(self, batch):
        return batch["cond"]

    def _hist(self, out, width):
        b = out.shape[0]
        rows = torch.cat([out[..., :width], -out[..., :width]], dim=0)           # cond rows, then uncond rows (quirk 12)
        return [rows + k for k in range(self.steps)]

    def forward_test(self, batch):
        assert batch["cond"].shape[0] == batch["x_T"].shape[0] == batch["motion_lens"].shape[0] > 0
        out = _row_fn(batch["cond"], batch["x_T"])
        return {"output": out, "influence_i1": self._hist(out, 262), "influence_i2": self._hist(out * 3, 262)}

    def forward(self, batch):
        d = self.forward_test(batch)
        out = d["output"]
        d.update(out1=self._hist(out, 524), out2=self._hist(out + 1, 524), out_influenced=self._hist(out + 2, 524))
        return d


def _sharded_worker(rank, world, port, batches, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = _StubModel()
    for B in batches:
        g = torch.Generator().manual_seed(11)
        batch = {"cond": torch.randn(B, 8, generator=g), "x_T": torch.randn(B, 5, 524, generator=g), "motion_lens": torch.full((B, 1), 5)} if rank == 0 else None
        full = sample_sharded(m, batch, fn="forward", owner=0, histories=True)                       # every rank receives everything
        torch.save(full, os.path.join(out_dir, f"S{B}_r{rank}.pt"))
        only = sample_sharded(m, batch, fn="forward_test", owner=0, gather_to=1)                     # only rank 1 receives; no histories asked
        assert (only is None) == (rank != 1)
        if only is not None:
            assert torch.equal(only["output"], full["output"]) and only["influence_i1"] == []
    # the evaluation loop, items dealt round-robin (generate_for_evaluation(shard_items=True))
    from mixermdm_amd.generation import generate_for_evaluation
    items = [{"text": ("t%d" % i,), "text_individual1": ("a%d" % i,), "text_individual2": ("b%d" % i,), "motion_lens": torch.tensor([4 + i % 3]),
              "cond": torch.full((1, 8), float(i)), } for i in range(5)]

    class _Eval(_StubModel):
        def forward_test(self, batch):
            B, T = batch["cond"].shape[0], int(batch["motion_lens"][0])
            x = torch.arange(T * 524, dtype=torch.float32).reshape(1, T, 524).repeat(B, 1, 1)
            return {"output": _row_fn(batch["cond"], x)}
    assert shard_items(5) == list(range(rank, 5, world))
    gen, mm = generate_for_evaluation(_Eval(), items, max_length=8, mm_idxs=(1, 4), mm_num_repeats=2, shard_items=True)
    torch.save((gen, mm), os.path.join(out_dir, f"E_r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_facade_level_sharded_sampling_and_item_sharded_evaluation(tmp_path):
    """sample_sharded / generate_for_evaluation(shard_items=True) on two gloo ranks against the unsharded calls: even, ragged and
    fewer-motions-than-ranks requests; outputs and gathered history lists in the reference's [2B, T, C] row order."""
    import numpy as np
    batches = [4, 3, 1]
    mp.spawn(_sharded_worker, args=(2, _free_port(), batches, str(tmp_path)), nprocs=2, join=True)
    m = _StubModel()
    for B in batches:
        g = torch.Generator().manual_seed(11)
        batch = {"cond": torch.randn(B, 8, generator=g), "x_T": torch.randn(B, 5, 524, generator=g), "motion_lens": torch.full((B, 1), 5)}
        ref = m.forward(batch)
        for r in range(2):
            got = torch.load(os.path.join(tmp_path, f"S{B}_r{r}.pt"))
            assert torch.equal(got["output"], ref["output"])
            for nm in ("influence_i1", "influence_i2", "out1", "out2", "out_influenced"):
                assert len(got[nm]) == len(ref[nm]) and all(torch.equal(a, b) for a, b in zip(got[nm], ref[nm])), (B, r, nm)
    g0, m0 = torch.load(os.path.join(tmp_path, "E_r0.pt"), weights_only=False)
    g1, m1 = torch.load(os.path.join(tmp_path, "E_r1.pt"), weights_only=False)
    assert [d["text"] for d in g0] == ["t%d" % i for i in range(5)] == [d["text"] for d in g1]
    assert [d["text"] for d in m0] == ["t1", "t4"] and m0[0]["mm_motions"].shape[0] == 2
    assert all(np.array_equal(a["motion1"], b["motion1"]) for a, b in zip(g0, g1))
    # and equal to the unsharded loop
    from mixermdm_amd.generation import generate_for_evaluation
    items = [{"text": ("t%d" % i,), "text_individual1": ("a%d" % i,), "text_individual2": ("b%d" % i,), "motion_lens": torch.tensor([4 + i % 3]),
              "cond": torch.full((1, 8), float(i)), } for i in range(5)]

    class _Eval(_StubModel):
        def forward_test(self, batch):
            B, T = batch["cond"].shape[0], int(batch["motion_lens"][0])
            x = torch.arange(T * 524, dtype=torch.float32).reshape(1, T, 524).repeat(B, 1, 1)
            return {"output": _row_fn(batch["cond"], x)}
    gref, mref = generate_for_evaluation(_Eval(), items, max_length=8, mm_idxs=(1, 4), mm_num_repeats=2)
    assert all(np.array_equal(a["motion2"], b["motion2"]) for a, b in zip(g0, gref)) and all(np.array_equal(a["mm_motions"], b["mm_motions"]) for a, b in zip(m0, mref))


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


def test_bench_marks_the_roofline_of_calls_too_small_to_fill_the_machine():
    """VERDICT r4 nit 7: at the reference's B = 1 call a GEMM launch is one tile's K loop long; bench.py says so beside `roofline.frac` instead of
    letting 0.17 read as kernel quality.  The headline batch carries no such note."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.small_launch_note(False, 16, 300) is None and bench.small_launch_note(True, 32, 196) is None
    note = bench.small_launch_note(False, 1, 299)
    assert note and "1196 rows" in note and "not kernel quality" in note
    assert "240 rows" in bench.small_launch_note(True, 1, 120)


def test_bench_stamps_the_library_and_names_the_same_batch_reference_point(monkeypatch):
    """every bench line says which library it ran (version string, path, whether MMDM_LIB overrode the in-tree build), and
    the N > 1 lines (32 motions per GPU) carry the committed one-GPU figure at the SAME per-GPU batch, labelled as not measured in that run."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_for_test2", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    st = bench.lib_stamp()
    assert st["mmdm_version"].startswith("gfx950;") and st["lib"] == os.path.join("mixermdm_amd", "libmmdm_hip.so") and st["lib_override"] is False
    monkeypatch.setenv("MMDM_LIB", os.path.join(root, "mixermdm_amd", "libmmdm_hip.so"))
    assert bench.lib_stamp()["lib_override"] is True
    monkeypatch.delenv("MMDM_LIB")
    ref = bench.n1_same_batch(32, "fp32")
    assert ref and ref["measured_in_this_run"] is False and ref["n_gpus"] == 1 and ref["per_gpu_batch"] == 32 and ref["source"].startswith("profiles/r")
    assert abs(ref["value"] - 32 / (ref["ms_per_step"] * 1e-3 * 1000)) < 1e-3
    assert bench.n1_same_batch(16, "fp32") is None and bench.n1_same_batch(32, "bf16") is None
