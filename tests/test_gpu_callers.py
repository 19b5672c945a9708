"""GPU: the caller-facing loose ends of SURVEY 8b / 8f -- the (B, T, S)-keyed hipGraph cache the per-item evaluation loop needs, history
destinations that are data (not graph arguments), the reference's history shapes for mixing modes 1-2, ClassifierFreeSampleModelX2 as a
stand-alone callable, module forwards that leave the schedule alone, and the request scatter / motion gather on RCCL."""
import os
import socket
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import mixer as MX            # noqa: E402  (checker only)
from oracle import schedule as OS         # noqa: E402
from oracle.layers import pe_table        # noqa: E402
from test_gpu_kernels import assert_close, rnd, dev   # noqa: E402
from test_gpu_sampler import STEP_TOL, golden_sampler  # noqa: E402

DIMS = dict(d_latent=128, d_ff=256, d_layers=2, m_latent=64, m_ff=128, m_layers=2)


def small(max_batch=2, max_frames=300, mode=4, precision="fp32"):
    from mixermdm_amd.sampler import Sampler
    from mixermdm_amd.synthetic import synthetic_state_dict, synthetic_stats
    sd = synthetic_state_dict(seed=7, std=0.05, bias_std=0.02, mixing_mode=mode, **DIMS)
    st = synthetic_stats()
    s = Sampler(d_heads=2, m_heads=2, max_batch=max_batch, max_frames=max_frames, mixing_mode=mode, precision=precision, **DIMS)
    s.load_state_dict(sd)
    s.set_norm_stats(st["mean_hml"], st["std_hml"], st["mean_ih"], st["std_ih"])
    s.prepare()
    W = dict(sd)
    W["sequence_pos_encoder.pe"] = pe_table(64)
    W["denoiser1.sequence_pos_encoder.pe"] = pe_table(128)
    W["denoiser2.sequence_pos_encoder.pe"] = pe_table(128)
    return s, W, (st["mean_hml"], st["std_hml"], st["mean_ih"], st["std_ih"])


def test_graph_cache_serves_alternating_lengths_without_recapture():
    """The evaluation datasets call the sampler per item with per-sample T (src/evaluation/datasets.py:101-122, 438): alternating
    T in {120, 196, 299} and B in {1, 2} must capture each (B, T, S) once and then only replay; results equal the eager path bitwise."""
    from mixermdm_amd.synthetic import synthetic_inputs
    s, _, _ = small()
    s.set_schedule("ddim20")
    keys = [(1, 120), (1, 196), (2, 299), (1, 120), (2, 299), (1, 196), (1, 120)]
    ref = {}
    for B, T in set(keys):
        cond, xT = synthetic_inputs(B, T, seed_cond=100 + T, seed_x=200 + T)
        ref[(B, T)] = s.sample(cond, xT, use_graph=False)
    assert s.graph_stats() == (0, 0, 0)
    for rounds in range(2):
        for B, T in keys:
            cond, xT = synthetic_inputs(B, T, seed_cond=100 + T, seed_x=200 + T)
            assert torch.equal(s.sample(cond, xT, use_graph=True), ref[(B, T)]), (B, T)
    cap, rep, cached = s.graph_stats()
    assert cap == 3 and cached == 3 and rep == 2 * len(keys) * 20, (cap, rep, cached)
    # a second schedule is a different key, and going back to the first one replays its graphs
    s.set_schedule("ddim50")
    cond, xT = synthetic_inputs(1, 120, seed_cond=220, seed_x=320)
    s.sample(cond, xT)
    s.set_schedule("ddim20")
    assert torch.equal(s.sample(cond, xT), s.sample(cond, xT, use_graph=False))
    assert s.graph_stats()[0] == 4 and s.graph_stats()[2] == 4
    s.close()


def test_graph_cache_evicts_least_recently_used(monkeypatch):
    from mixermdm_amd.synthetic import synthetic_inputs
    monkeypatch.setenv("MMDM_GRAPH_CACHE", "2")
    s, _, _ = small()
    s.set_schedule("ddim20")
    for T in (16, 24, 16, 32, 16, 24):       # 16 stays hot; 24 is evicted by 32 and captured again
        cond, xT = synthetic_inputs(1, T)
        s.begin(cond, xT)
        s.run(2, use_graph=True)
    cap, rep, cached = s.graph_stats()
    assert (cap, cached) == (4, 2) and rep == 12, (cap, rep, cached)
    s.close()


def test_history_buffers_of_an_earlier_call_are_never_written_again():
    """sample(history=...) followed by sample() on the same handle, same (B, T), no set_schedule in between: the second call replays the
    cached graph and must neither touch the first call's buffers nor keep copying (history destinations live in a device-side descriptor
    that mmdm_begin resets on the stream; they are not baked into graph nodes)."""
    from mixermdm_amd.synthetic import synthetic_inputs
    s, _, _ = small()
    s.set_schedule("ddim20")
    cond, xT = synthetic_inputs(2, 24)
    names = ("influence_i1", "influence_i2", "out1", "out2", "out_influenced")
    out1, hist = s.sample(cond, xT, history=names)
    snap = {k: v.clone() for k, v in hist.items()}
    for v in hist.values():
        v.fill_(-7.0)                         # stand-in for the caching allocator handing the memory to someone else
    cond2, xT2 = synthetic_inputs(2, 24, seed_cond=9, seed_x=10)
    out2 = s.sample(cond2, xT2)               # no history requested
    torch.cuda.synchronize()
    for k, v in hist.items():
        assert bool((v == -7.0).all()), f"{k}: a replayed step wrote into the previous call's history buffer"
    assert s.graph_stats()[0] == 1            # one capture served both calls
    # and a later call with history again gets exactly what the first one got
    out3, hist3 = s.sample(cond, xT, history=names)
    assert torch.equal(out3, out1) and not torch.equal(out2, out1)
    for k in names:
        assert torch.equal(hist3[k], snap[k]), k
    s.close()


@pytest.mark.parametrize("mode", [1, 2])
def test_history_shapes_of_mixing_modes_1_and_2_follow_the_reference(mode):
    """Modes 1 and 2 append the influence BEFORE any per-channel expansion: [2B, T, 1] (mixermdm.py:739-745, 794-796)."""
    from mixermdm_amd.synthetic import synthetic_inputs
    s, W, stats = small(mode=mode)
    B, T = 2, 20
    cond, xT = synthetic_inputs(B, T)
    s.set_schedule("ddim20")
    out, hist = s.sample(cond, xT, history=("influence_i1", "influence_i2", "out_influenced"))
    assert hist["influence_i1"].shape == (20, 2 * B, T, 1) and hist["influence_i2"].shape == (20, 2 * B, T, 1)
    assert hist["out_influenced"].shape == (20, 2 * B, T, 524)
    ohist = {}
    MX.mixer_ddim_step(W, MX.MixerSpec(d_heads=2, m_heads=2, mixing_mode=mode), stats, OS.make_schedule("cosine", 1000, "ddim20"), 3.5, 19,
                       xT, xT, cond, ohist)
    for k in ("influence_i1", "influence_i2"):
        assert tuple(ohist[k][0].shape) == (2 * B, T, 1)
        assert_close(hist[k][0], ohist[k][0].contiguous(), atol=2e-5, rtol=1e-4, what=f"mode {mode} {k}")
    if mode == 1:                             # one value per sequence, repeated over time
        assert bool((hist["influence_i1"][0] == hist["influence_i1"][0][:, :1]).all())
    s.close()


def test_cfg_x2_forward_vs_reference_golden(golden):
    """ClassifierFreeSampleModelX2.forward(x, x2, timesteps, cond, mask) (cfg_sampler.py:38-56) as a callable of its own."""
    s, g, t = golden_sampler(golden)
    out = s.module_forward(4, t("cfg_x"), t("cfg_cond"), 640, x2=t("cfg_x2"))
    assert out.shape == t("cfg:out").shape
    assert_close(out, t("cfg:out"), what="ClassifierFreeSampleModelX2.forward", **STEP_TOL)
    s.close()


def test_module_forward_keeps_the_schedule():
    """A teacher-forced forward borrows slot 0 of the schedule tables and must put it back: the loop after it equals the loop before it."""
    from mixermdm_amd.synthetic import synthetic_inputs
    s, _, _ = small()
    cond, xT = synthetic_inputs(2, 16)
    s.set_schedule("ddim20")
    a = s.sample(cond, xT)
    s.module_forward(2, rnd(1, 4, 16, 524), rnd(2, 4, 8 * 768), 333, x2=rnd(3, 4, 16, 524))
    s.module_forward(1, rnd(1, 4, 16, 524), rnd(2, 4, 3 * 768), 12)
    assert s.schedule is not None
    b = s.sample(cond, xT)
    assert torch.equal(a, b)
    s.close()


def test_facade_reuses_schedule_and_graph_across_forwards(tmp_path, golden):
    """MixerMDM.forward builds a fresh MixerDiffusion per call like the reference; the handle must recognise the same schedule by content
    (no table re-upload, no re-capture) and still follow a change of sampling_strategy."""
    from test_gpu_facade import tiny_model
    m, g, t = tiny_model(tmp_path, golden, strategy="ddim20")
    xT = t("loop:ddim20:x_T").cuda()
    B, T = xT.shape[:2]
    batch = {"cond": t("cfg_cond").cuda(), "x_T": xT, "motion_lens": torch.tensor([[T]] * B)}
    a = m.forward_test(dict(batch))["output"]
    cap0 = m._sampler.graph_stats()[0]
    b = m.forward_test(dict(batch))["output"]
    full = m.forward(dict(batch))            # history on: same graph, different (device-side) destinations
    assert torch.equal(a, b) and torch.equal(a, full["output"])
    assert m._sampler.graph_stats()[0] == cap0 == 1
    assert len(full["influence_i1"]) == 20 and full["influence_i1"][0].shape == (2 * B, T, 262)
    m.sampling_strategy = "ddim50"
    m.forward_test(dict(batch))
    assert m._sampler.graph_stats()[0] == 2
    m.sampling_strategy = "ddim20"
    assert torch.equal(m.forward_test(dict(batch))["output"], a) and m._sampler.graph_stats()[0] == 2
    # the CFG wrapper the facade builds is a callable of its own (cfg_sampler.py:38): reference golden
    out = m.cfg_model(t("cfg_x").cuda(), t("cfg_x2").cuda(), torch.full((B,), 640), cond=t("cfg_cond").cuda(), mask=None)
    assert_close(out, t("cfg:out"), what="facade ClassifierFreeSampleModelX2.forward", **STEP_TOL)


def test_scatter_sample_gather_on_rccl_world_1():
    """scatter_requests -> Sampler -> gather_motions through RCCL (backend "nccl") at world size 1: the collectives run for real and the
    result is the unsharded sample, bit for bit.  (World size 2 is covered on gloo by tests/test_distributed_cpu.py; batch rows are
    independent bitwise: test_gpu_fullsize.py.)"""
    import torch.distributed as dist
    from mixermdm_amd.distributed import scatter_requests, gather_motions, broadcast_state_dict
    from mixermdm_amd.synthetic import synthetic_inputs, synthetic_state_dict, mixer_shapes
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    d = dev()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=d)
    try:
        sd = synthetic_state_dict(seed=7, std=0.05, bias_std=0.02, **DIMS)
        got = broadcast_state_dict(sd, mixer_shapes(**DIMS), src=0, device=d)
        assert all(torch.equal(got[k].cpu(), sd[k]) for k in sd)
        s, _, _ = small(max_batch=3, max_frames=40)
        s.set_schedule("ddim20")
        cond, xT = synthetic_inputs(3, 40)
        ref = s.sample(cond, xT)
        c, x, (lo, hi, total) = scatter_requests(cond, xT, src=0, device=d)
        assert (lo, hi, total) == (0, 3, 3)
        full = gather_motions(s.sample(c, x), total)
        dist.barrier()
        assert torch.equal(full, ref)
        assert torch.equal(gather_motions(s.sample(c, x), total, dst=0), ref)       # rank-0-only gather (dist.gather on RCCL)
        s.close()
    finally:
        dist.destroy_process_group()


def test_facade_level_sharded_sampling_on_rccl_world_1(tmp_path, golden):
    """mixermdm_amd.distributed.sample_sharded around the real facade on RCCL (world size 1: broadcast, all_gather / gather execute on the GPU):
    forward (all five history lists) and forward_test equal the unsharded calls bit for bit, in the reference's [2B, T, C] history layout.
    World size 2 (ragged / empty shards) runs on gloo with a stub model: tests/test_distributed_cpu.py."""
    import torch.distributed as dist
    from mixermdm_amd.distributed import sample_sharded
    from test_gpu_facade import tiny_model
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    m, g, t = tiny_model(tmp_path, golden, strategy="ddim20")
    xT = t("loop:ddim20:x_T").cuda()
    B, T = xT.shape[:2]
    batch = {"cond": t("cfg_cond").cuda(), "x_T": xT, "motion_lens": torch.tensor([[T]] * B)}
    ref = m.forward(dict(batch))
    ref_t = m.forward_test(dict(batch))
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev())
    try:
        got = sample_sharded(m, dict(batch), fn="forward", owner=0, histories=True)
        assert torch.equal(got["output"], ref["output"])
        for nm in ("influence_i1", "influence_i2", "out1", "out2", "out_influenced"):
            assert len(got[nm]) == len(ref[nm]) == 20 and all(torch.equal(a, b) for a, b in zip(got[nm], ref[nm])), nm
        got_t = sample_sharded(m, dict(batch), fn="forward_test", owner=0, gather_to=0)
        assert torch.equal(got_t["output"], ref_t["output"]) and got_t["influence_i1"] == []
        d = np.abs(got["output"].cpu().numpy() - g["loop:ddim20:output"])
        assert d.mean() <= 2e-3                                   # and it is still the reference's loop
    finally:
        dist.destroy_process_group()


def test_bench_runs_its_rccl_path_under_torchrun_on_the_gpu_box():
    """`bench.py` as the driver launches it for N > 1 -- `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` -- with
    N = 1 and MMDM_BENCH_FORCE_DIST=1, as a CHILD process: the RCCL init -> weight broadcast -> barrier -> all_gather / all_reduce of the
    timings really execute on this box (the CPU suite only covers the launcher with --dry-run on gloo), at the per-GPU batch of BASELINE
    configs[3] (32 motions), and the JSON line names that workload."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(MMDM_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(root, "bench.py"), "--gpus", "1", "--batch", "32", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-alt", "--no-side", "--no-full-loop",
           "--profile-steps", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["config"]["batch_per_gpu"] == 32 and line["scaling"] == "weak" and line["outputs_finite"]
    assert "2-person MixerMDM" in line["config"]["workload"] and "batch 32 per GPU" in line["config"]["workload"]
    assert line["ms_per_step"] > 0 and abs(line["value"] - 32 / (line["ms_per_step"] * 1e-3 * 1000)) < 1e-3
    assert line["ms_per_step_ranks"]["min"] == line["ms_per_step_ranks"]["max"]          # the all_gather of the per-rank timings ran (one rank)
    # what the first real `bench.py --gpus 8` line will be checked against: the per-GPU shard is configs[3]'s 32 motions, and no HBM-traffic
    # figure is claimed for a batch no PMC pass was taken at (profiles/gemm_traffic.json is the B = 16 headline's)
    assert line["per_gpu_batch"] == 32 and line["n_gpus"] == 1 and line["config"]["parallelism"].startswith("batch-sharded x1")
    assert line["roofline"] is None or line["roofline"]["traffic"] is None
    # every line says which library it ran: the in-tree build, no MMDM_LIB override
    assert line["mmdm_version"].startswith("gfx950;") and line["lib"] == os.path.join("mixermdm_amd", "libmmdm_hip.so") and line["lib_override"] is False
