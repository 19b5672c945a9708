"""GPU: the sampler handle (weights -> module forwards -> DDIM steps -> loop) against the reference-captured golden
vectors (tiny dims, tests/golden/mixer.npz) and against the CPU oracle at the real model dimensions.

Tolerances: one Mixer.forward / one DDIM step: all but 0.05 % of elements within atol 2e-4 + rtol 2e-4 (geometry is
ill-conditioned for a few elements), none beyond 5e-2.  Loops: distributional (mean / 99th percentile).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import mixer as MX            # noqa: E402  (checker only)
from oracle import denoiser as DN         # noqa: E402
from oracle import schedule as OS         # noqa: E402
from oracle.layers import pe_table        # noqa: E402
from test_gpu_kernels import assert_close, rnd, dev   # noqa: E402

# frac: the golden inputs are N(0,1) noise, so a few rot6d vectors are nearly collinear and their Gram-Schmidt /
# quaternion round trip amplifies 1e-7 rounding differences to 1e-3 (the CPU oracle shows the same handful of outliers
# against the reference: tests/test_oracle_golden.py::close_frac); everything else must agree to 2e-4.  Steps compared with the ORACLE
# use compare_step: same tolerance, never smaller than 12 x the fp32 oracle's own distance from a float64 run of the same step in the
# element's (sample, person, channel class) group, plus the float64 yardstick (tests/parity_tol.py).
from parity_tol import STEP_TOL, compare_step, yardstick, oracle_step_pair      # noqa: E402


def check_vs_oracle(st, W, spec, ostats, osch, i, x, x2, cond, what, names=("x", "x2", "pred_xstart2")):
    """HIP state `st` after one step from (x, x2) at respaced index i  vs  the fp32 oracle (element-wise) and the float64 oracle (yardstick)."""
    r32, r64 = oracle_step_pair(W, spec, ostats, osch, 3.5, i, x, x2, cond)
    r32, r64 = {k: r32[k] for k in names}, {k: r64[k] for k in names}
    compare_step(st, r32, r64, what)
    yardstick(st, r32, r64, what)
    return r32


def golden_sampler(golden, mode=4, align=True, force=None, model2_kind=0, max_batch=2, out1=False):
    from mixermdm_amd.sampler import Sampler
    g, w, t = golden("mixer")
    W = w("mix.")
    if mode in (1, 2):
        W.update(w("mix_out1."))
    s = Sampler(d_latent=16, d_ff=32, d_layers=2, d_heads=int(g["d_heads"]), m_latent=16, m_ff=32, m_layers=2, m_heads=int(g["m_heads"]),
                mixing_mode=mode, align=align, force_influence_val=force, model2_kind=model2_kind, cfg_scale=float(g["cfg_scale"]),
                max_batch=max_batch, max_frames=16)
    s.load_state_dict(W)
    s.set_norm_stats(g["mean_hml"], g["std_hml"], g["mean_ih"], g["std_ih"])
    s.prepare()
    return s, g, t


@pytest.mark.parametrize("mode,align,force", [(4, True, None), (4, False, None), (4, True, 0.0), (4, True, 1.0), (1, True, None),
                                              (2, True, None), (3, True, None), (3, False, None)])
def test_mixer_forward_vs_reference_golden(golden, mode, align, force):
    """HIP Mixer.forward == the reference's Mixer.forward output captured in tests/golden/mixer.npz."""
    s, g, t = golden_sampler(golden, mode, align, force)
    out = s.module_forward(2, t("x1"), t("cond"), int(g["t"][0]), x2=t("x2"))
    assert_close(out, t(f"fwd:m{mode}:a{int(align)}:f{force}"), what="Mixer.forward", **STEP_TOL)
    s.close()


def test_mixer_forward_intergen_golden(golden):
    s, g, t = golden_sampler(golden, model2_kind=1)
    out = s.module_forward(2, t("x1"), t("cond"), int(g["t"][0]), x2=t("x2"))
    assert_close(out, t("fwd:intergen"), what="Mixer.forward (InterGen)", **STEP_TOL)
    s.close()


def test_denoisers_vs_reference_golden(golden):
    from mixermdm_amd.sampler import Sampler
    g, w, t = golden("denoisers")
    H = int(g["H"])
    # module_forward shares one t across rows: run each fixture row's timestep separately
    for kind, key, which, xk, ck in [(0, "ind.", 0, "x_ind", "cond_ind"), (0, "int.", 1, "x_int", "cond_int"), (1, "ig.", 1, "x_int", "cond_int")]:
        W = {}
        sd = w(key)
        if which == 0:
            W.update({"denoiser1." + k: v for k, v in sd.items()})
            s = Sampler(d_latent=16, d_ff=32, d_layers=2, d_heads=H, single_only=True, max_batch=1, max_frames=16)
        else:
            from mixermdm_amd.synthetic import synthetic_state_dict
            W = synthetic_state_dict(d_latent=16, d_ff=32, d_layers=2, m_latent=16, m_ff=32, m_layers=2)
            W.update({"denoiser2." + k: v for k, v in sd.items()})
            s = Sampler(d_latent=16, d_ff=32, d_layers=2, d_heads=H, m_latent=16, m_ff=32, m_layers=2, m_heads=2, model2_kind=kind,
                        max_batch=1, max_frames=16)
            s.set_norm_stats(*[np.ones(262, np.float32)] * 4)
        s.load_state_dict(W)
        s.prepare()
        ref = t(key[:-1] + ":out")
        for row, tt in enumerate(g["t"]):
            x = t(xk)[row:row + 1].repeat(2, 1, 1)
            c = t(ck)[row:row + 1].repeat(2, 1)
            out = s.module_forward(which, x, c, int(tt))
            assert_close(out[0], ref[row], atol=2e-5, rtol=1e-4, what=f"{key} row {row}")
        s.close()


def test_ddim_step_and_loop_vs_reference_golden(golden):
    """Teacher-forced DDIM steps (i = 32 and i = 0) and the full ddim50 / ddim20 loops against the reference."""
    s, g, t = golden_sampler(golden)
    B = t("cfg_x").shape[0]
    sch = s.set_schedule("ddim50")
    for i in [32, 0]:
        s.begin(t("cfg_cond"), t("cfg_x"))
        st = s.state()
        st["x2"].copy_(t("cfg_x2").to(dev()))
        # jump to respaced index i: run S-1-i steps is not teacher-forced, so poke the device step word instead
        _set_step(s, i)
        s.run(1, use_graph=False)
        st = s.state()
        for k, nm in [("sample", "x"), ("sample2", "x2"), ("pred_xstart", "pred_xstart"), ("pred_xstart2", "pred_xstart2")]:
            assert_close(st[nm], t(f"ddim:i{i}:{k}"), what=f"i={i} {k}", **STEP_TOL)
    for strat in ["ddim50", "ddim20"]:
        s.set_schedule(strat)
        out, hist = s.sample(t("cfg_cond"), t(f"loop:{strat}:x_T"), use_graph=True,
                             history=("influence_i1", "influence_i2", "out1", "out2", "out_influenced"))
        ref = g[f"loop:{strat}:output"]
        d = np.abs(out.cpu().numpy() - ref)
        # same distributional bound the CPU oracle meets against the reference (tests/test_oracle_golden.py::test_mixer_loop)
        assert d.mean() <= 2e-3 and np.percentile(d, 99) <= 3e-2, (strat, d.mean(), d.max())
        n = int(g[f"loop:{strat}:nsteps"])
        for name in ["influence_i1", "influence_i2", "out1", "out2", "out_influenced"]:
            assert hist[name].shape[0] == n and hist[name].shape[1] == 2 * B
            sums = hist[name].double().abs().sum(dim=(1, 2, 3)).cpu().numpy()
            np.testing.assert_allclose(sums[:-1], g[f"loop:{strat}:{name}:abssum"][:-1], rtol=2e-3)
            if strat == "ddim50":
                assert_close(hist[name][0], t(f"loop:{strat}:{name}:0"), what=name + "[0]", **STEP_TOL)
    s.close()


def _set_step(s, i):
    """Test helper: overwrite the device-side (step_idx, loop_pos) words through a 1-step-per-index schedule walk."""
    # run (S-1-i) no-op-free alternative: re-begin and run until index i would be slow; instead use the C ABI directly
    import ctypes as C
    S = s.schedule.num_timesteps
    # walk: begin() leaves step = S-1; run_step decrements. For teacher forcing keep x/x2 and only move the index:
    x, x2 = s.state()["x"].clone(), s.state()["x2"].clone()
    if S - 1 - i > 0:
        s.run(S - 1 - i, use_graph=False)
        st = s.state()
        st["x"].copy_(x)
        st["x2"].copy_(x2)
        torch.cuda.synchronize()


# ---------------------------------------------------------------------------------------------------
# real model dimensions vs the oracle
# ---------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def full():
    from mixermdm_amd.sampler import Sampler
    from mixermdm_amd.synthetic import synthetic_state_dict, synthetic_stats, FULL_DIMS
    sd = synthetic_state_dict(seed=0, std=0.02, bias_std=0.02, **FULL_DIMS)
    stats = synthetic_stats()
    s = Sampler(d_heads=8, m_heads=8, max_batch=2, max_frames=64, **FULL_DIMS)
    s.load_state_dict(sd)
    s.set_norm_stats(stats["mean_hml"], stats["std_hml"], stats["mean_ih"], stats["std_ih"])
    s.prepare()
    W = dict(sd)
    W["sequence_pos_encoder.pe"] = pe_table(512)
    W["denoiser1.sequence_pos_encoder.pe"] = pe_table(1024)
    W["denoiser2.sequence_pos_encoder.pe"] = pe_table(1024)
    ostats = (stats["mean_hml"], stats["std_hml"], stats["mean_ih"], stats["std_ih"])
    yield s, W, ostats
    s.close()


def test_full_size_denoisers_vs_oracle(full):
    s, W, _ = full
    n, T, t = 2, 48, 777
    x1, x2 = rnd(80, n, T, 262), rnd(81, n, T, 524)
    c1, c3 = rnd(82, n, 768), rnd(83, n, 3 * 768)
    ts = torch.full((n,), t, dtype=torch.long)
    ref = DN.in2in_denoiser(W, "denoiser1.", "individual", x1, ts, c1, 8)
    assert_close(s.module_forward(0, x1, c1, t), ref, atol=1e-4, rtol=1e-3, what="denoiser1 full size")
    ref = DN.in2in_denoiser(W, "denoiser2.", "interaction", x2, ts, c3, 8)
    assert_close(s.module_forward(1, x2, c3, t), ref, atol=1e-4, rtol=1e-3, what="denoiser2 full size")


def test_full_size_step_vs_oracle(full):
    s, W, ostats = full
    from mixermdm_amd.synthetic import synthetic_inputs
    B, T = 2, 40
    cond, xT = synthetic_inputs(B, T)
    spec = MX.MixerSpec(d_heads=8, m_heads=8)
    osch = OS.make_schedule("cosine", 1000, "ddim50")
    s.set_schedule("ddim50")
    s.begin(cond, xT)
    s.run(1, use_graph=False)
    st = s.state()
    r = check_vs_oracle(st, W, spec, ostats, osch, 49, xT, xT, cond, "full-size step B=2 T=40", ("x", "x2", "pred_xstart", "pred_xstart2"))
    rx, rx2 = r["x"], r["x2"]
    # second step from the oracle's state (teacher forced) exercises chains that differ
    st["x"].copy_(rx.to(dev()))
    st["x2"].copy_(rx2.to(dev()))
    torch.cuda.synchronize()
    s.run(1, use_graph=False)
    st = s.state()
    check_vs_oracle(st, W, spec, ostats, osch, 48, rx, rx2, cond, "full-size step 2 B=2 T=40", ("x", "x2"))


def test_graph_replay_equals_eager_and_is_deterministic(full):
    s, W, _ = full
    from mixermdm_amd.synthetic import synthetic_inputs
    cond, xT = synthetic_inputs(2, 32)
    s.set_schedule("ddim20")
    a = s.sample(cond, xT, use_graph=False)
    b = s.sample(cond, xT, use_graph=True)
    c = s.sample(cond, xT, use_graph=True)
    assert torch.equal(a, b) and torch.equal(b, c)
    assert torch.isfinite(a).all()


def test_batch_rows_are_independent(full):
    """Sharding property: sample k of a batch == the same sample run alone (bitwise) -> multi-GPU shards
    reproduce the single-GPU result without any cross-GPU math."""
    s, W, _ = full
    from mixermdm_amd.synthetic import synthetic_inputs
    cond, xT = synthetic_inputs(2, 24)
    s.set_schedule("ddim20")
    both = s.sample(cond, xT)
    for k in range(2):
        one = s.sample(cond[k:k + 1], xT[k:k + 1])
        assert torch.equal(one[0], both[k]), k


def test_history_stride_and_slots(full):
    s, W, _ = full
    from mixermdm_amd.synthetic import synthetic_inputs
    cond, xT = synthetic_inputs(1, 16)
    s.set_schedule("ddim20")
    out, h1 = s.sample(cond, xT, history=("influence_i1", "out_influenced"), history_every=1)
    out2, h4 = s.sample(cond, xT, history=("influence_i1", "out_influenced"), history_every=4)
    assert torch.equal(out, out2)
    assert h1["influence_i1"].shape == (20, 2, 16, 262) and h4["influence_i1"].shape == (5, 2, 16, 262)
    for k in range(5):
        assert torch.equal(h4["influence_i1"][k], h1["influence_i1"][4 * k])
        assert torch.equal(h4["out_influenced"][k], h1["out_influenced"][4 * k])
    w = h1["influence_i1"]
    assert (w >= 0).all() and (w <= 1).all()


def test_single_chain_vs_reference_golden(golden):
    from mixermdm_amd.sampler import Sampler
    g, w, t = golden("single")
    s = Sampler(d_latent=16, d_ff=32, d_layers=2, d_heads=int(g["H"]), single_only=True, cfg_scale=float(g["cfg_scale"]), max_batch=2, max_frames=16)
    s.load_state_dict({"denoiser1." + k: v for k, v in w("ind.").items()})
    s.prepare()
    for strat in ["ddim50", "ddim20"]:
        s.set_schedule(strat)
        out = s.sample(t("cond"), t("x_T"))
        d = np.abs(out.cpu().numpy() - g[f"loop:{strat}:output"])
        assert d.mean() <= 1e-4 and d.max() <= 1e-2, (d.mean(), d.max())
    s.close()


def test_errors_are_loud(golden):
    from mixermdm_amd.sampler import Sampler
    from mixermdm_amd import MMDMError
    from mixermdm_amd.synthetic import synthetic_state_dict
    dims = dict(d_latent=16, d_ff=32, d_layers=1, m_latent=16, m_ff=32, m_layers=1)
    s = Sampler(d_heads=2, m_heads=2, max_batch=1, max_frames=8, **dims)
    sd = synthetic_state_dict(**dims)
    with pytest.raises(MMDMError, match="Unexpected key"):
        s.load_state_dict({"nonsense.weight": torch.zeros(2, 2)})
    with pytest.raises(MMDMError, match="size mismatch"):
        s.load_state_dict({"motion_embed.weight": torch.zeros(16, 100)})
    missing = dict(sd)
    missing.pop("influence.out.weight")
    s.load_state_dict(missing)
    s.set_norm_stats(*[np.ones(262, np.float32)] * 4)
    with pytest.raises(MMDMError, match="Missing key"):
        s.prepare()
    s.load_state_dict(sd)
    s.prepare()
    s.set_schedule("ddim20")
    with pytest.raises(MMDMError, match="exceed"):
        s.begin(torch.zeros(2, 8 * 768), torch.zeros(2, 8, 524))
    s.begin(torch.zeros(1, 8 * 768), torch.zeros(1, 8, 524))
    with pytest.raises(MMDMError, match="left in the schedule"):
        s.run(21)
    with pytest.raises(ValueError, match="Mode not recognized"):
        Sampler(d_heads=2, m_heads=2, max_batch=1, max_frames=8, mixing_mode=9, **dims)
    s.close()


def test_interaction_standalone_4way_cfg_vs_reference_golden(golden):
    """single_only=2: in2IN interaction denoiser alone with ClassifierFreeSampleModelMultiple (cfg_sampler.py:59-98)."""
    from mixermdm_amd.sampler import Sampler
    g, w, t = golden("interaction")
    s = Sampler(d_latent=16, d_ff=32, d_layers=2, d_heads=int(g["H"]), single_only=2, cfg_scale=float(g["s"]),
                cfg_scale_interaction=float(g["s_int"]), cfg_scale_individual=float(g["s_ind"]), max_batch=2, max_frames=16)
    s.load_state_dict({"denoiser2." + k: v for k, v in w("int.").items()})
    s.prepare()
    s.set_schedule("ddim20")
    for graph in (False, True):
        out = s.sample(t("cond"), t("x_T"), use_graph=graph)
        d = np.abs(out.cpu().numpy() - g["loop:ddim20:output"])
        assert out.shape == (2, 12, 524) and d.mean() <= 1e-4 and d.max() <= 1e-2, (d.mean(), d.max())
    s.close()


@pytest.mark.parametrize("B,T", [(1, 1), (1, 2), (2, 17), (1, 299), (5, 33), (3, 20)])
def test_ragged_shapes_match_oracle_one_step(full_small, B, T):
    """Edge shapes (single frame, odd lengths, the infer script's T=299, B=3 -- where the reference's dim-less torch.cross
    (alignment.py:198) is wrong but the intended dim=-1 math is what both oracle and kernels implement)."""
    s, W, ostats = full_small
    from mixermdm_amd.synthetic import synthetic_inputs
    cond, xT = synthetic_inputs(B, T, seed_cond=11, seed_x=12)
    spec = MX.MixerSpec(d_heads=4, m_heads=4)
    osch = OS.make_schedule("cosine", 1000, "ddim20")
    s.set_schedule("ddim20")
    s.begin(cond, xT)
    s.run(1, use_graph=True)
    st = s.state()
    check_vs_oracle(st, W, spec, ostats, osch, 19, xT, xT, cond, f"ragged shape B={B} T={T}")


@pytest.fixture(scope="module")
def full_small():
    """Reduced-depth model at real head dims (dh = 128 / 64) for shape sweeps the oracle can follow quickly."""
    from mixermdm_amd.sampler import Sampler
    from mixermdm_amd.synthetic import synthetic_state_dict, synthetic_stats
    dims = dict(d_latent=512, d_ff=1024, d_layers=2, m_latent=256, m_ff=512, m_layers=2)
    sd = synthetic_state_dict(seed=4, std=0.03, bias_std=0.02, **dims)
    stats = synthetic_stats()
    s = Sampler(d_heads=4, m_heads=4, max_batch=5, max_frames=300, **dims)
    s.load_state_dict(sd)
    s.set_norm_stats(stats["mean_hml"], stats["std_hml"], stats["mean_ih"], stats["std_ih"])
    s.prepare()
    W = dict(sd)
    W["sequence_pos_encoder.pe"] = pe_table(256)
    W["denoiser1.sequence_pos_encoder.pe"] = pe_table(512)
    W["denoiser2.sequence_pos_encoder.pe"] = pe_table(512)
    yield s, W, (stats["mean_hml"], stats["std_hml"], stats["mean_ih"], stats["std_ih"])
    s.close()


def test_bf16_path_tracks_the_fp32_path():
    """precision="bf16" (BASELINE configs[4]-style: bf16 GEMM operands, fp32 accumulation / softmax / geometry / DDIM).
    Tolerance (stated for bf16, 8 mantissa bits): after one step the model output agrees with the fp32 path to 3e-2 relative RMS,
    the DDIM state to 2e-3 relative RMS; a ddim20 run stays finite and within 6e-2 relative RMS."""
    from mixermdm_amd.sampler import Sampler
    from mixermdm_amd.synthetic import synthetic_state_dict, synthetic_stats, synthetic_inputs, FULL_DIMS
    sd = synthetic_state_dict(seed=0, std=0.02, bias_std=0.02, **FULL_DIMS)
    st = synthetic_stats()
    B, T = 2, 48
    cond, xT = synthetic_inputs(B, T)
    res = {}
    for prec in ("fp32", "bf16"):
        s = Sampler(d_heads=8, m_heads=8, max_batch=B, max_frames=T, precision=prec, **FULL_DIMS)
        s.load_state_dict(sd)
        s.set_norm_stats(st["mean_hml"], st["std_hml"], st["mean_ih"], st["std_ih"])
        s.prepare()
        s.set_schedule("ddim20")
        s.begin(cond, xT)
        s.run(1, use_graph=False)
        one = {k: v.clone() for k, v in s.state().items()}
        s.run(19, use_graph=True)
        res[prec] = (one, s.state()["pred_xstart2"].clone())
        s.close()
    rel = lambda a, b: ((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item()
    assert rel(res["bf16"][0]["model_out"], res["fp32"][0]["model_out"]) < 3e-2
    assert rel(res["bf16"][0]["x"], res["fp32"][0]["x"]) < 2e-3 and rel(res["bf16"][0]["x2"], res["fp32"][0]["x2"]) < 2e-3
    assert torch.isfinite(res["bf16"][1]).all() and rel(res["bf16"][1], res["fp32"][1]) < 6e-2
    assert not torch.equal(res["bf16"][0]["x"], res["fp32"][0]["x"])        # the bf16 kernels really ran


def test_bf16_linear_matches_rounded_operand_reference():
    from mixermdm_amd import ops, MMDMError
    import torch.nn.functional as F
    M, N, K = 300, 512, 1024
    x, w, b, r = rnd(90, M, K), rnd(91, N, K, scale=0.03), rnd(92, N), rnd(93, M, N)
    xb, wb = ops.to_bf16(x.to(dev())), ops.to_bf16(w.to(dev()))
    assert torch.equal(xb.cpu(), x.bfloat16())                               # RNE conversion == torch's
    ref = F.linear(xb.float().cpu().double(), wb.float().cpu().double(), b.double())
    assert_close(ops.linear_bf16(xb, wb, b.to(dev())), ref.float(), atol=2e-5, rtol=1e-5)
    assert_close(ops.linear_bf16(xb, wb, b.to(dev()), "resid", r.to(dev())), (ref + r.double()).float(), atol=2e-5, rtol=1e-5)
    g = ops.linear_bf16(xb, wb, b.to(dev()), "gelu", out_dtype=torch.bfloat16)
    assert g.dtype == torch.bfloat16
    assert_close(g.float(), F.gelu(ref).float(), atol=1e-2, rtol=1e-2)
    with pytest.raises(MMDMError, match="K %% 32|K % 32"):
        ops.linear_bf16(xb[:, :48].contiguous(), wb[:, :48].contiguous())


def test_split_linear_is_as_accurate_as_the_fp32_mfma():
    """fp32-split GEMM (three fp16 MFMAs on two-way fp16 splits) against a float64 product: error within 1.5x of the native fp32 MFMA
    kernel's, orders of magnitude below bf16's; the split represents x to 2^-22; epilogues and split output included."""
    from mixermdm_amd import ops, MMDMError
    import torch.nn.functional as F
    M, N, K = 500, 384, 1024
    x, w, b, r = rnd(190, M, K), rnd(191, N, K, scale=0.03), rnd(192, N), rnd(193, M, N)
    xs, ws = ops.split_f32(x.to(dev())), ops.split_f32(w.to(dev()))
    back = xs[0].double() + xs[1].double() / ops.SPLIT_SCALE
    assert ((back.cpu() - x.double()).abs() <= 2.0 ** -22 * x.double().abs() + 2.0 ** -36).all()      # 11 + 11 significand bits
    assert torch.equal(xs[0].cpu(), x.half()) and torch.equal(xs[1].cpu(), ((x - x.half().float()) * ops.SPLIT_SCALE).half())
    ref = F.linear(x.double(), w.double(), b.double())
    got = ops.linear_split(xs, ws, b.to(dev()))
    nat = ops.linear(x.to(dev()), w.to(dev()), b.to(dev()))
    e_split, e_nat = (got.cpu().double() - ref).abs(), (nat.cpu().double() - ref).abs()
    assert e_split.mean() <= 1.5 * e_nat.mean() and e_split.max() <= 2.0 * e_nat.max(), (e_split.mean(), e_nat.mean(), e_split.max(), e_nat.max())
    bf = ops.linear_bf16(ops.to_bf16(x.to(dev())), ops.to_bf16(w.to(dev())), b.to(dev()))
    assert e_split.mean() * 200 < (bf.cpu().double() - ref).abs().mean()
    assert_close(ops.linear_split(xs, ws, b.to(dev()), "resid", r.to(dev())), (ref + r.double()).float(), atol=1e-5, rtol=1e-5)
    g = ops.linear_split(xs, ws, b.to(dev()), "gelu", split_out=True)
    gs = g[0].float() + g[1].float() / ops.SPLIT_SCALE
    assert_close(gs, F.gelu(ref).float(), atol=1e-5, rtol=1e-5)
    assert torch.equal(g, ops.split_f32(ops.linear_split(xs, ws, b.to(dev()), "gelu")))          # the split output is the split of the fp32 one
    with pytest.raises(MMDMError, match="K %% 32|K % 32"):
        ops.linear_split(ops.split_f32(x[:, :40].contiguous().to(dev())), ops.split_f32(w[:, :40].contiguous().to(dev())))


def test_split_operand_beyond_the_fp16_range_is_loud_not_wrong():
    """The fp32-split operand format covers |x| < 65504 (kernels.h); a value beyond it must surface as inf / NaN in every output it touches, never
    as a finite wrong number, and must leave the other rows alone."""
    from mixermdm_amd import ops
    M, N, K = 64, 128, 64
    x, w = rnd(301, M, K), rnd(302, N, K, scale=0.05)
    x[5, 7] = 1.0e5
    got = ops.linear_split(ops.split_f32(x.to(dev())), ops.split_f32(w.to(dev()))).cpu()
    assert not torch.isfinite(got[5]).any()
    rows = [r for r in range(M) if r != 5]
    ref = (x.double() @ w.double().T).float()
    assert torch.isfinite(got[rows]).all()
    assert_close(got[rows], ref[rows], atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize("B,T", [(2, 17), (1, 299), (3, 64)])
def test_fp32_split_path_matches_oracle_like_the_fp32_path(full_small, B, T):
    """precision="fp32_split" against the CPU oracle at real head sizes, SAME tolerance as the native fp32 path (STEP_TOL), and at
    least as close to the oracle as 2x the native path's own distance."""
    from mixermdm_amd.sampler import Sampler
    from mixermdm_amd.synthetic import synthetic_inputs
    s32, W, ostats = full_small
    dims = dict(d_latent=512, d_ff=1024, d_layers=2, m_latent=256, m_ff=512, m_layers=2)
    cond, xT = synthetic_inputs(B, T, seed_cond=21, seed_x=22)
    spec = MX.MixerSpec(d_heads=4, m_heads=4)
    osch = OS.make_schedule("cosine", 1000, "ddim20")
    s = Sampler(d_heads=4, m_heads=4, max_batch=B, max_frames=T, precision="fp32_split", **dims)
    s.load_state_dict({k: v for k, v in W.items() if not k.endswith("sequence_pos_encoder.pe")})
    s.set_norm_stats(*[t.numpy() for t in ostats])
    s.prepare()
    outs = {}
    for nm, smp in (("split", s), ("native", s32)):
        smp.set_schedule("ddim20")
        smp.begin(cond, xT)
        smp.run(1, use_graph=(nm == "split"))
        outs[nm] = {k: v.clone() for k, v in smp.state().items() if v is not None}
    rx2 = check_vs_oracle(outs["split"], W, spec, ostats, osch, 19, xT, xT, cond, f"fp32_split B={B} T={T}")["x2"]
    med = lambda a, b: (a.cpu() - b).abs().median().item()
    assert med(outs["split"]["x2"], rx2) <= 2 * med(outs["native"]["x2"], rx2) + 1e-7
    s.close()


@pytest.mark.parametrize("prec", ["fp32", "fp32_split", "bf16"])
def test_every_precision_mode_vs_reference_golden_at_latent_32(golden, prec):
    """tests/golden/mixer32.npz (captured from the reference at latent 32 / ff 64, the smallest sizes the bf16-matrix-core GEMMs take):
    fp32 and fp32_split must both meet the fp32 step tolerance against the REFERENCE's Mixer.forward / ddim_sample / loop; bf16 its own."""
    from mixermdm_amd.sampler import Sampler
    g, w, t = golden("mixer32")
    s = Sampler(d_latent=32, d_ff=64, d_layers=2, d_heads=2, m_latent=32, m_ff=64, m_layers=2, m_heads=2, cfg_scale=3.5, max_batch=2, max_frames=16, precision=prec)
    s.load_state_dict(w("mix."))
    s.set_norm_stats(g["mean_hml"], g["std_hml"], g["mean_ih"], g["std_ih"])
    s.prepare()
    tol = STEP_TOL if prec != "bf16" else dict(atol=6e-2, rtol=6e-2, frac=2e-2, hard=1.0)
    out = s.module_forward(2, t("x1"), t("cond"), int(g["t"][0]), x2=t("x2"))
    assert_close(out, t("fwd"), what=f"Mixer.forward [{prec}]", **tol)
    s.set_schedule("ddim20")
    out = s.sample(t("cfg_cond"), t("x_T"), use_graph=True)
    d = np.abs(out.cpu().numpy() - g["loop:ddim20:output"])
    if prec == "bf16":
        # this tiny N(0, 0.1)-weight model amplifies perturbations ~50x over the last DDIM steps (see test_oracle_golden.test_mixer_loop):
        # bf16's 4e-3 per-GEMM error does not survive 20 steps of it, so only the single forward above is compared for bf16
        assert torch.isfinite(out).all()
    else:
        # Distributional, with limits set from this fixture's own conditioning: the CPU oracle is 1.7e-3 (mean) / 2.9e-2 (p99) from the
        # reference on this loop and moves by 6.8e-3 / 1.2e-1 when x_T is perturbed by one ulp (20 DDIM steps of a random-weight model
        # amplify rounding noise ~1e5x).  The tight comparisons are the Mixer.forward above and the per-step tests.
        assert d.mean() <= 2e-2 and np.percentile(d, 99) <= 3e-1, (prec, d.mean(), np.percentile(d, 99), d.max())
    s.close()


def test_long_sequence_beyond_the_reference_default(full_small):
    """T = 700 frames (the reference's default is 300; its pe table allows 5000): one step against the oracle, own handle sized for it."""
    from mixermdm_amd.sampler import Sampler
    from mixermdm_amd.synthetic import synthetic_inputs
    _, W, ostats = full_small
    dims = dict(d_latent=512, d_ff=1024, d_layers=2, m_latent=256, m_ff=512, m_layers=2)
    B, T = 1, 700
    s = Sampler(d_heads=4, m_heads=4, max_batch=B, max_frames=T, **dims)
    s.load_state_dict({k: v for k, v in W.items() if not k.endswith("sequence_pos_encoder.pe")})
    s.set_norm_stats(*[t.numpy() for t in ostats])
    s.prepare()
    s.set_schedule("ddim20")
    cond, xT = synthetic_inputs(B, T, seed_cond=31, seed_x=32)
    s.begin(cond, xT)
    s.run(1, use_graph=True)
    st = s.state()
    check_vs_oracle(st, W, MX.MixerSpec(d_heads=4, m_heads=4), ostats, OS.make_schedule("cosine", 1000, "ddim20"), 19, xT, xT, cond, "T=700 B=1")
    from mixermdm_amd._lib import MMDMError
    with pytest.raises(MMDMError, match="exceed"):
        s.begin(*synthetic_inputs(1, 701))
    s.close()
