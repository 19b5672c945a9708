"""GPU: the reference-API mirror (mixermdm_amd.models.MixerMDM) end to end on the reference's golden loop."""
import numpy as np
import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu


def tiny_model(tmp_path, golden, strategy="ddim50", **kw):
    from mixermdm_amd.configs import CfgNode
    from mixermdm_amd.models import MixerMDM
    g, w, t = golden("mixer")
    sub = dict(NUM_LAYERS=2, NUM_HEADS=int(g["d_heads"]), DROPOUT=0.1, INPUT_DIM=262, LATENT_DIM=16, FF_SIZE=32)
    for name, nm in [("individual.yaml", "in2INind"), ("in2IN.yaml", "in2IN")]:
        yaml.safe_dump(dict(NAME=nm, **sub), open(tmp_path / name, "w"))
    cfg = CfgNode(dict(NAME="MixerMDM", GENERATOR=dict(sub, NUM_HEADS=int(g["m_heads"])), DISCRIMINATOR=dict(sub), ACTIVATION="gelu",
                       DIFFUSION_STEPS=1000, BETA_SCHEDULER="cosine", SAMPLER="uniform", MOTION_REP="global", CFG_WEIGHT=float(g["cfg_scale"]),
                       MIXING_MODE=4, FORCE_INFLUENCE_VAL="None", MODEL1="individual.yaml", MODEL2="in2IN.yaml"))
    m = MixerMDM(cfg, num_frames=16, sampling_strategy=strategy, config_root=str(tmp_path), **kw)
    sd = {"mixing." + k: v for k, v in w("mix.").items()}
    sd["model1.decoder.net_individual.out.linear.bias"] = torch.zeros(262)      # an off-path key of the real checkpoint
    m.load_state_dict(sd, strict=True)
    m.set_norm_stats(g["mean_hml"], g["std_hml"], g["mean_ih"], g["std_ih"])
    return m.to("cuda:0").eval(), g, t


def test_forward_matches_reference_loop_and_history_contract(tmp_path, golden):
    m, g, t = tiny_model(tmp_path, golden)
    B, T = t("loop:ddim50:x_T").shape[:2]
    batch = {"cond": t("cfg_cond").cuda(), "x_T": t("loop:ddim50:x_T").cuda(), "motion_lens": torch.tensor([[T]] * B)}
    out = m(batch)
    assert set(out) == {"output", "influence_i1", "influence_i2", "out1", "out2", "out_influenced"}
    assert out["output"].shape == (B, T, 524) and out["output"].is_cuda
    d = np.abs(out["output"].cpu().numpy() - g["loop:ddim50:output"])
    assert d.mean() <= 2e-3 and np.percentile(d, 99) <= 3e-2, (d.mean(), d.max())
    assert len(out["influence_i1"]) == 50 and out["influence_i1"][0].shape == (2 * B, T, 262)
    assert len(out["out1"]) == 50 and out["out_influenced"][0].shape == (2 * B, T, 524)
    assert m.mixing.mode == "eval"
    np.testing.assert_allclose(out["influence_i1"][0].cpu().numpy(), g["loop:ddim50:influence_i1:0"], atol=2e-4, rtol=2e-4)

    out_t = m.forward_test(batch)
    assert set(out_t) == {"output", "influence_i1", "influence_i2"} and m.mixing.mode == "eval_intermediate"
    assert torch.equal(out_t["output"], out["output"])

    m.sampling_strategy = "ddim20"                       # knob used by the eval scripts
    batch["x_T"] = t("loop:ddim20:x_T").cuda()
    out20 = m.forward_test(batch)
    d = np.abs(out20["output"].cpu().numpy() - g["loop:ddim20:output"])
    assert d.mean() <= 2e-3 and len(out20["influence_i1"]) == 20


def test_knobs_force_influence_and_inner_callables(tmp_path, golden):
    m, g, t = tiny_model(tmp_path, golden)
    x1, x2, cond = t("x1").cuda(), t("x2").cuda(), t("cond").cuda()
    ts = t("t").long().cuda()
    ref = g["fwd:m4:a1:fNone"]
    got = m.mixing(x1, ts, cond=cond, mask=None, x2=x2)
    bad = np.abs(got.cpu().numpy() - ref) > 2e-4 + 2e-4 * np.abs(ref)
    assert bad.mean() <= 5e-4
    m.mixing.force_influence_val = 1.0                   # src/evaluation/datasets.py:301,323 flips this between runs
    got = m.mixing(x1, ts, cond=cond, mask=None, x2=x2)
    ref = g["fwd:m4:a1:f1.0"]
    bad = np.abs(got.cpu().numpy() - ref) > 2e-4 + 2e-4 * np.abs(ref)
    assert bad.mean() <= 5e-4
    with pytest.raises(NotImplementedError):
        m.mixing(x1, torch.tensor([1, 2, 3, 4]).cuda(), cond=cond, mask=None, x2=x2)
    o = m.denoiser1(x1[..., :262], ts, cond=cond[:, 3 * 768:4 * 768], mask=None)
    assert o.shape == (4, x1.shape[1], 262) and torch.isfinite(o).all()
    m.mixing.mixing_mode = 9
    with pytest.raises(ValueError, match="Mixing mode not recognized"):
        m.mixing(x1, ts, cond=cond, mask=None, x2=x2)


def test_history_budget_is_enforced(tmp_path, golden, monkeypatch):
    import mixermdm_amd.models as M
    m, g, t = tiny_model(tmp_path, golden, strategy="ddim1000")
    monkeypatch.setattr(M, "HISTORY_BUDGET_BYTES", 2 << 20)
    batch = {"cond": t("cfg_cond").cuda(), "motion_lens": torch.tensor([[8]])}
    with pytest.raises(MemoryError, match="history_every"):
        m(batch)
    m.history_every = 250
    out = m(batch)
    assert len(out["influence_i1"]) == 4 and torch.isfinite(out["output"]).all()


def test_gaussian_filter_matches_scipy():
    from scipy.ndimage import gaussian_filter1d
    from mixermdm_amd import ops
    x = torch.randn(3, 37, 524, generator=torch.Generator().manual_seed(5))
    for sigma in [1.0, 2.5]:
        ref = gaussian_filter1d(x.numpy(), sigma, axis=1, mode="nearest")
        got = ops.gaussian_filter1d(x.cuda(), sigma).cpu().numpy()
        np.testing.assert_allclose(got, ref, atol=1e-6, rtol=1e-6)
    one = torch.randn(1, 1, 8).cuda()                         # T = 1: every tap clamps to the only frame
    np.testing.assert_allclose(ops.gaussian_filter1d(one).cpu().numpy(), one.cpu().numpy(), atol=1e-6)


def test_generation_harnesses(tmp_path, golden):
    """infer-script and eval-dataset callers: file layout, smoothing, per-item T, padding, mm repeats."""
    from scipy.ndimage import gaussian_filter1d
    from mixermdm_amd.generation import generate_one_sample, generate_for_evaluation
    m, g, t = tiny_model(tmp_path, golden, strategy="ddim20")
    cond = t("cfg_cond")[:1].cuda()
    xT = torch.randn(1, 12, 524, generator=torch.Generator().manual_seed(9)).cuda()
    motion = generate_one_sample(m, {"cond": cond, "x_T": xT}, "s0", str(tmp_path / "out"), window_size=12)
    assert motion.shape == (12, 2, 262)
    raw = m({"cond": cond, "x_T": xT, "motion_lens": torch.tensor([[12]])})["output"][0].reshape(12, 2, 262).cpu().numpy()
    np.testing.assert_allclose(motion, gaussian_filter1d(raw, 1, axis=0, mode="nearest"), atol=1e-6, rtol=1e-6)
    saved = np.load(tmp_path / "out" / "s0_motion.npy")
    assert np.array_equal(saved, motion)
    assert np.load(tmp_path / "out" / "s0_influence1.npy").shape == (20, 2, 12, 262)
    items = [dict(text=("a",), text_individual1=("b",), text_individual2=("c",), motion_lens=torch.tensor([T]), cond=cond) for T in (8, 16, 10)]
    gen, mm = generate_for_evaluation(m, items, max_length=16, mm_idxs=[1], mm_num_repeats=2)
    assert len(gen) == 3 and len(mm) == 1
    assert gen[0]["motion1"].shape == (16, 262) and np.all(gen[0]["motion1"][8:] == 0) and np.any(gen[0]["motion1"][:8] != 0)
    assert mm[0]["mm_motions"].shape == (2, 16, 2, 262) and gen[1]["text_individual2"] == "c"


def test_in2in_standalone_facades(golden):
    """in2IN(cfg, mode).forward_test for both shipped sub-model configs (tiny dims), against the reference goldens."""
    from mixermdm_amd.configs import CfgNode
    from mixermdm_amd.models import in2IN
    base = dict(NUM_LAYERS=2, DROPOUT=0.1, INPUT_DIM=262, LATENT_DIM=16, FF_SIZE=32, DIFFUSION_STEPS=1000, BETA_SCHEDULER="cosine", STRATEGY="ddim20")
    g, w, t = golden("single")
    m = in2IN(CfgNode(dict(base, NAME="in2INind", NUM_HEADS=int(g["H"]), CFG_WEIGHT=float(g["cfg_scale"]))), "individual")
    m.decoder.load_state_dict({"net_individual." + k: v for k, v in w("ind.").items()})
    m = m.to("cuda:0")
    out = m.forward_test({"cond_individual_individual1": t("cond").cuda(), "x_T": t("x_T").cuda(), "motion_lens": torch.tensor([12, 12])})
    d = np.abs(out["output"].cpu().numpy() - g["loop:ddim20:output"])
    assert out["output"].shape == (2, 12, 262) and d.mean() <= 1e-4 and d.max() <= 1e-2
    with pytest.raises(NotImplementedError, match="upstream of the HIP path"):
        m.forward_test({"text": ["walk"], "motion_lens": torch.tensor([12])})

    g, w, t = golden("interaction")
    m = in2IN(CfgNode(dict(base, NAME="in2IN", NUM_HEADS=int(g["H"]), CFG_WEIGHT=float(g["s"]), CFG_WEIGHT_INTERACTION=float(g["s_int"]),
                           CFG_WEIGHT_INDIVIDUAL=float(g["s_ind"]))), "interaction")
    m.decoder.load_state_dict({"net_interaction." + k: v for k, v in w("int.").items()})
    m = m.to("cuda:0")
    c = t("cond").cuda()
    out = m.forward_test({"cond_interaction": c[:, :768], "cond_interaction_individual1": c[:, 768:1536], "cond_interaction_individual2": c[:, 1536:],
                          "x_T": t("x_T").cuda(), "motion_lens": torch.tensor([12, 12])})
    d = np.abs(out["output"].cpu().numpy() - g["loop:ddim20:output"])
    assert out["output"].shape == (2, 12, 524) and d.mean() <= 1e-4 and d.max() <= 1e-2


def test_in2in_dual_facade_vs_reference_golden(golden):
    """in2IN(cfg, "dual"): net_individual + net_interaction composed by ClassifierFreeSampleDualMDM (in2in.py:318-329)."""
    from mixermdm_amd.configs import CfgNode
    from mixermdm_amd.models import in2IN
    g, w, t = golden("dual")
    cfg = CfgNode(dict(NAME="in2IN", NUM_LAYERS=2, NUM_HEADS=int(g["H"]), DROPOUT=0.1, INPUT_DIM=262, LATENT_DIM=16, FF_SIZE=32, DIFFUSION_STEPS=1000,
                       BETA_SCHEDULER="cosine", STRATEGY="ddim20", CFG_WEIGHT_INDIVIDUAL=float(g["s_ind"]), CFG_WEIGHT_INTERACTION=float(g["s_int"]),
                       W_FUNC="exp", W_VALUE=float(g["cfg:exp:value"])))
    m = in2IN(cfg, "dual")
    sd = {"net_individual." + k: v for k, v in w("ind.").items()}
    sd.update({"net_interaction." + k: v for k, v in w("int.").items()})
    m.decoder.load_state_dict(sd)
    m = m.to("cuda:0")
    c = t("cond").cuda()
    sl = lambda i: c[:, 768 * i:768 * (i + 1)]
    out = m.forward_test({"cond_interaction": sl(0), "cond_interaction_individual1": sl(1), "cond_interaction_individual2": sl(2),
                          "cond_individual_individual1": sl(3), "cond_individual_individual2": sl(4), "x_T": t("x_T").cuda(),
                          "motion_lens": torch.tensor([12, 12])})
    d = np.abs(out["output"].cpu().numpy() - g["loop:exp:ddim20:output"])
    assert out["output"].shape == (2, 12, 524) and d.mean() <= 1e-4 and d.max() <= 1e-2, (d.mean(), d.max())
    with pytest.raises(ValueError):
        in2IN(cfg, "trio")


def test_mixermdm_facade_with_mdm_as_model1(tmp_path, golden):
    """MODEL1.NAME == "MDM" (mixermdm.py:32-40): denoiser1 is an MDMDenoiser with its own sizes; cond rows [3*768 | 2*latent | 3*768]."""
    from mixermdm_amd.configs import CfgNode
    from mixermdm_amd.models import MixerMDM
    gm, wm, _ = golden("mixer")
    g, w, t = golden("mdm")
    sub = dict(NUM_LAYERS=2, NUM_HEADS=2, DROPOUT=0.1, INPUT_DIM=262, LATENT_DIM=16, FF_SIZE=32)
    for name, nm in [("mdm.yaml", "MDM"), ("in2IN.yaml", "in2IN")]:
        yaml.safe_dump(dict(NAME=nm, **sub), open(tmp_path / name, "w"))
    cfg = CfgNode(dict(NAME="MixerMDM", GENERATOR=dict(sub), DISCRIMINATOR=dict(sub), ACTIVATION="gelu", DIFFUSION_STEPS=1000, BETA_SCHEDULER="cosine",
                       SAMPLER="uniform", MOTION_REP="global", CFG_WEIGHT=3.5, MIXING_MODE=4, FORCE_INFLUENCE_VAL="None", MODEL1="mdm.yaml", MODEL2="in2IN.yaml"))
    m = MixerMDM(cfg, num_frames=16, sampling_strategy="ddim20", config_root=str(tmp_path))
    sd = {"mixing." + k: v for k, v in wm("mix.").items() if not k.startswith("denoiser1.")}
    sd.update({"mixing." + k: v for k, v in w("mixmdm.").items()})
    m.load_state_dict(sd, strict=True)
    m.set_norm_stats(gm["mean_hml"], gm["std_hml"], gm["mean_ih"], gm["std_ih"])
    m = m.to("cuda:0").eval()
    assert m.mixing.denoiser1.text_dim == 16
    out = m.forward_test({"cond": t("loop_cond").cuda(), "x_T": t("loop_x_T").cuda(), "motion_lens": torch.tensor([[12]] * 2)})
    d = np.abs(out["output"].cpu().numpy() - g["loop:ddim20:output"])
    assert d.mean() <= 2e-3 and np.percentile(d, 99) <= 3e-2, (d.mean(), d.max())


def test_facade_runs_the_text_stage_when_the_checkpoint_carries_it(tmp_path, golden):
    """load_state_dict with token_embedding / clip_transformer / clipTransEncoder keys -> generate_cond runs on the GPU from token ids
    and feeds forward_test (tiny 768-wide heads so the cond layout is the real [B, 8*768])."""
    from oracle import encoder as EN
    from test_gpu_extensions import clip_weights, enc_weights
    from test_gpu_kernels import rnd
    m, g, t = tiny_model(tmp_path, golden, strategy="ddim20")
    V, ctx, D, H = 30, 12, 768, 12
    W = clip_weights(900, V, ctx, D, 1, 64)
    W.pop("text_projection")                # MixerMDM aliases the tower's modules only (mixermdm.py:213-216): no projection in its state dict
    for j, (pfx, ln) in enumerate([("clipTransEncoder.", "clip_ln"), ("model1.clipTransEncoder_individual.", "model1.clip_ln_individual"),
                                   ("model2.clipTransEncoder_interaction.", "model2.clip_ln_interaction")]):
        for i in range(2):
            for k, v in enc_weights(1000 + 50 * j + 10 * i, D, 64, std=0.03).items():
                W[f"{pfx}layers.{i}.{k}"] = v
        W[ln + ".weight"], W[ln + ".bias"] = 1 + rnd(1100 + j, D) * 0.1, rnd(1110 + j, D) * 0.1
    sd = {k: p.detach().cpu() for k, p in m.named_parameters()}
    sd.update(W)
    m.load_state_dict(sd, strict=True)
    gen = torch.Generator().manual_seed(3)
    mk = lambda e: torch.cat([torch.randint(1, V - 1, (2, e), generator=gen), torch.full((2, 1), V - 1), torch.zeros(2, ctx - e - 1, dtype=torch.long)], 1)
    t1, t2, tI = mk(3), mk(6), mk(9)
    batch = {"tokens_text_individual1": t1, "tokens_text_individual2": t2, "tokens_text": tI, "motion_lens": torch.tensor([[8]] * 2)}
    cond = m.generate_cond(batch)
    assert cond.shape == (2, 8 * 768) and cond.is_cuda
    cI = EN.clip_text_tower(W, "", tI, H)
    ref_infl_I = EN.text_head(W, "clipTransEncoder.", "clip_ln", cI, tI, 8)
    np.testing.assert_allclose(cond[:, 5 * 768:6 * 768].cpu().numpy(), ref_infl_I.numpy(), atol=2e-4, rtol=2e-4)
    ref_int = EN.text_head(W, "model2.clipTransEncoder_interaction.", "model2.clip_ln_interaction", cI, tI, 8)
    np.testing.assert_allclose(cond[:, :768].cpu().numpy(), ref_int.numpy(), atol=2e-4, rtol=2e-4)
    out = m.forward_test(batch)
    assert out["output"].shape == (2, 8, 524) and torch.isfinite(out["output"]).all()


def test_in2in_facade_text_stage_from_checkpoint_keys(golden):
    """in2IN.load_state_dict with tower + clipTransEncoder_individual keys -> text_process on the GPU from token ids, checked against the
    reference-captured text fixture, then a full forward_test from text tokens."""
    from mixermdm_amd.configs import CfgNode
    from mixermdm_amd.models import in2IN
    g, w, t = golden("single")
    gt, wt, tt = golden("text")
    W = wt("txt.")
    m = in2IN(CfgNode(dict(NUM_LAYERS=2, DROPOUT=0.1, INPUT_DIM=262, LATENT_DIM=16, FF_SIZE=32, DIFFUSION_STEPS=1000, BETA_SCHEDULER="cosine", STRATEGY="ddim20",
                           NAME="in2INind", NUM_HEADS=int(g["H"]), CFG_WEIGHT=float(g["cfg_scale"]))), "individual")
    sd = {"decoder.net_individual." + k: v for k, v in w("ind.").items()}
    sd.update({k: v for k, v in W.items() if not k.startswith(("clipTransEncoder.", "clip_ln."))})
    m.load_state_dict(sd)
    m = m.to("cuda:0")
    # the fixture's text modules are 32 wide with 4-head layers (the real ones: 768 wide, 8 heads): check text_process alone against the reference vector
    m.text_num_heads = int(gt["H"])
    out = m.text_process({"tokens_text": tt("tokens").long()}, "individual", "text", "cond_individual_individual1")
    np.testing.assert_allclose(out["cond_individual_individual1"].cpu().numpy(), gt["in2in:individual:cond"], atol=2e-5, rtol=1e-4)
    out = m.text_process({"tokens_text": tt("tokens").long()}, "interaction", "text", "c")
    np.testing.assert_allclose(out["c"].cpu().numpy(), gt["in2in:interaction:cond"], atol=2e-5, rtol=1e-4)
    with pytest.raises(ValueError, match="Mode not recognized"):
        m.text_process({"tokens_text": tt("tokens").long()}, "dual", "text", "c2")
