"""CPU: the oracle (oracle/) against golden vectors captured from the reference (tests/golden/make_golden.py).

Tolerances: the oracle executes the same torch CPU kernels in (almost) the same order as the reference, so
layer-level agreement is ~1e-6; geometry is compared at 2e-5 absolute (atan2/sqrt chains), loops at 1e-4.
"""
import numpy as np
import pytest
import torch

from oracle import schedule as S
from oracle import layers as L
from oracle import geometry as G
from oracle import denoiser as DN
from oracle import mixer as MX
from oracle import encoder as EN

torch.set_grad_enabled(False)


def close(a, b, atol=2e-6, rtol=1e-5):
    a = a.numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    np.testing.assert_allclose(a, b, atol=atol, rtol=rtol)


def close_frac(a, b, atol=3e-5, rtol=1e-4, frac=5e-4, hard=1e-2):
    """Geometry-amplified comparisons: all but `frac` of the elements within atol/rtol, none beyond `hard`.
    (rot6d re-orthonormalisation of N(0,1) inputs is ill-conditioned for a few elements)"""
    a = a.numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    d = np.abs(a - b)
    bad = d > atol + rtol * np.abs(b)
    assert bad.mean() <= frac, f"{bad.sum()} / {bad.size} outside tolerance (max {d.max():.3g})"
    assert d.max() <= hard, d.max()


# ---- G1 -----------------------------------------------------------------------------------------
def test_schedule_tables(golden):
    g, _, _ = golden("schedule")
    np.testing.assert_array_equal(S.cosine_betas(1000), g["betas_cosine_1000"])
    np.testing.assert_array_equal(S.linear_betas(1000), g["betas_linear_1000"])
    for strat in ["ddim50", "ddim1000", "ddim20"]:
        sc = S.make_schedule("cosine", 1000, strat)
        np.testing.assert_array_equal(np.array(sc.timestep_map), g[strat + ":timestep_map"])
        for a in ["betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod"]:
            np.testing.assert_array_equal(getattr(sc, a), g[strat + ":" + a], err_msg=a)
    np.testing.assert_array_equal(sorted(S.space_timesteps(1000, [1000])), g["space:[1000]"])
    np.testing.assert_array_equal(sorted(S.space_timesteps(300, "10,15,20")), g["space:10,15,20@300"])
    np.testing.assert_array_equal(sorted(S.space_timesteps(1000, "ddim25")), g["space:ddim25@1000"])
    with pytest.raises(ValueError):
        S.space_timesteps(1000, "ddim999")


# ---- G2 -----------------------------------------------------------------------------------------
def test_pe_rows(golden):
    g, _, _ = golden("pe")
    for D in [64, 512, 1024]:
        np.testing.assert_array_equal(L.pe_table(D)[g["rows"]].numpy(), g[f"pe{D}"])


# ---- G3 -----------------------------------------------------------------------------------------
def test_layers(golden):
    g, w, t = golden("layers")
    H = int(g["H"])
    x, y, e1, e2 = t("x"), t("y"), t("emb"), t("emb2")
    close(L.adaln(w("adaln."), "", x, e1) if False else L.adaln({"n." + k: v for k, v in w("adaln.").items()}, "n", x, e1), g["adaln:out"])
    pre = lambda W, p: {p + "." + k: v for k, v in W.items()}
    close(L.self_attention(pre(w("sa."), "m"), "m", x, e1, H), g["sa:out"])
    close(L.cross_attention(pre(w("ca."), "m"), "m", x, y, e1, H), g["ca:out"])
    close(L.ffn(pre(w("ffn."), "m"), "m", x, e1), g["ffn:out"])
    close(L.block_double_cond(pre(w("bdc_ind."), "m"), "m", "individual", x, None, e1, None, H), g["bdc_ind:out"])
    close(L.block_double_cond(pre(w("bdc_int."), "m"), "m", "interaction", x, y, e1, e2, H), g["bdc_int:out"])
    close(L.block(pre(w("blk."), "m"), "m", x, y, e1, H), g["blk:out"])
    close(L.influence_block(pre(w("ibc."), "m"), "m", x, y, e1, e2, H), g["ibc:out"])


# ---- G4 -----------------------------------------------------------------------------------------
def test_denoisers(golden):
    g, w, t = golden("denoisers")
    H = int(g["H"])
    ts = torch.from_numpy(g["t"]).long()
    pe = [("sequence_pos_encoder.pe", 16)]
    close(DN.in2in_denoiser(w("ind.", pe), "", "individual", t("x_ind"), ts, t("cond_ind"), H), g["ind:out"], atol=5e-6)
    close(DN.in2in_denoiser(w("int.", pe), "", "interaction", t("x_int"), ts, t("cond_int"), H), g["int:out"], atol=5e-6)
    close(DN.inter_denoiser(w("ig.", pe), "", t("x_int"), ts, t("cond_int"), H), g["ig:out"], atol=5e-6)


# ---- G5 -----------------------------------------------------------------------------------------
def test_influence_modes(golden):
    g, w, t = golden("influence")
    H = int(g["H"])
    base = w("m4.")
    for mode in [1, 2, 3, 4]:
        W = dict(base)
        W.update(w(f"m{mode}."))
        out = DN.influence(W, "", mode, t("m_i"), t("m_I"), t("cond_i"), t("cond_I"), H)
        close(out, g[f"m{mode}:out"])


# ---- G6 -----------------------------------------------------------------------------------------
GEO_TOL = dict(atol=2e-5, rtol=1e-4)


def test_rotation_kats(golden):
    g, _, t = golden("geometry")
    mat = G.rotation_6d_to_matrix(t("kat:d6"))
    close(mat, g["kat:d6_matrix"], **GEO_TOL)
    q = G.matrix_to_quaternion(t("kat:d6_matrix"))
    close(q, g["kat:matrix_quat"], **GEO_TOL)
    close(G.quaternion_to_axis_angle(t("kat:matrix_quat")), g["kat:quat_aa"], **GEO_TOL)
    close(G.axis_angle_to_quaternion(t("kat:aa")), g["kat:aa_quat"], **GEO_TOL)
    close(G.quaternion_to_matrix(G.axis_angle_to_quaternion(t("kat:aa"))), g["kat:aa_matrix"], **GEO_TOL)
    close(G.matrix_to_rotation_6d(t("kat:aa_matrix")), g["kat:aa_d6"], **GEO_TOL)
    close(G.qbetween(t("kat:v0"), t("kat:v1")), g["kat:qbetween"], **GEO_TOL)
    close(G.qrot(t("kat:qbetween"), t("kat:v0")), g["kat:qrot"], **GEO_TOL)


@pytest.mark.parametrize("name", ["rand_a", "rand_b", "valid_c", "valid_d"])
def test_ih_smpl_center(golden, name):
    g, _, t = golden("geometry")
    s = G.ih_to_smpl(t(name))
    close(s, g[name + ":ih_to_smpl"], **GEO_TOL)
    close(G.smpl_to_ih(t(name + ":ih_to_smpl")), g[name + ":smpl_to_ih"], **GEO_TOL)
    cm = G.center_motion(t(name + ":ih_to_smpl"))
    close(cm, g[name + ":center"], **GEO_TOL)
    ih = G.smpl_to_ih(t(name + ":center"))
    close(ih, g[name + ":center_ih"], **GEO_TOL)
    assert np.all(g[name + ":center_ih"][..., 258:] == 0)       # feet channels become 0
    assert np.all(ih.numpy()[..., 258:] == 0)


@pytest.mark.parametrize("name,m1,m2", [("rand", "rand_a", "rand_b"), ("valid", "valid_c", "valid_d")])
def test_align(golden, name, m1, m2):
    g, _, t = golden("geometry")
    moved = G.align_motions(t(m1 + ":ih_to_smpl"), t(m2 + ":ih_to_smpl"))
    assert moved.shape[-1] == 201
    close(moved, g[name + ":align_m2"], **GEO_TOL)
    close(t(m1 + ":ih_to_smpl"), g[name + ":align_m1"], atol=0, rtol=0)  # motion1 returned unchanged (quirk 3)
    close(G.smpl_to_ih(moved), g[name + ":align_m2_ih"], **GEO_TOL)


# ---- G7-G10 -------------------------------------------------------------------------------------
def _mixer_ctx(golden):
    g, w, t = golden("mixer")
    stats = tuple(t(k) for k in ["mean_hml", "std_hml", "mean_ih", "std_ih"])
    pes = [("sequence_pos_encoder.pe", 16), ("denoiser1.sequence_pos_encoder.pe", 16), ("denoiser2.sequence_pos_encoder.pe", 16)]
    W = w("mix.", pes)
    return g, w, t, stats, W


MIX_TOL = dict(atol=3e-5, rtol=1e-4)


@pytest.mark.parametrize("mode", [1, 2, 3, 4])
@pytest.mark.parametrize("align", [True, False])
def test_mixer_forward(golden, mode, align):
    g, w, t, stats, W = _mixer_ctx(golden)
    W = dict(W)
    if mode in (1, 2):
        W.update({k: v for k, v in w("mix_out1.").items()})
    for force in ([None, 0.0, 1.0] if mode == 4 else [None]):
        spec = MX.MixerSpec(d_heads=int(g["d_heads"]), m_heads=int(g["m_heads"]), mixing_mode=mode, align=align, force_influence_val=force)
        hist = {}
        out = MX.mixer_forward(W, spec, stats, t("x1"), t("t").long(), t("cond"), t("x2"), hist)
        tag = f"fwd:m{mode}:a{int(align)}:f{force}"
        close_frac(out, g[tag], **MIX_TOL)
        if mode == 4 and force is None:
            for k in ["influence_i1", "influence_i2", "out1", "out2"]:
                close_frac(hist[k][0], g[tag + ":" + k], **MIX_TOL)
            if align:
                assert np.all(g[tag + ":out1"][..., 258:262] == 0) and np.all(g[tag + ":out1"][..., 520:524] == 0)


def test_mixer_forward_intergen(golden):
    g, w, t, stats, W = _mixer_ctx(golden)
    spec = MX.MixerSpec(d_heads=int(g["d_heads"]), m_heads=int(g["m_heads"]), model2="InterGen")
    close_frac(MX.mixer_forward(W, spec, stats, t("x1"), t("t").long(), t("cond"), t("x2")), g["fwd:intergen"], **MIX_TOL)


def test_cfg_and_ddim_step(golden):
    g, w, t, stats, W = _mixer_ctx(golden)
    spec = MX.MixerSpec(d_heads=int(g["d_heads"]), m_heads=int(g["m_heads"]))
    s = float(g["cfg_scale"])
    B = t("cfg_x").shape[0]
    out = MX.cfg_x2(W, spec, stats, s, t("cfg_x"), t("cfg_x2"), torch.full((B,), 640, dtype=torch.long), t("cfg_cond"))
    close_frac(out, g["cfg:out"], atol=1e-4, rtol=1e-4)
    sched = S.make_schedule("cosine", 1000, "ddim50")
    for i in [32, 0]:
        r = MX.mixer_ddim_step(W, spec, stats, sched, s, i, t("cfg_x"), t("cfg_x2"), t("cfg_cond"))
        for k, v in zip(["sample", "sample2", "pred_xstart", "pred_xstart2"], r):
            close_frac(v, g[f"ddim:i{i}:{k}"], atol=2e-4, rtol=2e-4)
    # quirk 6: at t==0 both pred_xstart are the raw model output
    np.testing.assert_array_equal(g["ddim:i0:pred_xstart"], g["ddim:i0:pred_xstart2"])


@pytest.mark.parametrize("strat", ["ddim50", "ddim20"])
def test_mixer_loop(golden, strat):
    g, w, t, stats, W = _mixer_ctx(golden)
    spec = MX.MixerSpec(d_heads=int(g["d_heads"]), m_heads=int(g["m_heads"]))
    sched = S.make_schedule("cosine", 1000, strat)
    hist = {}
    out, _, _ = MX.mixer_ddim_loop(W, spec, stats, sched, float(g["cfg_scale"]), t(f"loop:{strat}:x_T"), t("cfg_cond"), hist)
    ref = g[f"loop:{strat}:output"]
    d = np.abs(out.numpy() - ref)
    # End-to-end tolerance is distributional: on this tiny random model a 1e-6 perturbation of x_T
    # moves the output by mean 4e-4 / max 4e-2 (measured with the oracle itself; the last DDIM steps amplify ~50x),
    # and oracle-vs-reference lands at mean 5e-4.  Tight checks are the per-step ones above and the histories below.
    assert d.mean() <= 2e-3 and np.percentile(d, 99) <= 3e-2, (d.mean(), d.max())
    n = int(g[f"loop:{strat}:nsteps"])
    assert len(hist["out1"]) == n == sched.num_timesteps
    assert hist["influence_i1"][0].shape == (2 * out.shape[0], out.shape[1], 262)      # quirk 12: CFG-doubled batch
    for name in ["influence_i1", "influence_i2", "out1", "out2", "out_influenced"]:
        sums = np.array([float(v.double().abs().sum()) for v in hist[name]])
        np.testing.assert_allclose(sums[:-1], g[f"loop:{strat}:{name}:abssum"][:-1], rtol=1e-4)
        np.testing.assert_allclose(sums, g[f"loop:{strat}:{name}:abssum"], rtol=2e-3)
        if strat == "ddim50":
            close_frac(hist[name][0], g[f"loop:{strat}:{name}:0"], atol=1e-4, rtol=1e-4)


# ---- G11 ----------------------------------------------------------------------------------------
def test_single_chain(golden):
    g, w, t = golden("single")
    W = w("ind.", [("sequence_pos_encoder.pe", 16)])
    H, s = int(g["H"]), float(g["cfg_scale"])
    B = t("x_T").shape[0]
    close(MX.cfg_single(W, "", "individual", s, t("x_T"), torch.full((B,), 500, dtype=torch.long), t("cond"), H), g["cfg:out"], atol=2e-5)
    for strat in ["ddim50", "ddim20"]:
        out, _ = MX.single_ddim_loop(W, "", "individual", S.make_schedule("cosine", 1000, strat), s, t("x_T"), t("cond"), H)
        d = np.abs(out.numpy() - g[f"loop:{strat}:output"])
        assert d.mean() <= 1e-4 and d.max() <= 1e-2, (d.mean(), d.max())


# ---- G12 ----------------------------------------------------------------------------------------
def test_interaction_standalone_4way_cfg(golden):
    g, w, t = golden("interaction")
    W = w("int.", [("sequence_pos_encoder.pe", 16)])
    H = int(g["H"])
    s, si, sd = float(g["s"]), float(g["s_int"]), float(g["s_ind"])
    B = t("x_T").shape[0]
    out = MX.cfg_multiple(W, "", s, si, sd, t("x_T"), torch.full((B,), 500, dtype=torch.long), t("cond"), H)
    close(out, g["cfg:out"], atol=5e-5, rtol=1e-4)
    out, _ = MX.interaction_ddim_loop(W, "", S.make_schedule("cosine", 1000, "ddim20"), s, si, sd, t("x_T"), t("cond"), H)
    d = np.abs(out.numpy() - g["loop:ddim20:output"])
    assert d.mean() <= 1e-4 and d.max() <= 1e-2, (d.mean(), d.max())


# ---- G13 ----------------------------------------------------------------------------------------
def test_dual_sampler(golden):
    g, w, t = golden("dual")
    W = w("ind.", [("sequence_pos_encoder.pe", 16)])
    W = {"ind." + k: v for k, v in W.items()}
    W.update({"int." + k: v for k, v in w("int.", [("sequence_pos_encoder.pe", 16)]).items()})
    H, si, sI = int(g["H"]), float(g["s_ind"]), float(g["s_int"])
    B = t("x_T").shape[0]
    close(DN.in2in_denoiser(W, "ind.", "dual_individual", t("x_T"), torch.tensor([500, 20]), t("cond"), H), g["fwd:dual_individual"], atol=1e-5)
    ts = torch.full((B,), 500, dtype=torch.long)
    for func in ["exp", "lin", "const", "exp-inv"]:
        wv = MX.dual_weight(func, float(g[f"cfg:{func}:value"]), 500)
        close(MX.cfg_dual(W, "ind.", "int.", si, sI, wv, t("x_T"), ts, t("cond"), H), g[f"cfg:{func}:out"], atol=5e-5, rtol=1e-4)
    with pytest.raises(ValueError):
        MX.dual_weight("cos", 1.0, 10)
    sched = S.make_schedule("cosine", 1000, "ddim20")
    for func in ["exp", "lin"]:
        out, _ = MX.dual_ddim_loop(W, "ind.", "int.", sched, si, sI, func, float(g[f"cfg:{func}:value"]), t("x_T"), t("cond"), H)
        d = np.abs(out.numpy() - g[f"loop:{func}:ddim20:output"])
        assert d.mean() <= 1e-4 and d.max() <= 1e-2, (d.mean(), d.max())


# ---- G14 ----------------------------------------------------------------------------------------
def _mixmdm_weights(golden):
    gm, wm, _ = golden("mixer")
    g, w, t = golden("mdm")
    W = wm("mix.", [("sequence_pos_encoder.pe", 16), ("denoiser2.sequence_pos_encoder.pe", 16)])
    W = {k: v for k, v in W.items() if not k.startswith("denoiser1.")}
    W.update(w("mixmdm.", [("denoiser1.sequence_pos_encoder.pe", 16)]))
    stats = tuple(torch.from_numpy(gm[k]) for k in ["mean_hml", "std_hml", "mean_ih", "std_ih"])
    return g, t, W, stats


def test_mdm_denoiser_and_mixer_with_mdm(golden):
    g, w, t = golden("mdm")
    W = w("mdm.", [("sequence_pos_encoder.pe", 16)])
    H = int(g["H"])
    cond = t("cond").clone()
    close(EN.mdm_denoiser(W, "", t("x"), t("t").long(), cond, H), g["mdm:out"], atol=1e-5, rtol=1e-4)
    assert torch.equal(cond, t("cond"))                      # the oracle does not reproduce the in-place add on the argument
    g, t, W, stats = _mixmdm_weights(golden)
    spec = MX.MixerSpec(d_heads=2, m_heads=2, model1="MDM", d1_text_dim=16)
    hist = {}
    out = MX.mixer_forward(W, spec, stats, t("mix_x1"), t("mix_t").long(), t("mix_cond"), t("mix_x2"), hist)
    close_frac(hist["out1"][0], g["fwd:mixmdm:out1"], **MIX_TOL)
    close_frac(out, g["fwd:mixmdm"], **MIX_TOL)
    out, _, _ = MX.mixer_ddim_loop(W, spec, stats, S.make_schedule("cosine", 1000, "ddim20"), 3.5, t("loop_x_T"), t("loop_cond"))
    d = np.abs(out.numpy() - g["loop:ddim20:output"])
    assert d.mean() <= 2e-3 and np.percentile(d, 99) <= 3e-2, (d.mean(), d.max())


# ---- G15 ----------------------------------------------------------------------------------------
def test_text_heads(golden):
    """Everything in text_process around the (third-party, identity here) CLIP residual tower."""
    g, w, t = golden("text")
    W = w("txt.")
    H, tok = int(g["H"]), t("tokens").long()
    clip_out = EN.clip_text_tower(W, "", tok, H)          # no resblocks keys in the fixture -> embedding + ln_final only
    for enc, ln, key in [("clipTransEncoder.", "clip_ln", "mixer:cond"), ("clipTransEncoder_individual.", "clip_ln_individual", "in2in:individual:cond"),
                         ("clipTransEncoder_interaction.", "clip_ln_interaction", "in2in:interaction:cond"), ("clipTransEncoder.", "clip_ln", "intergen:cond")]:
        close(EN.text_head(W, enc, ln, clip_out, tok, H), g[key], atol=1e-5, rtol=1e-4)


# ---- G16 ----------------------------------------------------------------------------------------
def test_mixer32_fixture(golden):
    """The latent-32 fixture (pins the bf16-matrix-core GEMM modes on the GPU side): oracle == reference."""
    g, w, t = golden("mixer32")
    W = w("mix.", [("sequence_pos_encoder.pe", 32), ("denoiser1.sequence_pos_encoder.pe", 32), ("denoiser2.sequence_pos_encoder.pe", 32)])
    stats = tuple(torch.from_numpy(g[k]) for k in ["mean_hml", "std_hml", "mean_ih", "std_ih"])
    spec = MX.MixerSpec(d_heads=2, m_heads=2)
    close_frac(MX.mixer_forward(W, spec, stats, t("x1"), t("t").long(), t("cond"), t("x2")), g["fwd"], **MIX_TOL)
    sched = S.make_schedule("cosine", 1000, "ddim20")
    r = MX.mixer_ddim_step(W, spec, stats, sched, 3.5, 12, t("x_T"), t("x_T"), t("cfg_cond"))
    for k, v in zip(["sample", "sample2", "pred_xstart", "pred_xstart2"], r):
        close_frac(v, g[f"ddim:i12:{k}"], atol=2e-4, rtol=2e-4)


# ---- full model dimensions -----------------------------------------------------------------------
def test_fulldims_fixture_pins_the_oracle_at_the_real_model_sizes():
    """tests/golden/fulldims.npz: oracle == REFERENCE at D=1024/F=2048/L=8/H=8 (dh=128) + mixer 512/1024/4/8 (dh=64): Mixer.forward and
    ddim_sample (i=32, i=0) at B=2, T=32, and ddim1000 steps at the headline length T=300 (B=1; i=999 from x_T and a late step i=3)."""
    from conftest import fulldims_case
    g, _, W, stats, inp = fulldims_case()
    spec = MX.MixerSpec(d_heads=8, m_heads=8)
    x1, x2, cond, tt = inp["fwd"]
    hist = {}
    out = MX.mixer_forward(W, spec, stats, x1, torch.full((x1.shape[0],), tt, dtype=torch.long), cond, x2, hist)
    close_frac(out, g["fwd"], **MIX_TOL)
    close_frac(hist["influence_i1"][0][..., [0, 3, 66, 132, 258]], g["fwd:influence_i1"], atol=2e-5, rtol=1e-4)
    cb, xT, xb2 = inp["step"]
    sched = S.make_schedule("cosine", 1000, "ddim50")
    for i in [32, 0]:
        r = MX.mixer_ddim_step(W, spec, stats, sched, 3.5, i, xT, xb2, cb)
        for k, v in zip(["sample", "sample2", "pred_xstart", "pred_xstart2"], r):
            if k != "pred_xstart":
                close_frac(v, g[f"ddim50:i{i}:{k}"], atol=2e-4, rtol=2e-4)
    c300, x300 = inp["t300"]
    sched = S.make_schedule("cosine", 1000, "ddim1000")
    r = MX.mixer_ddim_step(W, spec, stats, sched, 3.5, 999, x300, x300, c300)
    close_frac(r[0], g["ddim1000:T300:i999:sample"], atol=2e-4, rtol=2e-4)
    close_frac(r[1], g["ddim1000:T300:i999:sample2"], atol=2e-4, rtol=2e-4)
    xa, xb = inp["late"]
    r = MX.mixer_ddim_step(W, spec, stats, sched, 3.5, 3, xa, xb, c300)
    close_frac(r[0], g["ddim1000:T300:i3:sample"], atol=2e-4, rtol=2e-4)
    close_frac(r[1], g["ddim1000:T300:i3:sample2"], atol=2e-4, rtol=2e-4)
