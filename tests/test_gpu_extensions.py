"""GPU: the SURVEY 8f rows -- MDMDenoiser as MODEL1, the in2IN "dual" sampler (ClassifierFreeSampleDualMDM), and the
text-conditioning stage (clipTransEncoder heads + CLIP-style tower) -- against reference-captured goldens
(tests/golden/{mdm,dual,text}.npz) and the CPU oracle.
"""
import math
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import mixer as MX            # noqa: E402  (checker only)
from oracle import denoiser as DN         # noqa: E402
from oracle import encoder as EN          # noqa: E402
from oracle import schedule as OS         # noqa: E402
from test_gpu_kernels import assert_close, rnd   # noqa: E402
from test_gpu_kernels import dev as _device          # noqa: E402


def dev(t):
    return t.to(_device())


from test_gpu_sampler import STEP_TOL     # noqa: E402


# ---- kernels ----------------------------------------------------------------------------------------
@pytest.mark.parametrize("rows,D", [(5, 16), (77, 768), (301, 256), (9, 1024), (3, 2048), (20000, 1024), (8193, 64)])
def test_layernorm(rows, D):
    from mixermdm_amd import ops
    x, g, b = rnd(1, rows, D) * 3 + 0.5, rnd(2, D), rnd(3, D)
    out = ops.layernorm(dev(x), dev(g), dev(b), 1e-5)
    assert_close(out, F.layer_norm(x, (D,), g, b, 1e-5), atol=2e-5, rtol=1e-5, what="layernorm")


def sdpa_ref(q, k, v, H, causal):
    n, Tq, HD = q.shape
    dh = HD // H
    sp = lambda t: t.view(n, -1, H, dh).transpose(1, 2)
    s = (sp(q) @ sp(k).transpose(-1, -2)) / math.sqrt(dh)
    if causal:
        s = s + torch.full((Tq, Tq), float("-inf")).triu_(1)
    return (torch.softmax(s, -1) @ sp(v)).transpose(1, 2).reshape(n, Tq, HD)


@pytest.mark.parametrize("dh,H,T,causal", [(64, 2, 77, True), (64, 2, 77, False), (128, 1, 301, False), (64, 3, 130, True), (96, 2, 77, False),
                                           (96, 1, 19, True), (8, 2, 13, False), (16, 2, 10, True), (48, 1, 5, False)])
def test_attention_without_zero_key(dh, H, T, causal):
    """Plain softmax (no add_zero_attn) with an optional causal mask, on the MFMA kernel (64/128) and both scalar fallbacks."""
    from mixermdm_amd import ops
    n = 3
    qkv = rnd(7, n, T, 3 * H * dh)
    ref = sdpa_ref(qkv[..., :H * dh], qkv[..., H * dh:2 * H * dh], qkv[..., 2 * H * dh:], H, causal)
    d = dev(qkv)
    out = ops.attention(d[..., :H * dh], d[..., H * dh:2 * H * dh], d[..., 2 * H * dh:], H, zero_key=False, causal=causal)
    assert_close(out, ref, atol=2e-5, rtol=1e-4, what=f"attention dh={dh} causal={causal}")


def test_attention_flag_errors():
    from mixermdm_amd import ops
    from mixermdm_amd._lib import MMDMError
    q = dev(rnd(1, 1, 8, 64))
    with pytest.raises(MMDMError, match="causal"):
        ops.attention(q, q, q, 1, zero_key=True, causal=True)


def enc_weights(seed, D, Fd, std=0.08):
    shapes = {"self_attn.in_proj_weight": (3 * D, D), "self_attn.in_proj_bias": (3 * D,), "self_attn.out_proj.weight": (D, D), "self_attn.out_proj.bias": (D,),
              "linear1.weight": (Fd, D), "linear1.bias": (Fd,), "linear2.weight": (D, Fd), "linear2.bias": (D,),
              "norm1.weight": (D,), "norm1.bias": (D,), "norm2.weight": (D,), "norm2.bias": (D,)}
    W = {}
    for i, (k, sh) in enumerate(shapes.items()):
        W[k] = rnd(seed + i, *sh) * std + (1.0 if k.endswith("norm1.weight") or k.endswith("norm2.weight") else 0.0)
    return W


@pytest.mark.parametrize("D,H,Fd,T", [(32, 4, 64, 10), (128, 2, 256, 77), (768, 8, 2048, 77), (256, 4, 1024, 301)])
def test_encoder_layer_post_norm_vs_oracle(D, H, Fd, T):
    """nn.TransformerEncoderLayer (post-norm, GELU): clipTransEncoder (768/8 -> 96-wide heads) and MDM's seqTransEncoder shapes."""
    from mixermdm_amd import ops
    W = enc_weights(20, D, Fd)
    x = rnd(40, 2, T, D)
    ref = EN.encoder_layer({"l." + k: v for k, v in W.items()}, "l", x, H)
    out = ops.encoder_layer_(dev(x), {k.replace("self_attn.", ""): dev(v) for k, v in W.items()}, H)
    assert_close(out, ref, atol=1e-4, rtol=1e-4, what="encoder layer")


def clip_weights(seed, V, ctx, D, L, Fd):
    W = {"token_embedding.weight": rnd(seed, V, D) * 0.1, "positional_embedding": rnd(seed + 1, ctx, D) * 0.05,
         "ln_final.weight": 1 + rnd(seed + 2, D) * 0.1, "ln_final.bias": rnd(seed + 3, D) * 0.1, "text_projection": rnd(seed + 4, D, 24) * 0.1}
    for i in range(L):
        e = enc_weights(seed + 100 * (i + 1), D, Fd)
        m = {"attn.in_proj_weight": "self_attn.in_proj_weight", "attn.in_proj_bias": "self_attn.in_proj_bias", "attn.out_proj.weight": "self_attn.out_proj.weight",
             "attn.out_proj.bias": "self_attn.out_proj.bias", "mlp.c_fc.weight": "linear1.weight", "mlp.c_fc.bias": "linear1.bias", "mlp.c_proj.weight": "linear2.weight",
             "mlp.c_proj.bias": "linear2.bias", "ln_1.weight": "norm1.weight", "ln_1.bias": "norm1.bias", "ln_2.weight": "norm2.weight", "ln_2.bias": "norm2.bias"}
        for ck, ek in m.items():
            W[f"clip_transformer.resblocks.{i}.{ck}"] = e[ek]
    return W


def test_clip_style_tower_vs_oracle():
    """Pre-norm causal QuickGELU blocks + ln_final + encode_text at CLIP's head size (dh = 64) and context (77).  The oracle side is a
    restatement of the published CLIP architecture (PARITY UNPINNED: the clip package is not under /root/reference)."""
    from mixermdm_amd.text import ClipTextTower
    V, ctx, D, L, H = 60, 77, 128, 3, 2
    W = clip_weights(300, V, ctx, D, L, 4 * D)
    g = torch.Generator().manual_seed(5)
    tok = torch.randint(1, V - 1, (3, ctx), generator=g)
    for b, e in enumerate([5, 76, 30]):
        tok[b, e] = V - 1
        tok[b, e + 1:] = 0
    tower = ClipTextTower(W, "", num_heads=H)
    assert_close(tower(tok), EN.clip_text_tower(W, "", tok, H), atol=2e-4, rtol=2e-4, what="clip tower")
    assert_close(tower.encode_text(tok), EN.clip_encode_text(W, "", tok, H), atol=2e-4, rtol=2e-4, what="encode_text")


# ---- text heads vs the reference ----------------------------------------------------------------------
def test_text_heads_vs_reference_golden(golden):
    from mixermdm_amd.text import ClipTextTower, TextHead
    g, w, t = golden("text")
    W = w("txt.")
    H, tok = int(g["H"]), t("tokens").long()
    clip_out = ClipTextTower(W, "", num_heads=H)(tok)       # the fixture's tower has no residual blocks (identity stand-in)
    for enc, ln, key in [("clipTransEncoder.", "clip_ln", "mixer:cond"), ("clipTransEncoder_individual.", "clip_ln_individual", "in2in:individual:cond"),
                         ("clipTransEncoder_interaction.", "clip_ln_interaction", "in2in:interaction:cond"), ("clipTransEncoder.", "clip_ln", "intergen:cond")]:
        assert_close(TextHead(W, enc, ln, H)(clip_out, tok), t(key), atol=2e-5, rtol=1e-4, what=key)


def test_generate_cond_layout_vs_oracle():
    """MixerMDM.generate_cond: 8 slices in the reference's order (mixermdm.py:342-354), tower shared by the three distinct prompts."""
    from mixermdm_amd.text import MixerTextEncoder
    V, ctx, D, H = 40, 20, 64, 2
    W = clip_weights(500, V, ctx, D, 1, 2 * D)
    for j, (pfx, ln) in enumerate([("clipTransEncoder.", "clip_ln"), ("model1.clipTransEncoder_individual.", "model1.clip_ln_individual"),
                                   ("model2.clipTransEncoder_interaction.", "model2.clip_ln_interaction")]):
        for i in range(2):
            for k, v in enc_weights(700 + 50 * j + 10 * i, D, 2 * D).items():
                W[f"{pfx}layers.{i}.{k}"] = v
        W[ln + ".weight"], W[ln + ".bias"] = 1 + rnd(800 + j, D) * 0.1, rnd(810 + j, D) * 0.1
    g = torch.Generator().manual_seed(9)
    mk = lambda e: torch.cat([torch.randint(1, V - 1, (2, e), generator=g), torch.full((2, 1), V - 1), torch.zeros(2, ctx - e - 1, dtype=torch.long)], 1)
    t1, t2, tI = mk(4), mk(9), mk(17)
    enc = MixerTextEncoder(W, clip_heads=H, head_heads=H)
    cond = enc.generate_cond({"tokens_text_individual1": t1, "tokens_text_individual2": t2, "tokens_text": tI})
    assert cond.shape == (2, 8 * D)
    c = {k: EN.clip_text_tower(W, "", tk, H) for k, tk in [("1", t1), ("2", t2), ("I", tI)]}
    hd = lambda pfx, ln, key, tk: EN.text_head(W, pfx, ln, c[key], tk, H)
    mi, m1, m2 = ("clipTransEncoder.", "clip_ln"), ("model1.clipTransEncoder_individual.", "model1.clip_ln_individual"), ("model2.clipTransEncoder_interaction.", "model2.clip_ln_interaction")
    ref = torch.cat([hd(*m2, "I", tI), hd(*m2, "1", t1), hd(*m2, "2", t2), hd(*m1, "1", t1), hd(*m1, "2", t2),
                     hd(*mi, "I", tI), hd(*mi, "1", t1), hd(*mi, "2", t2)], dim=1)
    assert_close(cond, ref, atol=1e-4, rtol=1e-4, what="generate_cond")


# ---- MDMDenoiser as MODEL1 ------------------------------------------------------------------------------
def mixmdm_sampler(golden, max_batch=2):
    from mixermdm_amd.sampler import Sampler
    gm, wm, _ = golden("mixer")
    g, w, t = golden("mdm")
    W = {k: v for k, v in wm("mix.").items() if not k.startswith("denoiser1.")}
    W.update(w("mixmdm."))
    s = Sampler(d_latent=16, d_ff=32, d_layers=2, d_heads=2, m_latent=16, m_ff=32, m_layers=2, m_heads=2, cfg_scale=3.5, model1_kind=1,
                max_batch=max_batch, max_frames=16)
    s.load_state_dict(W)
    s.set_norm_stats(gm["mean_hml"], gm["std_hml"], gm["mean_ih"], gm["std_ih"])
    s.prepare()
    return s, g, t


def test_mdm_denoiser_vs_reference_golden(golden):
    from mixermdm_amd.sampler import Sampler
    g, w, t = golden("mdm")
    s = Sampler(d_latent=16, d_ff=32, d_layers=2, d_heads=int(g["H"]), single_only=1, model1_kind=1, max_batch=1, max_frames=16)
    s.load_state_dict({"denoiser1." + k: v for k, v in w("mdm.").items()})
    s.prepare()
    x, c, ref = t("x"), t("cond"), t("mdm:out")
    for b, tt in enumerate(g["t"]):          # module_forward shares one t across rows
        out = s.module_forward(0, torch.cat([x[b:b + 1]] * 2), torch.cat([c[b:b + 1]] * 2), int(tt))
        assert_close(out[0], ref[b], atol=2e-5, rtol=1e-4, what=f"MDMDenoiser t={tt}")
    s.close()


def test_mixer_with_mdm_vs_reference_golden(golden):
    s, g, t = mixmdm_sampler(golden)
    out = s.module_forward(2, t("mix_x1"), t("mix_cond"), int(g["mix_t"][0]), x2=t("mix_x2"))
    assert_close(out, t("fwd:mixmdm"), what="Mixer.forward (MDM as model1)", **STEP_TOL)
    s.set_schedule("ddim20")
    for graph in (False, True):
        out = s.sample(t("loop_cond"), t("loop_x_T"), use_graph=graph)
        d = np.abs(out.cpu().numpy() - g["loop:ddim20:output"])
        assert d.mean() <= 2e-3 and np.percentile(d, 99) <= 3e-2, (d.mean(), d.max())
    s.close()


def test_standalone_mdm_chain_vs_oracle(golden):
    """MDM.forward (mdm.py:190-222): 2-way CFG single chain on MDMDenoiser, at a head size that uses the MFMA attention (T+1 = 41 keys)."""
    from mixermdm_amd.sampler import Sampler
    D, Fd, L, H, B, T = 128, 256, 2, 2, 2, 40
    W = {"input_process.poseEmbedding.weight": rnd(1, D, 262) * 0.05, "input_process.poseEmbedding.bias": rnd(2, D) * 0.05,
         "output_process.poseFinal.weight": rnd(3, 262, D) * 0.05, "output_process.poseFinal.bias": rnd(4, 262) * 0.05}
    for k in ("0", "2"):
        W[f"embed_timestep.time_embed.{k}.weight"], W[f"embed_timestep.time_embed.{k}.bias"] = rnd(5 + int(k), D, D) * 0.05, rnd(8 + int(k), D) * 0.05
    for i in range(L):
        for k, v in enc_weights(50 + 20 * i, D, Fd).items():
            W[f"seqTransEncoder.layers.{i}.{k}"] = v
    s = Sampler(d_latent=D, d_ff=Fd, d_layers=L, d_heads=H, single_only=1, model1_kind=1, cfg_scale=2.5, max_batch=B, max_frames=T)
    s.load_state_dict({"denoiser1." + k: v for k, v in W.items()})
    s.prepare()
    s.set_schedule("ddim20")
    cond, xT = rnd(70, B, D), rnd(71, B, T, 262)
    s.begin(cond, xT)
    s.run(1, use_graph=True)
    Wo = dict(W)
    from oracle.layers import pe_table
    Wo["sequence_pos_encoder.pe"] = pe_table(D)
    osch = OS.make_schedule("cosine", 1000, "ddim20")
    ts = torch.full((2 * B,), osch.timestep_map[19], dtype=torch.long)
    o = EN.mdm_denoiser(Wo, "", torch.cat([xT, xT]), ts, torch.cat([cond, torch.zeros_like(cond)]), H)
    x0 = 2.5 * o[:B] + (1 - 2.5) * o[B:]
    st = s.state()
    assert_close(st["pred_xstart"], x0, atol=1e-4, rtol=1e-4, what="MDM cfg x0")
    assert_close(st["x"], MX.ddim_update(osch, 19, xT, x0), atol=1e-4, rtol=1e-4, what="MDM ddim x")
    s.close()


# ---- in2IN "dual" sampler ----------------------------------------------------------------------------
def dual_sampler(golden):
    from mixermdm_amd.sampler import Sampler
    g, w, t = golden("dual")
    s = Sampler(d_latent=16, d_ff=32, d_layers=2, d_heads=int(g["H"]), single_only=3, cfg_scale_individual=float(g["s_ind"]),
                cfg_scale_interaction=float(g["s_int"]), max_batch=2, max_frames=16)
    W = {"denoiser1." + k: v for k, v in w("ind.").items()}
    W.update({"denoiser2." + k: v for k, v in w("int.").items()})
    s.load_state_dict(W)
    s.prepare()
    Wo = {"ind." + k: v for k, v in w("ind.", [("sequence_pos_encoder.pe", 16)]).items()}
    Wo.update({"int." + k: v for k, v in w("int.", [("sequence_pos_encoder.pe", 16)]).items()})
    return s, g, t, Wo


def test_dual_individual_forward_vs_reference_golden(golden):
    """in2INDenoiser "dual_individual" incl. the reference quirk: person b only goes through the last block."""
    s, g, t, _ = dual_sampler(golden)
    x, c, ref = t("x_T"), t("cond"), t("fwd:dual_individual")
    for b, tt in enumerate([500, 20]):
        out = s.module_forward(3, torch.cat([x[b:b + 1]] * 2), torch.cat([c[b:b + 1]] * 2), tt)
        assert_close(out[0], ref[b], atol=2e-5, rtol=1e-4, what=f"dual_individual t={tt}")
    s.close()


def test_dual_sampler_vs_reference_golden_and_oracle(golden):
    from mixermdm_amd._lib import MMDMError
    s, g, t, Wo = dual_sampler(golden)
    H, si, sI = int(g["H"]), float(g["s_ind"]), float(g["s_int"])
    xT, cond = t("x_T"), t("cond")
    osch = s.set_schedule("ddim20")
    with pytest.raises(MMDMError, match="set_dual_weights"):
        s.begin(cond, xT)
    with pytest.raises(ValueError):
        s.set_dual_weights("cos", 1.0)
    # one step of every composition function against the (golden-pinned) oracle
    t19 = osch.timestep_map[19]
    for func in ["exp", "lin", "const", "exp-inv"]:
        val = float(g[f"cfg:{func}:value"])
        wtab = s.set_dual_weights(func, val)
        assert abs(float(wtab[19]) - float(MX.dual_weight(func, val, t19))) <= 1e-7
        s.begin(cond, xT)
        s.run(1, use_graph=False)
        x0 = MX.cfg_dual(Wo, "ind.", "int.", si, sI, MX.dual_weight(func, val, t19), xT, torch.full((2,), t19, dtype=torch.long), cond, H)
        assert_close(s.state()["pred_xstart"], x0, atol=1e-4, rtol=1e-4, what=f"dual cfg {func}")
    # full loops against the reference
    for func in ["exp", "lin"]:
        s.set_dual_weights(func, float(g[f"cfg:{func}:value"]))
        for graph in (False, True):
            out = s.sample(cond, xT, use_graph=graph)
            d = np.abs(out.cpu().numpy() - g[f"loop:{func}:ddim20:output"])
            assert out.shape == (2, 12, 524) and d.mean() <= 1e-4 and d.max() <= 1e-2, (func, d.mean(), d.max())
    s.close()


# ---- the text stage at the REAL sizes of CLIP ViT-L/14's text tower and the clipTransEncoder heads -----------------------------
def test_text_stage_at_vit_l14_dimensions_vs_oracle():
    """Width 768, 12 blocks x 12 heads (dh = 64), context 77, vocabulary 49 408 (clip.load("ViT-L/14@336px"), src/models/mixermdm.py:212-217)
    and the 2-layer 768 / 8-head (dh = 96: zero-padded onto the DH = 128 MFMA kernel) / ff 2048 heads (mixermdm.py:246-259): tower and
    generate_cond against the oracle's restatement, and a wall-clock figure for the whole 8-slice conditioning of a B = 16 batch.
    PARITY UNPINNED for the tower (the clip package and its weights are not under /root/reference); the heads are pinned by text.npz."""
    import time
    from mixermdm_amd import ops
    from mixermdm_amd._lib import load_library
    from mixermdm_amd.text import ClipTextTower, MixerTextEncoder
    V, ctx, D, L, H, HH = 49408, 77, 768, 12, 12, 8
    W = clip_weights(900, V, ctx, D, L, 4 * D)
    # residual towers of 12 random blocks grow activations: keep the block outputs small so that 12 of them stay O(1) like a trained tower
    for k in list(W):
        if k.endswith("out_proj.weight") or k.endswith("c_proj.weight"):
            W[k] = W[k] * 0.25
    for j, (pfx, ln) in enumerate([("clipTransEncoder.", "clip_ln"), ("model1.clipTransEncoder_individual.", "model1.clip_ln_individual"),
                                   ("model2.clipTransEncoder_interaction.", "model2.clip_ln_interaction")]):
        for i in range(2):
            for k, v in enc_weights(950 + 50 * j + 10 * i, D, 2048, std=0.03).items():
                W[f"{pfx}layers.{i}.{k}"] = v
        W[ln + ".weight"], W[ln + ".bias"] = 1 + rnd(980 + j, D) * 0.1, rnd(990 + j, D) * 0.1
    g = torch.Generator().manual_seed(19)

    def prompts(B, lens):
        rows = []
        for b in range(B):
            e = lens[b % len(lens)]
            rows.append(torch.cat([torch.tensor([V - 2]), torch.randint(1, V - 2, (e,), generator=g), torch.tensor([V - 1]), torch.zeros(ctx - e - 2, dtype=torch.long)]))
        return torch.stack(rows)
    tok = prompts(3, [5, 30, 74])
    tower = ClipTextTower(W, "", num_heads=H)
    ref = EN.clip_text_tower(W, "", tok, H)
    assert_close(tower(tok), ref, atol=5e-4, rtol=5e-4, what="ViT-L/14-sized text tower")
    # the heads run their 96-wide heads on the MFMA kernel (zero-padded), not on the scalar any-size fallback
    x = dev(rnd(7, 2, ctx, 3 * D))
    a = ops.attention(x[..., :D], x[..., D:2 * D], x[..., 2 * D:], HH, zero_key=False)
    qh, kh, vh = [t.reshape(2, ctx, HH, 96).transpose(1, 2).double().cpu() for t in (x[..., :D], x[..., D:2 * D], x[..., 2 * D:])]
    want = (torch.softmax(qh @ kh.transpose(-1, -2) / math.sqrt(96), -1) @ vh).transpose(1, 2).reshape(2, ctx, D)
    assert_close(a, want.float(), atol=2e-5, rtol=1e-4, what="dh = 96 attention on the padded MFMA kernel")
    enc = MixerTextEncoder(W, clip_heads=H, head_heads=HH)
    t1, t2, tI = prompts(2, [6, 11]), prompts(2, [9, 20]), prompts(2, [17, 40])
    cond = enc.generate_cond({"tokens_text_individual1": t1, "tokens_text_individual2": t2, "tokens_text": tI})
    c = {k: EN.clip_text_tower(W, "", tk, H) for k, tk in [("1", t1), ("2", t2), ("I", tI)]}
    hd = lambda pfx, ln, key, tk: EN.text_head(W, pfx, ln, c[key], tk, HH)
    mi, m1, m2 = ("clipTransEncoder.", "clip_ln"), ("model1.clipTransEncoder_individual.", "model1.clip_ln_individual"), ("model2.clipTransEncoder_interaction.", "model2.clip_ln_interaction")
    want = torch.cat([hd(*m2, "I", tI), hd(*m2, "1", t1), hd(*m2, "2", t2), hd(*m1, "1", t1), hd(*m1, "2", t2),
                      hd(*mi, "I", tI), hd(*mi, "1", t1), hd(*mi, "2", t2)], dim=1)
    assert cond.shape == (2, 8 * D)
    assert_close(cond, want, atol=5e-4, rtol=5e-4, what="generate_cond at ViT-L/14 sizes")
    # timing: the conditioning of a B = 16 batch (3 tower passes + 8 head passes)
    big = {"tokens_text_individual1": prompts(16, [6, 11, 30]), "tokens_text_individual2": prompts(16, [9, 20, 44]), "tokens_text": prompts(16, [17, 40, 70])}
    enc.generate_cond(big)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        enc.generate_cond(big)
    torch.cuda.synchronize()
    print(f"text stage at ViT-L/14 sizes, B = 16: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms per generate_cond (3 tower passes + 8 heads)")
