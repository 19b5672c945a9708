#!/usr/bin/env python3
"""Generate tests/golden/*.npz by importing the REFERENCE (builder container only).

Run:  python tests/golden/make_golden.py            (needs /root/reference; never runs on the GPU box)

What it does:
  * creates three import stubs in a temp dir (``clip``, ``aitviewer.renderables.lines``, ``yacs.config``) --
    none of them touches arithmetic -- and four seeded synthetic normaliser files under a temp cwd;
  * puts ``/root/reference/src`` on sys.path and constructs the reference's own modules directly
    (in2INDenoiser, InterDenoiser, Influence, Mixer, ClassifierFreeSampleModel{,X2}, MixerDiffusion,
    MotionDiffusion, alignment / rotation helpers), bypassing the MixerMDM facade (needs CLIP + checkpoints);
  * re-draws every parameter N(0, std) so zero_module'd layers are non-trivial;
  * stores inputs, weights (reference state_dict key names, minus the 5000-row pe buffers) and the
    reference's outputs as small .npz fixtures.  Fixtures are data only; no reference source is copied.
"""
import os
import sys
import tempfile
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/src"

STUBS = {
    "clip/__init__.py": "def load(*a, **k): raise RuntimeError('clip stub')\n"
                        "def tokenize(*a, **k): raise RuntimeError('clip stub')\n",
    "aitviewer/__init__.py": "",
    "aitviewer/renderables/__init__.py": "",
    "aitviewer/renderables/lines.py": "class Lines: pass\n",
    "yacs/__init__.py": "",
    "yacs/config.py": "class CfgNode(dict):\n    pass\n",
}


def setup():
    tmp = tempfile.mkdtemp(prefix="mmdm_golden_")
    for rel, src in STUBS.items():
        p = os.path.join(tmp, "stubs", rel)
        os.makedirs(os.path.dirname(p), exist_ok=True)
        open(p, "w").write(src)
    cwd = os.path.join(tmp, "cwd")
    os.makedirs(os.path.join(cwd, "data", "HumanML3D"))
    rng = np.random.default_rng(3)
    stats = {}
    for key, rel in [("mean_ih", "data/global_mean.npy"), ("mean_hml", "data/HumanML3D/mean_ih_new.npy")]:
        stats[key] = rng.normal(0, 0.1, 262).astype(np.float32)
        np.save(os.path.join(cwd, rel), stats[key])
    for key, rel in [("std_ih", "data/global_std.npy"), ("std_hml", "data/HumanML3D/std_ih_new.npy")]:
        stats[key] = rng.uniform(0.5, 1.5, 262).astype(np.float32)
        np.save(os.path.join(cwd, rel), stats[key])
    os.chdir(cwd)
    sys.dont_write_bytecode = True
    sys.path[:0] = [os.path.join(tmp, "stubs"), REF]
    return stats


STATS = setup()

import torch  # noqa: E402
from models.in2in import in2INDenoiser  # noqa: E402
from models.intergen import InterDenoiser  # noqa: E402
from models.mixermdm import Mixer  # noqa: E402
from models.utils.influence import Influence, InfluenceBlockCross  # noqa: E402
from models.utils.layers import AdaLN, VanillaSelfAttention, VanillaCrossAttention, FFN  # noqa: E402
from models.utils.blocks import TransformerBlock, TransformerBlockDoubleCond  # noqa: E402
from models.utils.utils import PositionalEncoding  # noqa: E402
from models.utils.cfg_sampler import (ClassifierFreeSampleModel, ClassifierFreeSampleModelX2, ClassifierFreeSampleModelMultiple,  # noqa: E402
                                      ClassifierFreeSampleDualMDM)
from models.mdm import MDMDenoiser  # noqa: E402
from models.utils import gaussian_diffusion as gd  # noqa: E402
from utils import alignment as al  # noqa: E402
from utils import rotation_conversions as rc  # noqa: E402
from utils import quaternion as qt  # noqa: E402

torch.set_grad_enabled(False)


def reinit(mod, seed, std=0.1):
    """Per-parameter seeding (crc32 of the parameter name + seed): modules that share a parameter name and
    shape get identical values, so fixtures can store one weight set for several variants."""
    import zlib
    for name, p in mod.named_parameters():
        g = torch.Generator().manual_seed(zlib.crc32(name.encode()) + seed)
        p.copy_(torch.randn(p.shape, generator=g) * std)
    mod.eval()
    return mod


def sd(mod, prefix=""):
    """state_dict as numpy, without the [5000, D] pe buffers (recomputed; pinned separately by pe.npz)."""
    return {"w:" + prefix + k: v.numpy() for k, v in mod.state_dict().items() if not k.endswith("sequence_pos_encoder.pe")}


def rnd(seed, *shape, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def save(name, **arrs):
    out = {k: (v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arrs.items()}
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.0f} KiB  ({len(out)} arrays)")


# ---- G1 schedule ---------------------------------------------------------------------------------
def g_schedule():
    out = {"betas_cosine_1000": gd.get_named_beta_schedule("cosine", 1000),
           "betas_linear_1000": gd.get_named_beta_schedule("linear", 1000)}
    for strat in ["ddim50", "ddim1000", "ddim20"]:
        d = gd.MixerDiffusion(use_timesteps=gd.space_timesteps(1000, strat), betas=out["betas_cosine_1000"],
                              model_mean_type=gd.ModelMeanType.START_X, model_var_type=gd.ModelVarType.FIXED_SMALL,
                              loss_type=gd.LossType.MSE, rescale_timesteps=False)
        out[strat + ":timestep_map"] = np.array(d.timestep_map)
        for a in ["betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod"]:
            out[strat + ":" + a] = getattr(d, a)
    out["space:[1000]"] = np.array(sorted(gd.space_timesteps(1000, [1000])))
    out["space:10,15,20@300"] = np.array(sorted(gd.space_timesteps(300, "10,15,20")))
    out["space:ddim25@1000"] = np.array(sorted(gd.space_timesteps(1000, "ddim25")))
    save("schedule", **out)


# ---- G2 pe ---------------------------------------------------------------------------------------
def g_pe():
    rows = [0, 1, 299, 999]
    out = {"rows": np.array(rows)}
    for D in [64, 512, 1024]:
        out[f"pe{D}"] = PositionalEncoding(D, dropout=0).pe[rows]
    save("pe", **out)


# ---- G3 layers -----------------------------------------------------------------------------------
def g_layers():
    D, F_, H, T, B = 32, 64, 4, 12, 2
    x, y, e1, e2 = rnd(1, B, T, D), rnd(2, B, T, D), rnd(3, B, D), rnd(4, B, D)
    out = {"x": x, "y": y, "emb": e1, "emb2": e2, "H": H}
    m = reinit(AdaLN(D), 10); out.update(sd(m, "adaln.")); out["adaln:out"] = m(x, e1)
    m = reinit(VanillaSelfAttention(D, H, 0.1), 11); out.update(sd(m, "sa.")); out["sa:out"] = m(x, e1, None)
    m = reinit(VanillaCrossAttention(D, D, H, 0.1, D), 12); out.update(sd(m, "ca.")); out["ca:out"] = m(x, y, e1, None)
    m = reinit(FFN(D, F_, 0.1, D), 13); out.update(sd(m, "ffn.")); out["ffn:out"] = m(x, e1)
    m = reinit(TransformerBlockDoubleCond("individual", latent_dim=D, num_heads=H, ff_size=F_, dropout=0.1), 14)
    out.update(sd(m, "bdc_ind.")); out["bdc_ind:out"] = m(x, None, e1, None, None)
    m = reinit(TransformerBlockDoubleCond("interaction", latent_dim=D, num_heads=H, ff_size=F_, dropout=0.1), 15)
    out.update(sd(m, "bdc_int.")); out["bdc_int:out"] = m(x, y, e1, e2, None)
    m = reinit(TransformerBlock(latent_dim=D, num_heads=H, ff_size=F_, dropout=0.1), 16)
    out.update(sd(m, "blk.")); out["blk:out"] = m(x, y, e1, None)
    m = reinit(InfluenceBlockCross(latent_dim=D, num_heads=H, ff_size=F_), 17)
    out.update(sd(m, "ibc.")); out["ibc:out"] = m(x, y, e1, e2, None)
    save("layers", **out)


# ---- G4 denoisers --------------------------------------------------------------------------------
DEN = dict(latent_dim=16, ff_size=32, num_layers=2, num_heads=2, dropout=0.1)


def g_denoisers():
    B, T = 2, 12
    out = {"H": DEN["num_heads"], "t": np.array([980, 0])}
    t = torch.tensor([980, 0])
    x1, x2 = rnd(20, B, T, 262), rnd(21, B, T, 524)
    c1, c3 = rnd(22, B, 768), rnd(23, B, 768 * 3)
    out.update(x_ind=x1, x_int=x2, cond_ind=c1, cond_int=c3)
    m = reinit(in2INDenoiser(262, mode="individual", **DEN), 30); out.update(sd(m, "ind.")); out["ind:out"] = m(x1, t, cond=c1)
    m = reinit(in2INDenoiser(262, mode="interaction", **DEN), 31); out.update(sd(m, "int.")); out["int:out"] = m(x2, t, cond=c3)
    m = reinit(InterDenoiser(262, **DEN), 32); out.update(sd(m, "ig.")); out["ig:out"] = m(x2, t, cond=c3)
    save("denoisers", **out)


# ---- G5 influence --------------------------------------------------------------------------------
def g_influence():
    D, B, T = 32, 2, 12
    mi, mI, ci, cI = rnd(40, B, T, D), rnd(41, B, T, D), rnd(42, B, D), rnd(43, B, D)
    out = dict(m_i=mi, m_I=mI, cond_i=ci, cond_I=cI, H=4)
    for mode in [1, 2, 3, 4]:
        m = reinit(Influence(D, 2, 4, 64, mode), 50)   # name-seeded: modes differ only in out.{weight,bias} shape
        w = sd(m, f"m{mode}.")
        out.update(w if mode == 4 else {k: v for k, v in w.items() if ".out." in k})
        out[f"m{mode}:out"] = m(mi, mI, ci, cI, None)
    save("influence", **out)


# ---- G6 geometry ---------------------------------------------------------------------------------
def near_valid_motion(seed, B, T):
    """262-d motions whose rot6d block is close to real rotations, plus hand-picked edge cases."""
    m = rnd(seed, B, T, 262)
    aa = rnd(seed + 1, B, T, 21, 3) * 1.2
    aa[0, 0, 0] = 0.0                                   # angle == 0
    aa[0, 0, 1] = torch.tensor([1e-7, 0.0, 0.0])        # below the 1e-6 small-angle switch
    aa[0, 0, 2] = torch.tensor([3.14159, 0.0, 0.0])     # angle ~ pi
    aa[0, 0, 3] = torch.tensor([0.0, 3.1415926, 0.0])
    aa[0, 0, 4] = torch.tensor([2.0, -2.0, 1.0])        # angle = 3 > pi/2 both signs
    r6 = rc.matrix_to_rotation_6d(rc.axis_angle_to_matrix(aa)) + rnd(seed + 2, B, T, 21, 6) * 0.01
    m[:, :, 132:258] = r6.reshape(B, T, 126)
    return m


def g_geometry():
    B, T = 2, 12
    a, b = rnd(60, B, T, 262), rnd(61, B, T, 262)
    c, d = near_valid_motion(62, B, T), near_valid_motion(65, B, T)
    d[1, :, 0:3] = d[1, :1, 0:3]                        # zero root displacement for sample 1 (moved motion)
    c[0, :, 0] = torch.linspace(0, 1, T); c[0, :, 2] = 0.0  # collinear +x trajectory
    d[0, :, 0] = torch.linspace(0, -1, T); d[0, :, 2] = 0.0  # anti-parallel trajectory (qbetween near w=0)
    out = {}
    for nm, m in [("rand_a", a), ("rand_b", b), ("valid_c", c), ("valid_d", d)]:
        out[nm] = m
        s = al.ih_to_smpl(m)
        out[nm + ":ih_to_smpl"] = s
        out[nm + ":smpl_to_ih"] = al.smpl_to_ih(s)
        cm = al.center_motion(s)
        out[nm + ":center"] = cm
        out[nm + ":center_ih"] = al.smpl_to_ih(cm)
    for nm, m1, m2 in [("rand", a, b), ("valid", c, d)]:
        s1, s2 = al.ih_to_smpl(m1), al.ih_to_smpl(m2)
        r1, r2 = al.align_motions(s1, s2, None)
        out[nm + ":align_m1"] = r1
        out[nm + ":align_m2"] = r2
        out[nm + ":align_m2_ih"] = al.smpl_to_ih(r2)
    # helper KATs
    d6 = rnd(70, 64, 6)
    out["kat:d6"] = d6
    mat = rc.rotation_6d_to_matrix(d6)
    out["kat:d6_matrix"] = mat
    q = rc.matrix_to_quaternion(mat)
    out["kat:matrix_quat"] = q
    out["kat:quat_aa"] = rc.quaternion_to_axis_angle(q)
    aa = rnd(71, 64, 3) * 2
    aa[0] = 0.0
    aa[1] = torch.tensor([5e-7, 0, 0])
    out["kat:aa"] = aa
    out["kat:aa_quat"] = rc.axis_angle_to_quaternion(aa)
    out["kat:aa_matrix"] = rc.axis_angle_to_matrix(aa)
    out["kat:aa_d6"] = rc.matrix_to_rotation_6d(rc.axis_angle_to_matrix(aa))
    v0, v1 = rnd(72, 64, 3), rnd(73, 64, 3)
    out["kat:v0"], out["kat:v1"] = v0, v1
    qb = qt.qbetween(v0, v1)
    out["kat:qbetween"] = qb
    out["kat:qrot"] = qt.qrot(qb, v0)
    save("geometry", **out)


# ---- G7-G10 mixer --------------------------------------------------------------------------------
def build_mixer(mode=4, align=True, force=None, model2="in2IN"):
    d1 = in2INDenoiser(262, mode="individual", **DEN)
    d2 = in2INDenoiser(262, mode="interaction", **DEN) if model2 == "in2IN" else InterDenoiser(262, **DEN)
    mix = Mixer(d1, d2, nfeats=262, latent_dim=16, ff_size=32, text_dim=768, n_blocks=2, n_heads=2,
                mixing_mode=mode, store_influence=True, force_influence_val=force, mode="eval", align=align)
    return reinit(mix, 100)


def reset_hist(mix):
    mix.history_influence_i1, mix.history_influence_i2 = [], []
    mix.history_out1, mix.history_out2, mix.history_out_influenced = [], [], []


def make_diffusion(strategy):
    return gd.MixerDiffusion(use_timesteps=gd.space_timesteps(1000, strategy), betas=gd.get_named_beta_schedule("cosine", 1000),
                             model_mean_type=gd.ModelMeanType.START_X, model_var_type=gd.ModelVarType.FIXED_SMALL,
                             loss_type=gd.LossType.MSE, rescale_timesteps=False)


def g_mixer():
    B, T = 2, 8
    out = {k: v for k, v in STATS.items()}
    out.update(d_heads=2, m_heads=2, cfg_scale=3.5)
    B2 = 2 * B
    x1, x2 = rnd(80, B2, T, 524), rnd(81, B2, T, 524)
    cond = rnd(82, B2, 8 * 768)
    cond[B:] = 0
    t = torch.full((B2,), 640, dtype=torch.long)
    out.update(x1=x1, x2=x2, cond=cond, t=t)
    # G7: weights differ per mixing mode only through influence.out -> store the mode-4 set fully and per-mode 'influence.out'
    for mode in [1, 2, 3, 4]:
        for align in [True, False]:
            for force in ([None, 0.0, 1.0] if mode == 4 else [None]):
                mix = build_mixer(mode, align, force)
                if mode == 4 and align and force is None:
                    out.update(sd(mix, "mix."))          # the one full weight set (name-seeded)
                elif mode == 1 and align:
                    out.update({k.replace("w:mix.", "w:mix_out1."): v for k, v in sd(mix, "mix.").items() if "influence.out." in k})
                reset_hist(mix)
                tag = f"fwd:m{mode}:a{int(align)}:f{force}"
                out[tag] = mix(x1, t, cond=cond, mask=None, x2=x2)
                if mode == 4 and force is None:
                    out[tag + ":influence_i1"] = mix.history_influence_i1[0]
                    out[tag + ":influence_i2"] = mix.history_influence_i2[0]
                    out[tag + ":out1"] = mix.history_out1[0]
                    out[tag + ":out2"] = mix.history_out2[0]
    # G8 CFG wrapper + G9 ddim_sample at t>0 and t==0, G10 tiny loop (mode 4, align)
    mix = build_mixer(4, True, None)
    cfg = ClassifierFreeSampleModelX2(mix, 3.5)
    xb, xb2, cb = rnd(83, B, T, 524), rnd(84, B, T, 524), rnd(85, B, 8 * 768)
    out.update(cfg_x=xb, cfg_x2=xb2, cfg_cond=cb)
    reset_hist(mix)
    out["cfg:out"] = cfg(xb, xb2, torch.full((B,), 640, dtype=torch.long), cond=cb, mask=None)
    diff = make_diffusion("ddim50")
    for i in [32, 0]:
        reset_hist(mix)
        r = diff.ddim_sample(cfg, xb, xb2, torch.tensor([i] * B), clip_denoised=False, model_kwargs={"mask": None, "cond": cb})
        for k in ["sample", "sample2", "pred_xstart", "pred_xstart2"]:
            out[f"ddim:i{i}:{k}"] = r[k]
    for strat in ["ddim50", "ddim20"]:
        reset_hist(mix)
        diff = make_diffusion(strat)
        xT = rnd(86, B, T, 524)
        out[f"loop:{strat}:x_T"] = xT
        res = diff.ddim_sample_loop(cfg, (B, T, 524), noise=xT.clone(), clip_denoised=False, progress=False,
                                    model_kwargs={"mask": None, "cond": cb})
        out[f"loop:{strat}:output"] = res
        n = len(mix.history_out1)
        out[f"loop:{strat}:nsteps"] = n
        for name, lst in [("influence_i1", mix.history_influence_i1), ("influence_i2", mix.history_influence_i2),
                          ("out1", mix.history_out1), ("out2", mix.history_out2), ("out_influenced", mix.history_out_influenced)]:
            out[f"loop:{strat}:{name}:sum"] = np.array([float(v.double().sum()) for v in lst])
            out[f"loop:{strat}:{name}:abssum"] = np.array([float(v.double().abs().sum()) for v in lst])
            if strat == "ddim50":
                for k in [0, n - 1]:
                    out[f"loop:{strat}:{name}:{k}"] = lst[k]
    # InterGen as model2, one forward
    mix = build_mixer(4, True, None, model2="InterGen")
    # name-seeded: denoiser2.* keys of InterDenoiser are a subset of the in2IN ones with identical values,
    # EXCEPT nothing extra is needed: TransformerBlock has the same sa/ca/ffn parameter names.
    reset_hist(mix)
    out["fwd:intergen"] = mix(x1, t, cond=cond, mask=None, x2=x2)
    save("mixer", **out)


# ---- G11 single chain ----------------------------------------------------------------------------
def g_single():
    B, T = 2, 12
    m = reinit(in2INDenoiser(262, mode="individual", **DEN), 30)
    cfg = ClassifierFreeSampleModel(m, 3.5)
    out = dict(sd(m, "ind."))
    out.update(H=DEN["num_heads"], cfg_scale=3.5)
    xT, c = rnd(90, B, T, 262), rnd(91, B, 768)
    out.update(x_T=xT, cond=c)
    out["cfg:out"] = cfg(xT, torch.full((B,), 500, dtype=torch.long), cond=c, mask=None)
    for strat in ["ddim50", "ddim20"]:
        diff = gd.MotionDiffusion(use_timesteps=gd.space_timesteps(1000, strat), motion_rep="global", mode="individual",
                                  betas=gd.get_named_beta_schedule("cosine", 1000),
                                  model_mean_type=gd.ModelMeanType.START_X, model_var_type=gd.ModelVarType.FIXED_SMALL,
                                  loss_type=gd.LossType.MSE, rescale_timesteps=False)
        out[f"loop:{strat}:output"] = diff.ddim_sample_loop(cfg, (B, T, 262), noise=xT.clone(), clip_denoised=False, progress=False,
                                                            model_kwargs={"mask": None, "cond": c})
    save("single", **out)


# ---- G12 stand-alone interaction sampler (4-way CFG) -----------------------------------------------------
def g_interaction():
    B, T = 2, 12
    m = reinit(in2INDenoiser(262, mode="interaction", **DEN), 31)
    cfg = ClassifierFreeSampleModelMultiple(m, 3, 3, 1)          # configs/models/in2IN.yaml CFG_WEIGHT{,_INTERACTION,_INDIVIDUAL}
    out = dict(sd(m, "int."))
    out.update(H=DEN["num_heads"], s=3.0, s_int=3.0, s_ind=1.0)
    xT, c = rnd(95, B, T, 524), rnd(96, B, 3 * 768)
    out.update(x_T=xT, cond=c)
    out["cfg:out"] = cfg(xT, torch.full((B,), 500, dtype=torch.long), cond=c, mask=None)
    diff = gd.MotionDiffusion(use_timesteps=gd.space_timesteps(1000, "ddim20"), motion_rep="global", mode="interaction",
                              betas=gd.get_named_beta_schedule("cosine", 1000),
                              model_mean_type=gd.ModelMeanType.START_X, model_var_type=gd.ModelVarType.FIXED_SMALL,
                              loss_type=gd.LossType.MSE, rescale_timesteps=False)
    out["loop:ddim20:output"] = diff.ddim_sample_loop(cfg, (B, T, 524), noise=xT.clone(), clip_denoised=False, progress=False,
                                                      model_kwargs={"mask": None, "cond": c})
    save("interaction", **out)


# ---- G13 in2IN "dual" sampler (ClassifierFreeSampleDualMDM) ---------------------------------
def g_dual():
    B, T = 2, 12
    m_ind = reinit(in2INDenoiser(262, mode="dual_individual", **DEN), 30)
    m_int = reinit(in2INDenoiser(262, mode="dual_interaction", **DEN), 31)
    out = dict(sd(m_ind, "ind."))
    out.update(sd(m_int, "int."))
    out.update(H=DEN["num_heads"], s_ind=2.0, s_int=3.0)
    xT, c = rnd(97, B, T, 524), rnd(98, B, 5 * 768)
    out.update(x_T=xT, cond=c)
    out["fwd:dual_individual"] = m_ind(xT, torch.tensor([500, 20]), cond=c)
    for func, value in [("exp", 0.004), ("lin", 0.0), ("const", 0.3), ("exp-inv", 0.002)]:
        cfg = ClassifierFreeSampleDualMDM(m_ind, m_int, 2.0, 3.0, func, value)
        out[f"cfg:{func}:out"] = cfg(xT, torch.full((B,), 500, dtype=torch.long), cond=c, mask=None)
        out[f"cfg:{func}:value"] = value
    for func, value in [("exp", 0.004), ("lin", 0.0)]:
        cfg = ClassifierFreeSampleDualMDM(m_ind, m_int, 2.0, 3.0, func, value)
        diff = gd.MotionDiffusion(use_timesteps=gd.space_timesteps(1000, "ddim20"), motion_rep="global", mode="dual",
                                  betas=gd.get_named_beta_schedule("cosine", 1000),
                                  model_mean_type=gd.ModelMeanType.START_X, model_var_type=gd.ModelVarType.FIXED_SMALL,
                                  loss_type=gd.LossType.MSE, rescale_timesteps=False)
        out[f"loop:{func}:ddim20:output"] = diff.ddim_sample_loop(cfg, (B, T, 524), noise=xT.clone(), clip_denoised=False, progress=False,
                                                                  model_kwargs={"mask": None, "cond": c})
    save("dual", **out)


# ---- G14 MDMDenoiser as MODEL1 -------------------------------------------------------------
def g_mdm():
    B, T = 2, 12
    D = DEN["latent_dim"]
    mk = lambda: MDMDenoiser(None, 262, D, DEN["ff_size"], DEN["num_layers"], DEN["num_heads"], DEN["dropout"], "gelu")
    m = reinit(mk(), 33)
    out = dict(sd(m, "mdm."))
    out.update(H=DEN["num_heads"], t=np.array([980, 0]))
    x, c = rnd(24, B, T, 262), rnd(25, B, D)
    out.update(x=x, cond=c)
    out["mdm:out"] = m(x, torch.tensor([980, 0]), cond=c.clone())            # forward adds the timestep embedding to cond IN PLACE
    # Mixer with MDM as denoiser1.  MDMDenoiser hard-codes text_dim = 256 (mdm.py:238) while cond must be latent-sized
    # (mdm.py:279): the fixture uses the tiny latent and sets the attribute accordingly.
    ref_mix = build_mixer(4, True, None)
    d1 = mk()
    d1.text_dim = D
    d2 = in2INDenoiser(262, mode="interaction", **DEN)
    mix = reinit(Mixer(d1, d2, nfeats=262, latent_dim=16, ff_size=32, text_dim=768, n_blocks=2, n_heads=2,
                       mixing_mode=4, store_influence=True, force_influence_val=None, mode="eval", align=True), 100)
    a, b = sd(ref_mix, "mix."), sd(mix, "mix.")
    for k, v in b.items():
        if ".denoiser1." in k:
            out[k.replace("w:mix.", "w:mixmdm.")] = v
        else:
            assert np.array_equal(v, a[k]), k          # everything else is the mixer.npz weight set (name-seeded)
    B2 = 2 * B
    x1, x2 = rnd(80, B2, T, 524), rnd(81, B2, T, 524)
    cond = rnd(87, B2, 6 * 768 + 2 * D)
    cond[B:] = 0
    t = torch.full((B2,), 640, dtype=torch.long)
    out.update(mix_x1=x1, mix_x2=x2, mix_cond=cond.clone(), mix_t=t)
    reset_hist(mix)
    out["fwd:mixmdm"] = mix(x1, t, cond=cond, mask=None, x2=x2)
    out["fwd:mixmdm:out1"] = mix.history_out1[0]
    cfg = ClassifierFreeSampleModelX2(mix, 3.5)
    xb, cb = rnd(88, B, T, 524), rnd(89, B, 6 * 768 + 2 * D)
    out.update(loop_x_T=xb, loop_cond=cb)
    reset_hist(mix)
    out["loop:ddim20:output"] = make_diffusion("ddim20").ddim_sample_loop(cfg, (B, T, 524), noise=xb.clone(), clip_denoised=False,
                                                                         progress=False, model_kwargs={"mask": None, "cond": cb})
    save("mdm", **out)


# ---- G15 text-conditioning stage -----------------------------------------------------------
def g_text():
    """text_process of MixerMDM / in2IN / InterGen, called UNBOUND on a stand-in ``self`` (their __init__ needs the CLIP
    package and its weights).  The CLIP residual tower is third-party and absent, so ``clip_transformer`` is the identity
    here: the fixture pins everything around it -- token + positional embedding, ln_final, the 2-layer post-norm
    clipTransEncoder, clip_ln and the EOT (argmax) gather."""
    import types
    import torch.nn as nn
    import clip as clip_stub
    from models.mixermdm import MixerMDM
    from models.in2in import in2IN
    from models.intergen import InterGen
    Dt, Hh, Ff, V, ctx, B = 32, 4, 64, 50, 10, 3
    tokens = torch.tensor([[49, 3, 7, 48, 0, 0, 0, 0, 0, 0], [49, 5, 5, 5, 5, 5, 5, 5, 5, 48], [49, 48, 0, 0, 0, 0, 0, 0, 0, 0]])
    clip_stub.tokenize = lambda raw, truncate=True: tokens
    def enc():
        return nn.TransformerEncoder(nn.TransformerEncoderLayer(d_model=Dt, nhead=Hh, dim_feedforward=Ff, dropout=0.1, activation="gelu",
                                                                batch_first=True), num_layers=2)
    fake = nn.Module()
    fake.token_embedding = nn.Embedding(V, Dt)
    fake.positional_embedding = nn.Parameter(torch.zeros(ctx, Dt))
    fake.clip_transformer = nn.Linear(1, 1)        # only its device is read (text_process line 1); replaced by the identity below
    fake.ln_final = nn.LayerNorm(Dt)
    fake.clipTransEncoder, fake.clip_ln = enc(), nn.LayerNorm(Dt)
    fake.clipTransEncoder_individual, fake.clip_ln_individual = enc(), nn.LayerNorm(Dt)
    fake.clipTransEncoder_interaction, fake.clip_ln_interaction = enc(), nn.LayerNorm(Dt)
    reinit(fake, 110)
    fake.dtype = torch.float32
    out = {k: v for k, v in sd(fake, "txt.").items() if "clip_transformer" not in k}
    out.update(tokens=tokens, H=Hh)
    class Tower(nn.Module):
        def __init__(self):
            super().__init__()
            self.p = nn.Parameter(torch.zeros(1))
        def forward(self, x):
            return x
    fake.clip_transformer = Tower()
    batch = {"text": ["a", "b", "c"]}
    MixerMDM.text_process(fake, batch, "text", "cond")
    out["mixer:cond"] = batch["cond"]
    for mode in ["individual", "interaction"]:
        in2IN.text_process(fake, batch, mode, "text", "cond_" + mode)
        out[f"in2in:{mode}:cond"] = batch["cond_" + mode]
    InterGen.text_process(fake, batch, "interaction", "text", "cond_ig")
    out["intergen:cond"] = batch["cond_ig"]
    save("text", **out)


# ---- G16 mixer at dimensions the bf16-matrix-core GEMM modes accept (K % 32 == 0) ---------------------------------------
def g_mixer32():
    """Same objects as g_mixer with latent 32 / ff 64: the smallest sizes the fp32-split (and bf16) GEMM kernels take, so that those modes
    are pinned against the reference itself and not only against the oracle."""
    den = dict(latent_dim=32, ff_size=64, num_layers=2, num_heads=2, dropout=0.1)
    d1 = in2INDenoiser(262, mode="individual", **den)
    d2 = in2INDenoiser(262, mode="interaction", **den)
    mix = reinit(Mixer(d1, d2, nfeats=262, latent_dim=32, ff_size=64, text_dim=768, n_blocks=2, n_heads=2, mixing_mode=4, store_influence=True,
                       force_influence_val=None, mode="eval", align=True), 200)
    B, T = 2, 8
    out = {k: v for k, v in STATS.items()}
    out.update(sd(mix, "mix."))
    B2 = 2 * B
    x1, x2, cond = rnd(280, B2, T, 524), rnd(281, B2, T, 524), rnd(282, B2, 8 * 768)
    cond[B:] = 0
    t = torch.full((B2,), 640, dtype=torch.long)
    out.update(x1=x1, x2=x2, cond=cond, t=t)
    reset_hist(mix)
    out["fwd"] = mix(x1, t, cond=cond, mask=None, x2=x2)
    cfg = ClassifierFreeSampleModelX2(mix, 3.5)
    xT, cb = rnd(286, B, T, 524), rnd(285, B, 8 * 768)
    out.update(x_T=xT, cfg_cond=cb)
    reset_hist(mix)
    r = make_diffusion("ddim20").ddim_sample(cfg, xT, xT.clone(), torch.tensor([12] * B), clip_denoised=False, model_kwargs={"mask": None, "cond": cb})
    for k in ["sample", "sample2", "pred_xstart", "pred_xstart2"]:
        out[f"ddim:i12:{k}"] = r[k]
    reset_hist(mix)
    out["loop:ddim20:output"] = make_diffusion("ddim20").ddim_sample_loop(cfg, (B, T, 524), noise=xT.clone(), clip_denoised=False, progress=False,
                                                                         model_kwargs={"mask": None, "cond": cb})
    save("mixer32", **out)


# ---- full model dimensions (configs/models/*.yaml), weights from the repo's seeded generator ------------------------------
def g_fulldims():
    """The REFERENCE at the real model sizes (D=1024, F=2048, L=8, H=8; mixer 512/1024/4/8), with the weights of
    mixermdm_amd.synthetic.synthetic_state_dict(seed=0, std=0.02, bias_std=0.02) loaded into the reference modules by name.  The fixture
    holds NO weights and no inputs -- only the seeds they are drawn from (same deterministic CPU generators on every box), the
    normaliser statistics and the reference's outputs: Mixer.forward and MixerDiffusion.ddim_sample at B=2, T=32 and one ddim1000
    step at the headline length T=300 (B=1)."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from mixermdm_amd.synthetic import synthetic_state_dict, synthetic_inputs, FULL_DIMS
    den = dict(latent_dim=1024, ff_size=2048, num_layers=8, num_heads=8, dropout=0.1)
    d1 = in2INDenoiser(262, mode="individual", **den)
    d2 = in2INDenoiser(262, mode="interaction", **den)
    mix = Mixer(d1, d2, nfeats=262, latent_dim=512, ff_size=1024, text_dim=768, n_blocks=4, n_heads=8, mixing_mode=4, store_influence=True,
                force_influence_val=None, mode="eval", align=True)
    wsd = synthetic_state_dict(seed=0, std=0.02, bias_std=0.02, **FULL_DIMS)
    res = mix.load_state_dict(wsd, strict=False)
    assert not res.unexpected_keys and all(k.endswith("sequence_pos_encoder.pe") for k in res.missing_keys), res
    mix.eval()
    out = {k: v for k, v in STATS.items()}
    out.update(weights_seed=0, weights_std=0.02, weights_bias_std=0.02, cfg_scale=3.5, d_heads=8, m_heads=8)
    # Mixer.forward on a CFG-doubled batch, inputs rnd(seed, ...) as in the other fixtures
    B, T = 2, 32
    B2 = 2 * B
    seeds = dict(x1=380, x2=381, cond=382)
    x1, x2, cond = rnd(seeds["x1"], B2, T, 524), rnd(seeds["x2"], B2, T, 524), rnd(seeds["cond"], B2, 8 * 768)
    cond[B:] = 0
    tt = 640
    out.update(fwd_seeds=np.array([seeds["x1"], seeds["x2"], seeds["cond"]]), fwd_shape=np.array([B2, T]), fwd_t=tt)
    reset_hist(mix)
    out["fwd"] = mix(x1, torch.full((B2,), tt, dtype=torch.long), cond=cond, mask=None, x2=x2)
    out["fwd:influence_i1"] = mix.history_influence_i1[0][..., [0, 3, 66, 132, 258]]      # one channel of five of the 23 groups
    cfg = ClassifierFreeSampleModelX2(mix, 3.5)
    # ddim_sample (ddim50, i = 32 and the un-normalised i = 0 branch) from synthetic_inputs(B, T) = the bench's input generator
    cb, xT = synthetic_inputs(B, T)
    xb2 = rnd(383, B, T, 524)
    out.update(step_B=B, step_T=T, step_x2_seed=383)
    diff = make_diffusion("ddim50")
    for i in [32, 0]:
        reset_hist(mix)
        r = diff.ddim_sample(cfg, xT, xb2, torch.tensor([i] * B), clip_denoised=False, model_kwargs={"mask": None, "cond": cb})
        for k in ["sample", "sample2", "pred_xstart2"]:
            out[f"ddim50:i{i}:{k}"] = r[k]
    # the headline length: one ddim1000 step at T = 300, B = 1, both chains at x_T (the first step of the benchmarked loop) ...
    cb, xT = synthetic_inputs(1, 300)
    diff = make_diffusion("ddim1000")
    reset_hist(mix)
    r = diff.ddim_sample(cfg, xT, xT.clone(), torch.tensor([999]), clip_denoised=False, model_kwargs={"mask": None, "cond": cb})
    for k in ["sample", "sample2"]:
        out[f"ddim1000:T300:i999:{k}"] = r[k]
    # ... and a late one (i = 3: alpha_bar ~ 1) from chains that differ
    xa, xb_ = rnd(384, 1, 300, 524), rnd(385, 1, 300, 524)
    out.update(late_seeds=np.array([384, 385]))
    reset_hist(mix)
    r = diff.ddim_sample(cfg, xa, xb_, torch.tensor([3]), clip_denoised=False, model_kwargs={"mask": None, "cond": cb})
    for k in ["sample", "sample2"]:
        out[f"ddim1000:T300:i3:{k}"] = r[k]
    save("fulldims", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["schedule", "pe", "layers", "denoisers", "influence", "geometry", "mixer", "single", "interaction", "dual", "mdm", "text", "mixer32",
                             "fulldims"]
    for w in which:
        globals()["g_" + w]()
