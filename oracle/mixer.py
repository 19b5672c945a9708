"""Mixer step + two-chain DDIM loop, CPU restatement (oracle; test infrastructure only).

Reference: /root/reference/src/models/mixermdm.py:660-810 (Mixer.forward),
/root/reference/src/models/utils/cfg_sampler.py:38-56 (ClassifierFreeSampleModelX2),
/root/reference/src/models/utils/gaussian_diffusion.py:1769-2091 (MixerDiffusion sampler),
:799-1069 + cfg_sampler.py:5-28 (single-chain variant used by configs 1-2).
"""
import torch
from . import layers as L
from . import geometry as G
from .denoiser import in2in_denoiser, inter_denoiser, influence
from .encoder import mdm_denoiser


class MixerSpec:
    """Static description of a Mixer instance (what Mixer.__init__ stores -- mixermdm.py:606-657)."""

    def __init__(self, d_heads=8, m_heads=8, mixing_mode=4, align=True, force_influence_val=None,
                 nfeats=262, text_dim=768, d1_text_dim=768, d2_text_dim=768, model2="in2IN", model1="in2INind"):
        self.d_heads, self.m_heads = d_heads, m_heads
        self.mixing_mode, self.align, self.force = mixing_mode, align, force_influence_val
        self.nfeats, self.text_dim = nfeats, text_dim
        self.d1_text_dim, self.d2_text_dim = d1_text_dim, d2_text_dim
        self.model2 = model2
        self.model1 = model1      # "in2INind" | "MDM" (mixermdm.py:32-40, 264-272); MDM: d1_text_dim = its latent size (mdm.py:279)


def expand_influence(w, mode, T):
    """mixermdm.py:739-786: 1/23 weights -> 262 channels."""
    if mode == 1:
        return w.unsqueeze(1).expand(-1, T, -1)
    if mode == 2:
        return w
    if mode == 3:
        w = w.unsqueeze(1).expand(-1, T, -1)
    elif mode != 4:
        raise ValueError("Mixing mode not recognized")
    j = w[..., :22].repeat_interleave(3, dim=-1)
    r = w[..., :21].repeat_interleave(6, dim=-1)
    f = w[..., 22:].expand(-1, -1, 4)
    return torch.cat([j, j, r, f], dim=-1)


def mixer_forward(W, spec, stats, x1, timesteps, cond, x2, hist=None):
    """Mixer.forward (mode "eval"/"eval_intermediate", mask=None) -- mixermdm.py:660-810.

    stats = (mean_hml, std_hml, mean_ih, std_ih), each [262] (utils/utils.py:44-82).
    cond layout [interaction, int_ind1, int_ind2, ind_ind1, ind_ind2, infl_I, infl_i1, infl_i2] (mixermdm.py:342-354).
    hist: optional dict of lists mirroring history_* (mixermdm.py:794-796, 805-808).
    """
    nf, td1, td2, td = spec.nfeats, spec.d1_text_dim, spec.d2_text_dim, spec.text_dim
    mean_h, std_h, mean_i, std_i = stats
    B, T = x1.shape[:2]
    x1 = G._f(x1)           # the reference's .float() (mixermdm.py:663-664); float64 inputs stay float64 (tests/parity_tol.py yardstick)
    x2 = G._f(x2)
    cond1_1 = cond[:, td2 * 3:td2 * 3 + td1]
    cond1_2 = cond[:, td2 * 3 + td1:td2 * 3 + td1 * 2]
    cond2 = cond[:, :td * 3]
    base = td2 * 3 + td1 * 2
    pe = W["sequence_pos_encoder.pe"]
    te = L.timestep_embed(W, "embed_timestep", pe, timesteps)
    cond_I = te + L.linear(W, "text_embed", cond[:, base:base + td2])
    cond_i1 = te + L.linear(W, "text_embed", cond[:, base + td2:base + 2 * td2])
    cond_i2 = te + L.linear(W, "text_embed", cond[:, base + 2 * td2:base + 3 * td2])

    if spec.model1 == "MDM":
        o11 = mdm_denoiser(W, "denoiser1.", x1[:, :, :nf], timesteps, cond1_1, spec.d_heads)
        o12 = mdm_denoiser(W, "denoiser1.", x1[:, :, nf:], timesteps, cond1_2, spec.d_heads)
    else:
        o11 = in2in_denoiser(W, "denoiser1.", "individual", x1[:, :, :nf], timesteps, cond1_1, spec.d_heads, nf)
        o12 = in2in_denoiser(W, "denoiser1.", "individual", x1[:, :, nf:], timesteps, cond1_2, spec.d_heads, nf)
    if spec.model2 == "InterGen":
        o2 = inter_denoiser(W, "denoiser2.", x2, timesteps, cond2, spec.d_heads, nf)
    else:
        o2 = in2in_denoiser(W, "denoiser2.", "interaction", x2, timesteps, cond2, spec.d_heads, nf)

    o11 = o11 * std_h + mean_h
    o12 = o12 * std_h + mean_h
    o2 = (o2.reshape(B, T, 2, -1) * std_i + mean_i).reshape(B, T, -1)
    o21, o22 = o2[..., :nf], o2[..., nf:]
    if spec.align:
        s11, s12, s21, s22 = G.ih_to_smpl(o11), G.ih_to_smpl(o12), G.ih_to_smpl(o21), G.ih_to_smpl(o22)
        dg = hist.setdefault("align_diag", []) if hist is not None else None      # conditioning of the two alignments (person 1, person 2)
        s11 = G.align_motions(s21, s11, dg)
        s12 = G.align_motions(s22, s12, dg)
        o11, o12, o21, o22 = G.smpl_to_ih(s11), G.smpl_to_ih(s12), G.smpl_to_ih(s21), G.smpl_to_ih(s22)
    out1 = torch.cat([o11, o12], dim=-1)
    out2 = torch.cat([o21, o22], dim=-1)

    emb = lambda m: L.linear(W, "motion_embed", m) + pe[:T].unsqueeze(0)
    i1 = influence(W, "influence.", spec.mixing_mode, emb(o11), emb(o21), cond_i1, cond_I, spec.m_heads)
    i2 = influence(W, "influence.", spec.mixing_mode, emb(o12), emb(o22), cond_i2, cond_I, spec.m_heads)
    i1 = expand_influence(i1, spec.mixing_mode, T)
    i2 = expand_influence(i2, spec.mixing_mode, T)
    if spec.force is not None:
        i1 = torch.ones_like(i1) * spec.force
        i2 = torch.ones_like(i2) * spec.force
    mixed = torch.cat([o21 + i1 * (o11 - o21), o22 + i2 * (o12 - o22)], dim=-1)
    if hist is not None:
        hist.setdefault("influence_i1", []).append(i1)
        hist.setdefault("influence_i2", []).append(i2)
        hist.setdefault("out1", []).append(out1)
        hist.setdefault("out2", []).append(out2)
        hist.setdefault("out_influenced", []).append(mixed)
    return mixed


def cfg_x2(W, spec, stats, s, x, x2, timesteps, cond, hist=None):
    """ClassifierFreeSampleModelX2.forward -- cfg_sampler.py:38-56."""
    B = x.shape[0]
    out = mixer_forward(W, spec, stats, torch.cat([x, x]), torch.cat([timesteps, timesteps]),
                        torch.cat([cond, torch.zeros_like(cond)]), torch.cat([x2, x2]), hist)
    return s * out[:B] + (1 - s) * out[B:]


def process_xstart(x, stats, t0_positive, align=True, diag=None):
    """MixerDiffusion.p_mean_variance.process_xstart -- gaussian_diffusion.py:2031-2062 (clip_denoised=False).

    When t[0]==0 BOTH returns are the raw x: x1 is only rebuilt inside the ``if t[0] > 0`` branch (:2052-2056).
    """
    mean_h, std_h, mean_i, std_i = stats
    B, T = x.shape[:2]
    if not t0_positive:
        return x.clone(), x.clone()
    x11, x12 = x[..., :262], x[..., 262:]
    if align:
        x11 = G.smpl_to_ih(G.center_motion(G.ih_to_smpl(x11), diag))
        x12 = G.smpl_to_ih(G.center_motion(G.ih_to_smpl(x12), diag))
    x1 = torch.cat([(x11 - mean_h) / std_h, (x12 - mean_h) / std_h], dim=-1)
    x2 = ((x.reshape(B, T, 2, -1) - mean_i) / std_i).reshape(B, T, -1)
    return x1, x2


def ddim_update(sched, i, x, x0):
    """_predict_eps_from_xstart (:558-562) + DDIM eta=0 mean (:1936-1965); tables cast to fp32 at gather (:1264-1277)."""
    # the coefficient VALUES are the fp32-rounded table entries in every case; in a float64 run they are widened again, so that run differs
    # from the fp32 one by arithmetic rounding only
    f = lambda arr: torch.tensor(arr[i], dtype=torch.float64).float().to(x.dtype)
    eps = (f(sched.sqrt_recip_alphas_cumprod) * x - x0) / f(sched.sqrt_recipm1_alphas_cumprod)
    ab_prev = f(sched.alphas_cumprod_prev)
    return x0 * torch.sqrt(ab_prev) + torch.sqrt(1 - ab_prev - 0.0 ** 2) * eps


def mixer_ddim_step(W, spec, stats, sched, s, i, x, x2, cond, hist=None, xstart_align=True):
    """One MixerDiffusion.ddim_sample at respaced index i -- gaussian_diffusion.py:1902-1965.

    Returns (sample, sample2, pred_xstart, pred_xstart2).  The model sees timestep_map[i] (:2200-2205).
    """
    B = x.shape[0]
    ts = torch.full((B,), sched.timestep_map[i], dtype=torch.long)
    out = cfg_x2(W, spec, stats, s, x, x2, ts, cond, hist)
    p1, p2 = process_xstart(out, stats, i > 0, xstart_align, hist.setdefault("center_diag", []) if hist is not None else None)
    return ddim_update(sched, i, x, p1), ddim_update(sched, i, x2, p2), p1, p2


def mixer_ddim_loop(W, spec, stats, sched, s, x_T, cond, hist=None, first_steps=None):
    """MixerDiffusion.ddim_sample_loop -- gaussian_diffusion.py:1769-1899: both chains start from x_T (:1863);
    returns the last step's pred_xstart2 (:1820).  ``first_steps`` truncates for bounded-time tests."""
    x, x2 = x_T.clone(), x_T.clone()
    p2 = None
    idx = list(range(sched.num_timesteps))[::-1]
    if first_steps is not None:
        idx = idx[:first_steps]
    for i in idx:
        x, x2, _, p2 = mixer_ddim_step(W, spec, stats, sched, s, i, x, x2, cond, hist)
    return p2, x, x2


# ---- single-chain variant (BASELINE configs 1-2) -------------------------------------------------

def cfg_single(W, p, mode, s, x, timesteps, cond, num_heads):
    """ClassifierFreeSampleModel.forward -- cfg_sampler.py:12-28."""
    B = x.shape[0]
    out = in2in_denoiser(W, p, mode, torch.cat([x, x]), torch.cat([timesteps, timesteps]),
                         torch.cat([cond, torch.zeros_like(cond)]), num_heads)
    return s * out[:B] + (1 - s) * out[B:]


def single_ddim_loop(W, p, mode, sched, s, x_T, cond, num_heads, first_steps=None):
    """GaussianDiffusion.ddim_sample_loop (eta=0, START_X, clip_denoised=False) -- gaussian_diffusion.py:799-849, 946-1069;
    returns the final pred_xstart, in normalised space (in2in.py:343-353)."""
    x = x_T.clone()
    B = x.shape[0]
    x0 = None
    idx = list(range(sched.num_timesteps))[::-1]
    if first_steps is not None:
        idx = idx[:first_steps]
    for i in idx:
        ts = torch.full((B,), sched.timestep_map[i], dtype=torch.long)
        x0 = cfg_single(W, p, mode, s, x, ts, cond, num_heads)
        x = ddim_update(sched, i, x, x0)
    return x0, x


def cfg_multiple(W, p, s, s_int, s_ind, x, timesteps, cond, num_heads, model2="in2IN"):
    """ClassifierFreeSampleModelMultiple.forward -- cfg_sampler.py:67-98 (4 copies: full | interaction-only | individuals-only | zeros)."""
    B = x.shape[0]
    c_int = torch.zeros_like(cond)
    c_int[:, :768] = cond[:, :768]
    c_ind = torch.zeros_like(cond)
    c_ind[:, 768:] = cond[:, 768:]
    cc = torch.cat([cond, c_int, c_ind, torch.zeros_like(cond)], dim=0)
    xx, tt = torch.cat([x] * 4), torch.cat([timesteps] * 4)
    out = inter_denoiser(W, p, xx, tt, cc, num_heads) if model2 == "InterGen" else in2in_denoiser(W, p, "interaction", xx, tt, cc, num_heads)
    return (s * out[:B]) + (s_int * out[B:2 * B]) + (s_ind * out[2 * B:3 * B]) + ((1 - (s + s_int + s_ind)) * out[3 * B:])


def interaction_ddim_loop(W, p, sched, s, s_int, s_ind, x_T, cond, num_heads, first_steps=None):
    """in2INDiffusion.forward, mode "interaction" (in2in.py:330-341) on MotionDiffusion's single-chain DDIM loop."""
    x = x_T.clone()
    B = x.shape[0]
    x0 = None
    idx = list(range(sched.num_timesteps))[::-1]
    if first_steps is not None:
        idx = idx[:first_steps]
    for i in idx:
        ts = torch.full((B,), sched.timestep_map[i], dtype=torch.long)
        x0 = cfg_multiple(W, p, s, s_int, s_ind, x, ts, cond, num_heads)
        x = ddim_update(sched, i, x, x0)
    return x0, x


# ---- in2IN "dual" sampler (DualMDM composition) -------------------------------------------------

def dual_weight(func, value, t):
    """ClassifierFreeSampleDualMDM.weight -- cfg_sampler.py:113-125, evaluated on the (remapped) timestep t in float64."""
    import numpy as np
    x = np.asarray([t])
    if func == "exp":
        return np.exp(-value * (1000 - x))[0]
    if func == "exp-inv":
        return 1 - np.exp(-value * (1000 - x))[0]
    if func == "lin":
        return 1 - ((1000 - x) / 1000)[0]
    if func == "const":
        return value
    raise ValueError("Unknown function")


def cfg_dual(W, p_ind, p_int, s_ind, s_int, w, x, timesteps, cond, num_heads):
    """ClassifierFreeSampleDualMDM.forward -- cfg_sampler.py:127-150: both denoisers on the CFG-doubled batch,
    per-model guidance, then out_int + w (out_ind - out_int) with the time-scheduled scalar w."""
    B = x.shape[0]
    xx, tt = torch.cat([x, x]), torch.cat([timesteps, timesteps])
    cc = torch.cat([cond, torch.zeros_like(cond)])
    o_int = in2in_denoiser(W, p_int, "dual_interaction", xx, tt, cc, num_heads)
    o_ind = in2in_denoiser(W, p_ind, "dual_individual", xx, tt, cc, num_heads)
    g_int = o_int[B:] + s_int * (o_int[:B] - o_int[B:])
    g_ind = o_ind[B:] + s_ind * (o_ind[:B] - o_ind[B:])
    return g_int + float(w) * (g_ind - g_int)


def dual_ddim_loop(W, p_ind, p_int, sched, s_ind, s_int, func, value, x_T, cond, num_heads, first_steps=None):
    """in2INDiffusion.forward, mode "dual" (in2in.py:318-329) on MotionDiffusion's single-chain DDIM loop."""
    x = x_T.clone()
    B = x.shape[0]
    x0 = None
    idx = list(range(sched.num_timesteps))[::-1]
    if first_steps is not None:
        idx = idx[:first_steps]
    for i in idx:
        t = sched.timestep_map[i]
        ts = torch.full((B,), t, dtype=torch.long)
        x0 = cfg_dual(W, p_ind, p_int, s_ind, s_int, dual_weight(func, value, t), x, ts, cond, num_heads)
        x = ddim_update(sched, i, x, x0)
    return x0, x
