"""Denoiser / Influence CPU restatement (oracle; test infrastructure only).

Reference: /root/reference/src/models/in2in.py:401-462 (in2INDenoiser.forward),
/root/reference/src/models/intergen.py:260-288 (InterDenoiser.forward),
/root/reference/src/models/utils/influence.py:93-126 (Influence.forward).
Inference path only: mask=None (all keys valid), dropout off.
"""
import torch
from . import layers as L


def in2in_denoiser(W, p, mode, x, timesteps, cond, num_heads, nfeats=262):
    """in2INDenoiser.forward, modes "individual" / "interaction" / "dual_interaction" / "dual_individual" -- in2in.py:401-462.

    W[p+"sequence_pos_encoder.pe"] is the [5000, D] table buffer (utils.py:24-35);
    the timestep embedding indexes the SAME table with the remapped t (utils.py:54-55).
    Interaction CA uses the other stream's previous-layer h for both updates (in2in.py:439-440).
    """
    if mode not in ("individual", "interaction", "dual_interaction", "dual_individual"):
        raise ValueError("Mode not recognized")
    pe = W[p + "sequence_pos_encoder.pe"]
    T = x.shape[1]
    te = L.timestep_embed(W, p + "embed_timestep", pe, timesteps)
    txt = lambda c: L.linear(W, p + "text_embed", c)
    num_layers = 1 + max(int(k[len(p + "blocks."):].split(".")[0]) for k in W if k.startswith(p + "blocks."))
    x_a = x[..., :nfeats]
    h_a = L.linear(W, p + "motion_embed", x_a) + pe[:T].unsqueeze(0)
    if mode == "individual":
        emb1 = te + txt(cond[:, :768])
        for i in range(num_layers):
            h_a = L.block_double_cond(W, f"{p}blocks.{i}", mode, h_a, None, emb1, None, num_heads)
        return L.linear(W, p + "out.linear", h_a)
    x_b = x[..., nfeats:]
    h_b = L.linear(W, p + "motion_embed", x_b) + pe[:T].unsqueeze(0)
    if mode == "dual_individual":
        # two persons, each through the individual blocks with its own text (cond columns 3*768.. and 4*768..) -- in2in.py:420-422, 441-443
        emb1 = te + txt(cond[:, 768 * 3:768 * 4])
        emb2 = te + txt(cond[:, 768 * 4:])
        # QUIRK (in2in.py:448-451): h_b_prev is only advanced in the interaction modes, so in "dual_individual" every
        # layer recomputes person b from the EMBEDDED input and the result is the LAST block applied once.
        for i in range(num_layers):
            h_a = L.block_double_cond(W, f"{p}blocks.{i}", mode, h_a, None, emb1, None, num_heads)
        h_b = L.block_double_cond(W, f"{p}blocks.{num_layers - 1}", mode, h_b, None, emb2, None, num_heads)
        return torch.cat([L.linear(W, p + "out.linear", h_a), L.linear(W, p + "out.linear", h_b)], dim=-1)
    emb = te + txt(cond[:, :768])
    emb1 = te + txt(cond[:, 768:768 * 2])
    emb2 = te + txt(cond[:, 768 * 2:768 * 3])
    for i in range(num_layers):
        n_a = L.block_double_cond(W, f"{p}blocks.{i}", mode, h_a, h_b, emb1, emb, num_heads)
        n_b = L.block_double_cond(W, f"{p}blocks.{i}", mode, h_b, h_a, emb2, emb, num_heads)
        h_a, h_b = n_a, n_b
    return torch.cat([L.linear(W, p + "out.linear", h_a), L.linear(W, p + "out.linear", h_b)], dim=-1)


def inter_denoiser(W, p, x, timesteps, cond, num_heads, nfeats=262):
    """InterDenoiser.forward -- intergen.py:260-288 (one shared emb, TransformerBlock)."""
    pe = W[p + "sequence_pos_encoder.pe"]
    T = x.shape[1]
    emb = L.timestep_embed(W, p + "embed_timestep", pe, timesteps) + L.linear(W, p + "text_embed", cond[:, :768])
    num_layers = 1 + max(int(k[len(p + "blocks."):].split(".")[0]) for k in W if k.startswith(p + "blocks."))
    h_a = L.linear(W, p + "motion_embed", x[..., :nfeats]) + pe[:T].unsqueeze(0)
    h_b = L.linear(W, p + "motion_embed", x[..., nfeats:]) + pe[:T].unsqueeze(0)
    for i in range(num_layers):
        n_a = L.block(W, f"{p}blocks.{i}", h_a, h_b, emb, num_heads)
        n_b = L.block(W, f"{p}blocks.{i}", h_b, h_a, emb, num_heads)
        h_a, h_b = n_a, n_b
    return torch.cat([L.linear(W, p + "out.linear", h_a), L.linear(W, p + "out.linear", h_b)], dim=-1)


def influence(W, p, mode, m_i, m_I, cond_i, cond_I, num_heads):
    """Influence.forward -- influence.py:93-126.  CA keys/values are always the INPUT m_I (:115-117)."""
    num_layers = 1 + max(int(k[len(p + "blocks."):].split(".")[0]) for k in W if k.startswith(p + "blocks."))
    h = m_i
    for i in range(num_layers):
        h = L.influence_block(W, f"{p}blocks.{i}", h, m_I, cond_i, cond_I, num_heads)
    if mode in (1, 3):
        h = h.mean(dim=1)
    return torch.sigmoid(L.linear(W, p + "out", h))
