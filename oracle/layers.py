"""Layer-level CPU restatement (oracle; test infrastructure only).

Weights are passed as a flat ``dict[str, Tensor]`` using the reference's
``state_dict`` key names under a prefix, so golden fixtures can carry the
reference module's own ``state_dict()`` unchanged.

Reference: /root/reference/src/models/utils/layers.py, blocks.py, utils.py.
"""
import math
import torch
import torch.nn.functional as F


def pe_table(d_model, max_len=5000):
    """PositionalEncoding.__init__ -- utils.py:24-35."""
    pe = torch.zeros(max_len, d_model)
    pos = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
    div = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


def linear(W, p, x):
    return F.linear(x, W[p + ".weight"], W.get(p + ".bias"))


def timestep_embed(W, p, pe, timesteps):
    """TimestepEmbedder.forward -- utils.py:41-55: Linear-SiLU-Linear on pe[t]."""
    h = linear(W, p + ".time_embed.0", pe[timesteps])
    return linear(W, p + ".time_embed.2", F.silu(h))


def adaln(W, p, h, emb):
    """AdaLN.forward -- layers.py:15-25 (scale first, shift second; LN eps 1e-6, no affine)."""
    e = linear(W, p + ".emb_layers.1", F.silu(emb))
    scale, shift = torch.chunk(e, 2, dim=-1)
    hn = F.layer_norm(h, (h.shape[-1],), eps=1e-6)
    return hn * (1 + scale[:, None]) + shift[:, None]


def mha_zero_attn(W, p, q_in, kv_in, num_heads):
    """nn.MultiheadAttention(batch_first, add_zero_attn=True), eval, no masks.

    Used at layers.py:33-44 (self) and :74-87 (cross).  Packed in_proj (q,k,v order);
    one extra key with logit 0 and value 0 is appended after projection
    (torch.nn.functional.multi_head_attention_forward, add_zero_attn branch).
    """
    D = q_in.shape[-1]
    w, b = W[p + ".in_proj_weight"], W[p + ".in_proj_bias"]
    q = F.linear(q_in, w[:D], b[:D])
    k = F.linear(kv_in, w[D:2 * D], b[D:2 * D])
    v = F.linear(kv_in, w[2 * D:], b[2 * D:])
    B, Tq, _ = q.shape
    Tk = k.shape[1]
    dh = D // num_heads
    q = q.view(B, Tq, num_heads, dh).transpose(1, 2)
    k = k.view(B, Tk, num_heads, dh).transpose(1, 2)
    v = v.view(B, Tk, num_heads, dh).transpose(1, 2)
    zk = torch.zeros(B, num_heads, 1, dh, dtype=q.dtype)
    k = torch.cat([k, zk], dim=2)
    v = torch.cat([v, zk], dim=2)
    s = (q @ k.transpose(-1, -2)) * (1.0 / math.sqrt(dh))
    a = torch.softmax(s, dim=-1)
    o = (a @ v).transpose(1, 2).reshape(B, Tq, D)
    return linear(W, p + ".out_proj", o)


def self_attention(W, p, x, emb, num_heads):
    """VanillaSelfAttention.forward -- layers.py:36-45."""
    xn = adaln(W, p + ".norm", x, emb)
    return mha_zero_attn(W, p + ".attention", xn, xn, num_heads)


def cross_attention(W, p, x, xf, emb, num_heads):
    """VanillaCrossAttention.forward -- layers.py:77-88 (both norms use the same emb)."""
    xn = adaln(W, p + ".norm", x, emb)
    xfn = adaln(W, p + ".xf_norm", xf, emb)
    return mha_zero_attn(W, p + ".attention", xn, xfn, num_heads)


def ffn(W, p, x, emb):
    """FFN.forward -- layers.py:99-106 (exact-erf GELU, dropout off)."""
    xn = adaln(W, p + ".norm", x, emb)
    return linear(W, p + ".linear2", F.gelu(linear(W, p + ".linear1", xn)))


def block_double_cond(W, p, mode, x, y, emb, emb_interaction, num_heads):
    """TransformerBlockDoubleCond.forward -- blocks.py:49-63."""
    h1 = self_attention(W, p + ".sa_block", x, emb, num_heads) + x
    if mode in ("individual", "dual_individual"):
        h2 = h1
    else:
        h2 = cross_attention(W, p + ".ca_block", h1, y, emb_interaction, num_heads) + h1
    return ffn(W, p + ".ffn", h2, emb) + h2


def block(W, p, x, y, emb, num_heads):
    """TransformerBlock.forward -- blocks.py:21-28 (InterGen's block: one emb)."""
    h1 = self_attention(W, p + ".sa_block", x, emb, num_heads) + x
    h2 = cross_attention(W, p + ".ca_block", h1, y, emb, num_heads) + h1
    return ffn(W, p + ".ffn", h2, emb) + h2


def influence_block(W, p, m_i, m_I, emb_i, emb_I, num_heads):
    """InfluenceBlockCross.forward -- influence.py:34-48."""
    h1 = self_attention(W, p + ".sa_block", m_i, emb_i, num_heads) + m_i
    h2 = cross_attention(W, p + ".ca_block", h1, m_I, emb_I, num_heads) + h1
    return ffn(W, p + ".ffn", h2, emb_I) + h2
