"""Post-/pre-norm transformer encoder pieces, CPU restatement (oracle; test infrastructure only).

Covers the modules of the path that are plain ``torch.nn`` encoders rather than AdaLN blocks:
  * ``nn.TransformerEncoderLayer`` (post-norm, exact GELU, LayerNorm eps 1e-5) as used by MDMDenoiser.seqTransEncoder
    (/root/reference/src/models/mdm.py:252-264) and by the clipTransEncoder text heads
    (/root/reference/src/models/mixermdm.py:246-258, /root/reference/src/models/in2in.py:60-90);
  * MDMDenoiser.forward (/root/reference/src/models/mdm.py:273-298);
  * the text-conditioning stage ``text_process`` (mixermdm.py:283-312, in2in.py:109-135, mdm.py:99-120).

The CLIP text tower itself (token embedding, 12 pre-norm residual attention blocks with a causal mask and QuickGELU,
ln_final) is NOT under /root/reference: it comes from the third-party package ``clip==1.0`` (environment.yaml:49;
call sites mixermdm.py:212-217, 297-303).  ``clip_text_tower`` restates its published architecture (CLIP model.py:
ResidualAttentionBlock / Transformer / encode_text).  PARITY UNPINNED for that function: the package and its weights are
absent here, so no reference vectors exist for it; everything downstream of the tower is pinned by tests/golden/text.npz.
"""
import math
import torch
import torch.nn.functional as F
from . import layers as L


def layer_norm(W, p, x, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), W[p + ".weight"], W[p + ".bias"], eps)


def mha_plain(W, p, x, num_heads, causal=False):
    """nn.MultiheadAttention self-attention without add_zero_attn; optional causal mask (CLIP's build_attention_mask:
    -inf above the diagonal).  Packed in_proj (q,k,v order)."""
    D = x.shape[-1]
    w, b = W[p + ".in_proj_weight"], W[p + ".in_proj_bias"]
    q, k, v = F.linear(x, w, b).chunk(3, dim=-1)
    B, T, _ = x.shape
    dh = D // num_heads
    q = q.view(B, T, num_heads, dh).transpose(1, 2)
    k = k.view(B, T, num_heads, dh).transpose(1, 2)
    v = v.view(B, T, num_heads, dh).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) * (1.0 / math.sqrt(dh))
    if causal:
        s = s + torch.full((T, T), float("-inf")).triu_(1)
    o = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(B, T, D)
    return L.linear(W, p + ".out_proj", o)


def encoder_layer(W, p, x, num_heads):
    """nn.TransformerEncoderLayer(batch_first=True, norm_first=False, activation="gelu"), eval, no masks:
    x = norm1(x + SA(x)); x = norm2(x + linear2(gelu(linear1(x))))."""
    x = layer_norm(W, p + ".norm1", x + mha_plain(W, p + ".self_attn", x, num_heads))
    return layer_norm(W, p + ".norm2", x + L.linear(W, p + ".linear2", F.gelu(L.linear(W, p + ".linear1", x))))


def _num_layers(W, p):
    return 1 + max([int(k[len(p):].split(".")[0]) for k in W if k.startswith(p)], default=-1)


def encoder(W, p, x, num_heads):
    """nn.TransformerEncoder (no final norm): p + "layers.{i}"."""
    for i in range(_num_layers(W, p + "layers.")):
        x = encoder_layer(W, f"{p}layers.{i}", x, num_heads)
    return x


def mdm_denoiser(W, p, x, timesteps, cond, num_heads):
    """MDMDenoiser.forward, mask=None -- mdm.py:273-298.  cond [B, D] already lives in the latent space (MDM.embed_text,
    mdm.py:37,116); the timestep embedding is ADDED to it and the sum is the extra token 0 of the sequence; the
    positional encoding covers all T+1 tokens; the output drops token 0.  (The reference adds in place, ``cond += ...``,
    on the caller's tensor -- SURVEY quirk 10; the oracle does not mutate its argument.)"""
    pe = W[p + "sequence_pos_encoder.pe"]
    T = x.shape[1]
    tok = cond + L.timestep_embed(W, p + "embed_timestep", pe, timesteps)
    h = L.linear(W, p + "input_process.poseEmbedding", x)
    h = torch.cat([tok.unsqueeze(1), h], dim=1) + pe[:T + 1].unsqueeze(0)
    h = encoder(W, p + "seqTransEncoder.", h, num_heads)[:, 1:]
    return L.linear(W, p + "output_process.poseFinal", h)


# ---- text-conditioning stage ----------------------------------------------------------------------

def quick_gelu(x):
    return x * torch.sigmoid(1.702 * x)


def clip_text_tower(W, p, tokens, num_heads):
    """CLIP text transformer up to ln_final (clip==1.0 model.py: token_embedding + positional_embedding, then
    ``resblocks.{i}``: x += attn(ln_1(x), causal); x += c_proj(QuickGELU(c_fc(ln_2(x)))); ln_final).  PARITY UNPINNED
    (see the module docstring).  Keys follow the names the reference aliases them to (mixermdm.py:213-216):
    token_embedding.weight, positional_embedding, clip_transformer.resblocks.{i}.{ln_1,attn,ln_2,mlp.c_fc,mlp.c_proj}, ln_final."""
    x = W[p + "token_embedding.weight"][tokens] + W[p + "positional_embedding"][: tokens.shape[1]]
    for i in range(_num_layers(W, p + "clip_transformer.resblocks.")):
        q = f"{p}clip_transformer.resblocks.{i}"
        x = x + mha_plain(W, q + ".attn", layer_norm(W, q + ".ln_1", x), num_heads, causal=True)
        x = x + L.linear(W, q + ".mlp.c_proj", quick_gelu(L.linear(W, q + ".mlp.c_fc", layer_norm(W, q + ".ln_2", x))))
    return layer_norm(W, p + "ln_final", x)


def text_head(W, enc_p, ln_p, clip_out, tokens, num_heads=8):
    """text_process after the tower -- mixermdm.py:305-312 / in2in.py:122-133: 2-layer post-norm encoder over all 77 tokens
    (no padding mask), LayerNorm, then the row at the EOT token = argmax of the token ids."""
    out = layer_norm(W, ln_p, encoder(W, enc_p, clip_out, num_heads))
    return out[torch.arange(tokens.shape[0]), tokens.argmax(dim=-1)]


def clip_encode_text(W, p, tokens, num_heads):
    """CLIP.encode_text (clip==1.0 model.py): tower, EOT row, @ text_projection; used by MDM.text_process (mdm.py:115).
    PARITY UNPINNED (third-party)."""
    x = clip_text_tower(W, p, tokens, num_heads)
    return x[torch.arange(tokens.shape[0]), tokens.argmax(dim=-1)] @ W[p + "text_projection"]
