"""Seeded synthetic weights / inputs with the reference's state_dict key names (SURVEY.md 8d "Synthetic inputs").

There is no network for checkpoints or datasets, so benchmarks and parity tests use random-init weights of the
reference architecture: every parameter N(0, std) from a CPU torch.Generator (zero_module'd layers included, SURVEY
quirk 11), biases zero unless ``bias_std`` is given.  Key names follow Mixer.state_dict() (src/models/mixermdm.py:134-148).
"""
import torch


def _block_shapes(D, F, with_ca=True):
    sh = {
        "sa_block.norm.emb_layers.1.weight": (2 * D, D), "sa_block.norm.emb_layers.1.bias": (2 * D,),
        "sa_block.attention.in_proj_weight": (3 * D, D), "sa_block.attention.in_proj_bias": (3 * D,),
        "sa_block.attention.out_proj.weight": (D, D), "sa_block.attention.out_proj.bias": (D,),
        "ffn.norm.emb_layers.1.weight": (2 * D, D), "ffn.norm.emb_layers.1.bias": (2 * D,),
        "ffn.linear1.weight": (F, D), "ffn.linear1.bias": (F,),
        "ffn.linear2.weight": (D, F), "ffn.linear2.bias": (D,),
    }
    if with_ca:
        sh.update({
            "ca_block.norm.emb_layers.1.weight": (2 * D, D), "ca_block.norm.emb_layers.1.bias": (2 * D,),
            "ca_block.xf_norm.emb_layers.1.weight": (2 * D, D), "ca_block.xf_norm.emb_layers.1.bias": (2 * D,),
            "ca_block.attention.in_proj_weight": (3 * D, D), "ca_block.attention.in_proj_bias": (3 * D,),
            "ca_block.attention.out_proj.weight": (D, D), "ca_block.attention.out_proj.bias": (D,),
        })
    return sh


def denoiser_shapes(prefix, D, F, L, text_dim=768, nfeats=262):
    """in2INDenoiser / InterDenoiser parameters (src/models/in2in.py:358-399): both modes carry the ca_block weights."""
    sh = {
        "motion_embed.weight": (D, nfeats), "motion_embed.bias": (D,),
        "text_embed.weight": (D, text_dim), "text_embed.bias": (D,),
        "embed_timestep.time_embed.0.weight": (D, D), "embed_timestep.time_embed.0.bias": (D,),
        "embed_timestep.time_embed.2.weight": (D, D), "embed_timestep.time_embed.2.bias": (D,),
        "out.linear.weight": (nfeats, D), "out.linear.bias": (nfeats,),
    }
    for i in range(L):
        for k, v in _block_shapes(D, F).items():
            sh[f"blocks.{i}.{k}"] = v
    return {prefix + k: v for k, v in sh.items()}


def mdm_denoiser_shapes(prefix, D, F, L, nfeats=262):
    """MDMDenoiser parameters (src/models/mdm.py:234-271): pose embedding / head, timestep MLP, post-norm nn.TransformerEncoder layers.
    LayerNorm weights are named *.norm{1,2}.weight (drawn like any other 1-d parameter: callers that want gamma ~ 1 add it)."""
    sh = {
        "input_process.poseEmbedding.weight": (D, nfeats), "input_process.poseEmbedding.bias": (D,),
        "embed_timestep.time_embed.0.weight": (D, D), "embed_timestep.time_embed.0.bias": (D,),
        "embed_timestep.time_embed.2.weight": (D, D), "embed_timestep.time_embed.2.bias": (D,),
        "output_process.poseFinal.weight": (nfeats, D), "output_process.poseFinal.bias": (nfeats,),
    }
    for i in range(L):
        q = f"seqTransEncoder.layers.{i}."
        sh.update({q + "self_attn.in_proj_weight": (3 * D, D), q + "self_attn.in_proj_bias": (3 * D,),
                   q + "self_attn.out_proj.weight": (D, D), q + "self_attn.out_proj.bias": (D,),
                   q + "linear1.weight": (F, D), q + "linear1.bias": (F,), q + "linear2.weight": (D, F), q + "linear2.bias": (D,),
                   q + "norm1.weight": (D,), q + "norm1.bias": (D,), q + "norm2.weight": (D,), q + "norm2.bias": (D,)})
    return {prefix + k: v for k, v in sh.items()}


def mixer_shapes(d_latent, d_ff, d_layers, m_latent, m_ff, m_layers, mixing_mode=4, text_dim=768, nfeats=262, single_only=False,
                 model1="in2INind", d1_latent=0, d1_ff=0, d1_layers=0):
    D1, F1, L1 = d1_latent or d_latent, d1_ff or d_ff, d1_layers or d_layers
    sh = mdm_denoiser_shapes("denoiser1.", D1, F1, L1, nfeats) if model1 == "MDM" else denoiser_shapes("denoiser1.", D1, F1, L1, text_dim, nfeats)
    if single_only:
        return sh
    sh.update(denoiser_shapes("denoiser2.", d_latent, d_ff, d_layers, text_dim, nfeats))
    nw = 23 if mixing_mode >= 3 else 1
    sh.update({
        "motion_embed.weight": (m_latent, nfeats), "motion_embed.bias": (m_latent,),
        "text_embed.weight": (m_latent, text_dim), "text_embed.bias": (m_latent,),
        "embed_timestep.time_embed.0.weight": (m_latent, m_latent), "embed_timestep.time_embed.0.bias": (m_latent,),
        "embed_timestep.time_embed.2.weight": (m_latent, m_latent), "embed_timestep.time_embed.2.bias": (m_latent,),
        "influence.out.weight": (nw, m_latent), "influence.out.bias": (nw,),
    })
    for i in range(m_layers):
        for k, v in _block_shapes(m_latent, m_ff).items():
            sh[f"influence.blocks.{i}.{k}"] = v
    return sh


def synthetic_state_dict(seed=0, std=0.02, bias_std=0.0, **dims):
    """CPU fp32 state dict (no pe buffers; the Sampler regenerates them)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, shape in mixer_shapes(**dims).items():
        if len(shape) == 1 and (k.endswith("norm1.weight") or k.endswith("norm2.weight")):
            sd[k] = 1 + torch.randn(shape, generator=g) * bias_std if bias_std else torch.ones(shape)      # LayerNorm gamma
        elif len(shape) == 1:
            sd[k] = torch.randn(shape, generator=g) * bias_std if bias_std else torch.zeros(shape)
        else:
            sd[k] = torch.randn(shape, generator=g) * std
    return sd


def synthetic_stats(seed=3):
    """Normaliser statistics: mean N(0, 0.1), std U(0.5, 1.5) (data files are not in the repo)."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k in ("mean_hml", "mean_ih"):
        out[k] = torch.randn(262, generator=g) * 0.1
    for k in ("std_hml", "std_ih"):
        out[k] = torch.rand(262, generator=g) + 0.5
    return out


def synthetic_inputs(B, T, seed_cond=1, seed_x=2, single=False, text_dim=768):
    cond = torch.randn(B, text_dim if single else 8 * text_dim, generator=torch.Generator().manual_seed(seed_cond))
    x_T = torch.randn(B, T, 262 if single else 524, generator=torch.Generator().manual_seed(seed_x))
    return cond, x_T


FULL_DIMS = dict(d_latent=1024, d_ff=2048, d_layers=8, m_latent=512, m_ff=1024, m_layers=4)   # configs/models/*.yaml
