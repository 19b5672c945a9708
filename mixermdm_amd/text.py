"""Text-conditioning stage on the GPU: the reference's ``text_process`` / ``generate_cond``.

Reference: MixerMDM.text_process / generate_cond  src/models/mixermdm.py:283-356; in2IN.text_process  src/models/in2in.py:109-135;
InterGen.text_process  src/models/intergen.py:68-93.  Per prompt batch the reference runs the frozen CLIP text tower
(token + positional embedding, 12 causal pre-norm blocks with QuickGELU, ln_final), then a trainable 2-layer post-norm
``nn.TransformerEncoder`` + LayerNorm head, and keeps the row at the EOT token (argmax of the token ids).  MixerMDM does this 8
times per batch (5 through the sub-models' heads, 3 through its own) on only THREE distinct prompts: here the tower runs once
per distinct prompt list and its output is shared by the heads.

All arithmetic runs in the HIP library (token embedding, LayerNorm, GEMMs, attention, gathers); Python only orders the calls.
Tokenisation is host string processing that belongs to the third-party ``clip`` package (BPE vocabulary file): ``tokenize`` uses
it when importable, otherwise callers pass token ids (``batch["tokens_<name>"]``).
"""
import torch

from . import ops

_ENC_KEYS = ("self_attn.in_proj_weight", "self_attn.in_proj_bias", "self_attn.out_proj.weight", "self_attn.out_proj.bias",
             "linear1.weight", "linear1.bias", "linear2.weight", "linear2.bias", "norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias")
_CLIP_MAP = {"in_proj_weight": "attn.in_proj_weight", "in_proj_bias": "attn.in_proj_bias", "out_proj.weight": "attn.out_proj.weight",
             "out_proj.bias": "attn.out_proj.bias", "linear1.weight": "mlp.c_fc.weight", "linear1.bias": "mlp.c_fc.bias",
             "linear2.weight": "mlp.c_proj.weight", "linear2.bias": "mlp.c_proj.bias", "norm1.weight": "ln_1.weight", "norm1.bias": "ln_1.bias",
             "norm2.weight": "ln_2.weight", "norm2.bias": "ln_2.bias"}


def tokenize(texts, context_length=77):
    """clip.tokenize(texts, truncate=True) when the ``clip`` package is installed (the reference's own dependency, clip==1.0)."""
    try:
        import clip
    except ImportError as e:
        raise RuntimeError("tokenisation needs the `clip` package (BPE vocabulary); pass token ids instead, e.g. batch['tokens_text']") from e
    return clip.tokenize(texts, context_length=context_length, truncate=True)


def _count(sd, prefix):
    idx = [int(k[len(prefix):].split(".")[0]) for k in sd if k.startswith(prefix)]
    return 1 + max(idx) if idx else 0


class ClipTextTower:
    """token_embedding + positional_embedding + clip_transformer.resblocks + ln_final, as aliased onto the reference modules
    (src/models/mixermdm.py:212-217).  State-dict keys: ``<p>token_embedding.weight``, ``<p>positional_embedding``,
    ``<p>clip_transformer.resblocks.{i}.{ln_1,attn,ln_2,mlp.c_fc,mlp.c_proj}.*``, ``<p>ln_final.{weight,bias}``."""

    def __init__(self, sd, prefix="", num_heads=12, device="cuda", blocks="clip_transformer.resblocks."):
        """blocks: "clip_transformer.resblocks." for the aliased modules, "transformer.resblocks." for a whole CLIP model
        (MDM.clip_model, src/models/mdm.py:39,72)."""
        g = lambda k: sd[prefix + k].detach().to(device=device, dtype=torch.float32).contiguous()
        self.table, self.pos = g("token_embedding.weight"), g("positional_embedding")
        self.ln_w, self.ln_b = g("ln_final.weight"), g("ln_final.bias")
        self.num_heads = num_heads
        self.layers = []
        for i in range(_count(sd, prefix + blocks)):
            q = f"{blocks}{i}."
            self.layers.append({k: g(q + v) for k, v in _CLIP_MAP.items()})
        self.text_projection = g("text_projection") if prefix + "text_projection" in sd else None

    def __call__(self, tokens):
        """tokens [B, L] (any int dtype, host or device) -> ln_final(transformer(embed(tokens))) [B, L, D]."""
        tok = tokens.to(device=self.table.device, dtype=torch.int32).contiguous()
        x = ops.token_embed(self.table, tok, self.pos)
        ws = None
        for w in self.layers:
            ops.encoder_layer_(x, w, self.num_heads, norm_first=True, activation="quickgelu", causal=True, eps=1e-5, workspace=ws)
        return ops.layernorm(x, self.ln_w, self.ln_b, 1e-5)

    def encode_text(self, tokens):
        """CLIP.encode_text: EOT row of the tower output @ text_projection (MDM.text_process, src/models/mdm.py:115)."""
        x = self(tokens)
        eot = _eot_rows(tokens, x)
        return ops.linear(eot, self.text_projection.t().contiguous())


def _eot_rows(tokens, x):
    B, L, D = x.shape
    idx = (torch.arange(B) * L + tokens.cpu().long().argmax(dim=-1)).to(device=x.device, dtype=torch.int32)
    return ops.gather_rows(x.reshape(B * L, D), idx)


class TextHead:
    """clipTransEncoder (nn.TransformerEncoder, 2 post-norm layers, 8 heads, exact GELU) + clip_ln + EOT gather
    (src/models/mixermdm.py:305-312).  ``enc_prefix`` e.g. "clipTransEncoder." / "clipTransEncoder_individual.", ``ln_prefix`` "clip_ln" ..."""

    def __init__(self, sd, enc_prefix, ln_prefix, num_heads=8, device="cuda"):
        g = lambda k: sd[k].detach().to(device=device, dtype=torch.float32).contiguous()
        self.layers = []
        for i in range(_count(sd, enc_prefix + "layers.")):
            q = f"{enc_prefix}layers.{i}."
            self.layers.append({k.replace("self_attn.", ""): g(q + k) for k in _ENC_KEYS})
        self.ln_w, self.ln_b = g(ln_prefix + ".weight"), g(ln_prefix + ".bias")
        self.num_heads = num_heads

    def __call__(self, clip_out, tokens):
        x = clip_out.clone()
        for w in self.layers:
            ops.encoder_layer_(x, w, self.num_heads, norm_first=False, activation="gelu", causal=False, eps=1e-5)
        B, L, D = x.shape
        return _eot_rows(tokens, ops.layernorm(x, self.ln_w, self.ln_b, 1e-5))


def tokenize_mdm(texts):
    """MDM.text_process tokenisation (src/models/mdm.py:103-111): context 22 (20 words + SOT/EOT), zero-padded to 77."""
    t = tokenize(texts, context_length=22)
    return torch.cat([t, torch.zeros(t.shape[0], 77 - 22, dtype=t.dtype)], dim=1)


class MdmTextHead:
    """MDM.text_process (src/models/mdm.py:99-120): its own CLIP model's encode_text, then embed_text (Linear 512 -> latent).
    Keys: ``<p>clip_model.{token_embedding.weight,positional_embedding,transformer.resblocks.*,ln_final.*,text_projection}``,
    ``<p>embed_text.{weight,bias}``."""

    def __init__(self, sd, prefix="model1.", num_heads=8, device="cuda"):
        self.tower = ClipTextTower(sd, prefix + "clip_model.", num_heads, device, blocks="transformer.resblocks.")
        self.w = sd[prefix + "embed_text.weight"].detach().to(device=device, dtype=torch.float32).contiguous()
        self.b = sd[prefix + "embed_text.bias"].detach().to(device=device, dtype=torch.float32).contiguous()

    def __call__(self, tokens):
        return ops.linear(self.tower.encode_text(tokens), self.w, self.b)


class MixerTextEncoder:
    """MixerMDM.generate_cond (src/models/mixermdm.py:314-356): the 8 condition vectors of a batch, concatenated in the order
    [interaction, int_ind1, int_ind2, ind_ind1, ind_ind2, infl_I, infl_i1, infl_i2].

    sd: the MixerMDM state dict (keys as in src/models/mixermdm.py:134-256): ``token_embedding.*``, ``positional_embedding``,
    ``clip_transformer.*``, ``ln_final.*`` (the shared tower), ``clipTransEncoder.* / clip_ln.*`` (mixer head),
    ``model1.clipTransEncoder_individual.* / model1.clip_ln_individual.*``, ``model2.clipTransEncoder_interaction.* /
    model2.clip_ln_interaction.*`` (in2IN sub-models; InterGen as MODEL2 uses ``model2.clipTransEncoder.* / model2.clip_ln.*``)."""

    def __init__(self, sd, clip_heads=12, head_heads=8, device="cuda", model2="in2IN", model1="in2INind", mdm_clip_heads=8):
        self.tower = ClipTextTower(sd, "", clip_heads, device)
        self.head_mixer = TextHead(sd, "clipTransEncoder.", "clip_ln", head_heads, device)
        self.mdm = MdmTextHead(sd, "model1.", mdm_clip_heads, device) if model1 == "MDM" else None
        self.head_ind = None if self.mdm else TextHead(sd, "model1.clipTransEncoder_individual.", "model1.clip_ln_individual", head_heads, device)
        if model2 == "InterGen":
            self.head_int = TextHead(sd, "model2.clipTransEncoder.", "model2.clip_ln", head_heads, device)
        else:
            self.head_int = TextHead(sd, "model2.clipTransEncoder_interaction.", "model2.clip_ln_interaction", head_heads, device)

    def generate_cond(self, batch):
        def toks(name):
            if "tokens_" + name in batch:
                return torch.as_tensor(batch["tokens_" + name])
            return tokenize(batch[name])
        t1, t2 = toks("text_individual1"), toks("text_individual2")
        tI = toks("text_interaction") if ("text_interaction" in batch or "tokens_text_interaction" in batch) else toks("text")
        c1, c2, cI = self.tower(t1), self.tower(t2), self.tower(tI)
        if self.mdm:      # MODEL1 = MDM: its own tokenisation (context 22) and CLIP model (mixermdm.py:326-327 -> mdm.py:99-120)
            mt = lambda name: torch.as_tensor(batch["tokens_mdm_" + name]) if "tokens_mdm_" + name in batch else tokenize_mdm(batch[name])
            ind = [self.mdm(mt("text_individual1")), self.mdm(mt("text_individual2"))]
        else:
            ind = [self.head_ind(c1, t1), self.head_ind(c2, t2)]
        parts = [self.head_int(cI, tI), self.head_int(c1, t1), self.head_int(c2, t2), *ind,
                 self.head_mixer(cI, tI), self.head_mixer(c1, t1), self.head_mixer(c2, t2)]
        return torch.cat(parts, dim=1)
