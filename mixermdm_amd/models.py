"""Host-side mirror of the reference's `src/models` operator API for the denoising path, backed by libmmdm_hip.so.

Reference surface mirrored:
  * ``MixerMDM(cfg, num_frames=300, sampling_strategy="ddim50", store_influence=True, align=True)`` with
    ``forward(batch)`` / ``forward_test(batch)`` / ``load_state_dict`` and the knobs ``.mixing.mode``,
    ``.mixing.force_influence_val``, ``.mixing_mode``, ``.sampling_strategy``, ``.cfg_mixing_weight``
    (src/models/mixermdm.py:18-19, 490-602);
  * the inner callables: ``Mixer(x1, timesteps, cond, mask, x2)`` (mixermdm.py:660), denoisers
    ``f(x, timesteps, cond=, mask=)`` (in2in.py:401), ``ClassifierFreeSampleModelX2`` (cfg_sampler.py:31-56) and
    ``MixerDiffusion.ddim_sample_loop`` (gaussian_diffusion.py:1769-1820).

Python only moves pointers: parameters live in torch tensors under the reference's state_dict key names and are copied
into the HIP handle's packed layout; every arithmetic op of the loop is a HIP kernel.  Text encoding (CLIP tower +
clipTransEncoder, mixermdm.py:283-356) is upstream of this path: pass ``batch["cond"]`` ([B, 8*768], layout
mixermdm.py:342-354) or register ``text_encoder``.
"""
import os
import numpy as np
import torch
from torch import nn

from .configs import get_config
from .sampler import Sampler, pe_table
from .schedule import get_named_beta_schedule, space_timesteps, RespacedSchedule
from .synthetic import mixer_shapes, synthetic_state_dict, synthetic_stats

# state_dict prefixes of the reference's MixerMDM that are NOT on the denoising path (CLIP tower, sub-model facades,
# training-only discriminators); accepted and ignored by load_state_dict.
OFF_PATH_PREFIXES = ("model1.", "model2.", "denoiser1.", "denoiser2.", "discriminator_i.", "discriminator_I.", "clipTransEncoder.", "clip_ln.",
                     "token_embedding.", "clip_transformer.", "positional_embedding", "ln_final.")

HISTORY_BUDGET_BYTES = 32 << 30


def _holder_tree(root, name, tensor):
    """Register `tensor` as parameter `name` ("a.b.0.weight") on nested container modules so that state_dict() yields the
    reference's key names."""
    parts = name.split(".")
    mod = root
    for p in parts[:-1]:
        if p not in mod._modules:
            mod.add_module(p, nn.Module())
        mod = mod._modules[p]
    mod.register_parameter(parts[-1], nn.Parameter(tensor, requires_grad=False))


class _Callable(nn.Module):
    """Base of the inner callables: shares the owner's sampler."""

    def __init__(self, owner):
        super().__init__()
        object.__setattr__(self, "_owner", owner)

    @staticmethod
    def _uniform_t(timesteps):
        t = timesteps.reshape(-1)
        if not bool((t == t[0]).all()):
            raise NotImplementedError("the HIP path runs one timestep per call (the sampler always passes t = [i]*B: gaussian_diffusion.py:1872)")
        return int(t[0])


class DenoiserHandle(_Callable):
    """in2INDenoiser / InterDenoiser protocol: f(x [n,T,nf*k], timesteps [n], cond=[n,.], mask=None) -> [n,T,nf*k]."""

    def __init__(self, owner, which):
        super().__init__(owner)
        self.which = which
        self.text_dim = 768

    def forward(self, x, timesteps, mask=None, cond=None):
        if mask is not None:
            raise NotImplementedError("key_padding masks are not used on the inference path (mask=None: mixermdm.py:533)")
        n = x.shape[0]
        if n % 2:
            raise ValueError("the HIP denoiser works on the CFG-doubled batch (even number of rows)")
        smp = self._owner._sampler_for(n // 2, x.shape[1])
        return smp.module_forward(self.which, x, cond, self._uniform_t(timesteps))


class Mixer(_Callable):
    """Mixer.forward protocol (mixermdm.py:660): out_influenced for the CFG-doubled batch.  Holds the reference's knobs."""

    def __init__(self, owner, mixing_mode, store_influence, force_influence_val, mode="train", align=True):
        super().__init__(owner)
        self.mixing_mode = mixing_mode
        self.store_influence = store_influence
        self.force_influence_val = force_influence_val
        self.mode = mode
        self.align = align
        self.history_influence_i1, self.history_influence_i2 = [], []
        self.history_out1, self.history_out2, self.history_out_influenced = [], [], []

    def forward(self, x1, timesteps, cond=None, mask=None, x2=None):
        if mask is not None:
            raise NotImplementedError("mask must be None on the inference path")
        if self.mixing_mode not in (1, 2, 3, 4):
            raise ValueError("Mixing mode not recognized")
        smp = self._owner._sampler_for(x1.shape[0] // 2, x1.shape[1])
        return smp.module_forward(2, x1, cond, self._uniform_t(timesteps), x2=x2)


class ClassifierFreeSampleModelX2(nn.Module):
    """cfg_sampler.py:31-56.  Inside the sampling loop the doubling and the s*cond + (1-s)*uncond combine are part of the captured HIP
    step; called on its own (the reference's inner-callable protocol ``f(x, x2, timesteps, cond, mask)``, cfg_sampler.py:38) it runs the
    same kernels through ``mmdm_module_forward(which=4)``: B un-doubled rows in, B combined rows out."""

    def __init__(self, model, cfg_scale):
        super().__init__()
        self.model = model
        self.s = cfg_scale

    def forward(self, x, x2, timesteps, cond=None, mask=None):
        if mask is not None:
            raise NotImplementedError("mask must be None on the inference path")
        if cond is None:
            raise NotImplementedError("the HIP mixer is text-conditioned: cond [B, 8*768] is required (mixermdm.py:342-354)")
        mixer = self.model
        if mixer.mixing_mode not in (1, 2, 3, 4):
            raise ValueError("Mixing mode not recognized")
        owner = mixer._owner
        smp = owner._sampler_for(x.shape[0], x.shape[1], cfg_scale=self.s)
        return smp.module_forward(4, x, cond, _Callable._uniform_t(timesteps), x2=x2)


class MixerDiffusion:
    """Two-chain DDIM sampler (gaussian_diffusion.py:1434-1463, 1769-1965): schedule tables on the host, loop on the GPU."""

    def __init__(self, use_timesteps, align=True, *, betas, **kwargs):
        self.align = align
        self.schedule = RespacedSchedule(betas, use_timesteps)
        self.timestep_map = self.schedule.timestep_map
        self.num_timesteps = self.schedule.num_timesteps
        self.original_num_steps = self.schedule.original_num_steps

    def ddim_sample_loop(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None,
                         device=None, progress=False, eta=0.0, skip_timesteps=0, init_image=None, randomize_class=False,
                         cond_fn_with_grad=False, dump_steps=None, const_noise=False, x_start=None):
        if dump_steps is not None or const_noise:
            raise NotImplementedError()                                   # gaussian_diffusion.py:1794-1797
        if clip_denoised or denoised_fn is not None or cond_fn is not None or eta != 0.0 or skip_timesteps or init_image is not None \
                or randomize_class or cond_fn_with_grad or x_start is not None:
            raise NotImplementedError("HIP sampler implements the configuration MixerMDM.forward uses: clip_denoised=False, eta=0, no guidance fn")
        mixer = model.model if isinstance(model, ClassifierFreeSampleModelX2) else model
        owner = mixer._owner
        model_kwargs = model_kwargs or {}
        if model_kwargs.get("mask") is not None:
            raise NotImplementedError("mask must be None")
        cond = model_kwargs["cond"]
        B, T, _ = shape
        dev = owner.device
        x_T = noise if noise is not None else torch.randn(*shape, device=dev)
        return owner._run_loop(self, cond, x_T, cfg_scale=getattr(model, "s", owner.cfg_mixing_weight))


class MixerMDM(nn.Module):
    def __init__(self, cfg, num_frames=300, sampling_strategy="ddim50", store_influence=True, align=True, config_root=None):
        super().__init__()
        self.cfg = cfg
        root = config_root or os.getcwd()
        self.cfg_model1 = get_config(os.path.join(root, cfg.MODEL1) if not os.path.isabs(cfg.MODEL1) else cfg.MODEL1)
        self.cfg_model2 = get_config(os.path.join(root, cfg.MODEL2) if not os.path.isabs(cfg.MODEL2) else cfg.MODEL2)
        if self.cfg_model1.NAME not in ("in2INind", "MDM"):
            raise NotImplementedError(f"MODEL1.NAME={self.cfg_model1.NAME}")
        if self.cfg_model2.NAME not in ("in2IN", "InterGen"):
            raise NotImplementedError(f"MODEL2.NAME={self.cfg_model2.NAME}")
        self.align = align
        self.store_influence = store_influence
        self.num_frames = num_frames
        g = cfg.GENERATOR if "GENERATOR" in cfg else cfg
        self.nfeats = g.INPUT_DIM
        c1, c2 = self.cfg_model1, self.cfg_model2
        # MODEL1 and MODEL2 are separate configs (mixermdm.py:32-40): denoiser1 carries its own sizes (d1_*), denoiser2 the d_* ones
        self.dims = dict(d_latent=c2.LATENT_DIM, d_ff=c2.FF_SIZE, d_layers=c2.NUM_LAYERS, m_latent=g.LATENT_DIM, m_ff=g.FF_SIZE, m_layers=g.NUM_LAYERS,
                         d1_latent=c1.LATENT_DIM, d1_ff=c1.FF_SIZE, d1_layers=c1.NUM_LAYERS)
        self.d_heads, self.d1_heads, self.m_heads = c2.NUM_HEADS, c1.NUM_HEADS, g.NUM_HEADS
        self.model1_kind = 1 if c1.NAME == "MDM" else 0
        self.cfg_mixing_weight = cfg.CFG_WEIGHT
        self.text_dim = 768
        self.mixing_mode = cfg.MIXING_MODE
        self.diffusion_steps = cfg.DIFFUSION_STEPS
        self.beta_scheduler = cfg.BETA_SCHEDULER
        self.sampling_strategy = sampling_strategy
        self.betas = get_named_beta_schedule(self.beta_scheduler, self.diffusion_steps)
        self.history_every = 1
        self.precision = "fp32"             # "fp32" (native fp32 MFMA, the parity path) | "fp32_split" (same accuracy, bf16 matrix cores) | "bf16"
        self.text_encoder = None            # optional callable(batch) -> cond [B, 8*768] overriding the built-in text stage
        self._text_sd, self._text_enc = None, None
        self.mixing = Mixer(self, self.mixing_mode, store_influence, cfg.FORCE_INFLUENCE_VAL, align=align)
        self.mixing.add_module("denoiser1", DenoiserHandle(self, 0))
        if self.model1_kind == 1:
            # MDMDenoiser.text_dim is hard-coded to 256 (mdm.py:238) while its cond is added to the latent-sized timestep embedding
            # (mdm.py:279): the cond slice width is the latent size (they coincide for the reference's MDM, LATENT_DIM = 256).
            self.mixing.denoiser1.text_dim = c1.LATENT_DIM
        self.mixing.add_module("denoiser2", DenoiserHandle(self, 1))
        # the reference also exposes them as .denoiser1/.denoiser2 (mixermdm.py:67-68); plain attributes here, not re-registered
        object.__setattr__(self, "denoiser1", self.mixing.denoiser1)
        object.__setattr__(self, "denoiser2", self.mixing.denoiser2)
        # parameters under the reference's key names (mixing.*), zero-initialised until loaded
        self._model1 = "MDM" if self.model1_kind else "in2INind"
        for k, shp in mixer_shapes(mixing_mode=self.mixing_mode, model1=self._model1, **self.dims).items():
            _holder_tree(self, "mixing." + k, torch.zeros(shp))
        self._sampler = None
        self._dirty = True
        self._stats = None
        self._try_load_norm_stats()
        self._try_load_submodel_checkpoints()

    # ---- weights / stats -----------------------------------------------------------------------------
    def _try_load_norm_stats(self):
        """MotionNormalizerTorch{,HML3D}.__init__ paths (src/utils/utils.py:46-47, 66-67)."""
        paths = ["./data/HumanML3D/mean_ih_new.npy", "./data/HumanML3D/std_ih_new.npy", "./data/global_mean.npy", "./data/global_std.npy"]
        if all(os.path.exists(p) for p in paths):
            self.set_norm_stats(*[np.load(p) for p in paths])

    def _try_load_submodel_checkpoints(self):
        """The reference's constructor loads the two frozen sub-models from MODEL1/MODEL2.CHECKPOINT (mixermdm.py:43-59): raw state dicts
        for in2IN (keys ``decoder.net_{individual,interaction}.*``), ``{"state_dict": ...}`` with a Lightning ``model.`` prefix for MDM
        (denoiser under ``model.*``) and InterGen (``decoder.net.*``).  Here they are optional: when a file exists its denoiser weights
        initialise ``mixing.denoiser{1,2}.*`` (a MixerMDM checkpoint loaded afterwards carries the same tensors and overrides them)."""
        own = dict(self.named_parameters())
        for which, cfgm in (("1", self.cfg_model1), ("2", self.cfg_model2)):
            path = cfgm.CHECKPOINT if "CHECKPOINT" in cfgm else None
            if not path or not os.path.exists(path):
                continue
            ck = torch.load(path, map_location="cpu")
            if cfgm.NAME == "MDM":
                sd = {k[6:]: v for k, v in ck["state_dict"].items()}
                pfx = "model."
            elif cfgm.NAME == "InterGen":
                sd = {k.replace("model.", "") if "model" in k else k: v for k, v in ck["state_dict"].items()}
                pfx = "decoder.net."
            else:
                sd = ck
                pfx = "decoder.net_individual." if which == "1" else "decoder.net_interaction."
            with torch.no_grad():
                for k, v in sd.items():
                    if k.startswith(pfx):
                        tgt = "mixing.denoiser%s.%s" % (which, k[len(pfx):])
                        if tgt in own and tuple(own[tgt].shape) == tuple(v.shape):
                            own[tgt].copy_(v)
            self._dirty = True

    def set_norm_stats(self, mean_hml, std_hml, mean_ih, std_ih):
        self._stats = [np.asarray(a, dtype=np.float32).reshape(262) for a in (mean_hml, std_hml, mean_ih, std_ih)]
        self._dirty = True

    def init_synthetic(self, seed=0, std=0.02, bias_std=0.0, stats_seed=3):
        """Random-init weights of the reference architecture + synthetic normaliser stats (no checkpoints offline)."""
        sd = synthetic_state_dict(seed=seed, std=std, bias_std=bias_std, mixing_mode=self.mixing_mode, model1=self._model1, **self.dims)
        self.load_state_dict({"mixing." + k: v for k, v in sd.items()})
        st = synthetic_stats(stats_seed)
        self.set_norm_stats(st["mean_hml"], st["std_hml"], st["mean_ih"], st["std_ih"])
        return self

    def load_state_dict(self, state_dict, strict=True):
        own = {k for k, _ in self.named_parameters()}
        mine, unexpected = {}, []
        for k, v in state_dict.items():
            if k in own:
                mine[k] = v
            elif k.endswith("sequence_pos_encoder.pe") or k.startswith(OFF_PATH_PREFIXES):
                continue                       # regenerated tables / off-path modules of the reference's checkpoint
            else:
                unexpected.append(k)
        missing = sorted(own - set(mine))
        if strict and (missing or unexpected):
            raise RuntimeError("Error(s) in loading state_dict for MixerMDM:\n\tMissing key(s): %s\n\tUnexpected key(s): %s" % (missing[:8], unexpected[:8]))
        params = dict(self.named_parameters())
        with torch.no_grad():
            for k, v in mine.items():
                if tuple(v.shape) != tuple(params[k].shape):
                    raise RuntimeError(f"size mismatch for {k}: copying a param with shape {tuple(v.shape)}, the shape in current model is {tuple(params[k].shape)}")
                params[k].copy_(v)
        self._dirty = True
        # text-conditioning stage: when the checkpoint carries the CLIP tower and the three clipTransEncoder heads
        # (mixermdm.py:212-259), generate_cond runs on the GPU too; built lazily on the model's device
        need = ("token_embedding.weight", "positional_embedding", "ln_final.weight", "clipTransEncoder.layers.0.linear1.weight", "clip_ln.weight")
        if all(k in state_dict for k in need):
            pre = ("token_embedding.", "positional_embedding", "clip_transformer.", "ln_final.", "clipTransEncoder.", "clip_ln.",
                   "model1.clipTransEncoder", "model1.clip_ln", "model2.clipTransEncoder", "model2.clip_ln", "model1.clip_model.", "model1.embed_text.")
            self._text_sd = {k: v for k, v in state_dict.items() if k.startswith(pre)}
            self._text_enc = None
        return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)

    def _apply(self, fn, recurse=True):
        self._dirty = True
        return super()._apply(fn, recurse)

    @property
    def device(self):
        return next(self.parameters()).device

    def _sampler_for(self, B, T, cfg_scale=None):
        if cfg_scale is not None and float(cfg_scale) != float(self.cfg_mixing_weight):
            self.cfg_mixing_weight = cfg_scale         # the guidance scale is a handle constant (fused into the blend kernel)
        dev = self.device
        if dev.type != "cuda":
            raise RuntimeError("MixerMDM runs on an MI355X only: call .to('cuda:N') first (no CPU path)")
        m = self.mixing
        key = (str(dev), m.mixing_mode, bool(m.align), m.force_influence_val, self.cfg_model2.NAME, self.cfg_model1.NAME, self.precision,
               float(self.cfg_mixing_weight))
        s = self._sampler
        if s is None or s._key != key or B > s.cfg.max_batch or T > s.cfg.max_frames:
            if s is not None:
                s.close()
            s = Sampler(d_heads=self.d_heads, d1_heads=self.d1_heads, model1_kind=self.model1_kind, m_heads=self.m_heads,
                        mixing_mode=m.mixing_mode, align=m.align, xstart_align=True,
                        model2_kind=1 if self.cfg_model2.NAME == "InterGen" else 0, force_influence_val=m.force_influence_val,
                        cfg_scale=self.cfg_mixing_weight, max_batch=max(B, s.cfg.max_batch if s else 1),
                        max_frames=max(T, self.num_frames), device=dev, precision=self.precision, **self.dims)
            s._key = key
            self._sampler, self._dirty = s, True
        if self._dirty:
            if self._stats is None:
                raise RuntimeError("normaliser statistics missing: ./data/global_{mean,std}.npy and ./data/HumanML3D/{mean,std}_ih_new.npy "
                                   "not found (src/utils/utils.py:46-47,66-67); call set_norm_stats(...)")
            s.load_state_dict({k[len("mixing."):]: p.data for k, p in self.named_parameters()})
            s.set_norm_stats(*self._stats)
            s.prepare()
            s._strategy = None
            self._dirty = False
        return s

    # ---- conditioning -------------------------------------------------------------------------------
    def generate_cond(self, batch):
        if "cond" in batch:
            return batch["cond"]
        if self.text_encoder is None and getattr(self, "_text_sd", None) is not None:
            if self._text_enc is None or self._text_enc.tower.table.device != self.device:
                from .text import MixerTextEncoder
                heads = self._text_sd["clip_transformer.resblocks.0.attn.in_proj_weight"].shape[1] // 64 if "clip_transformer.resblocks.0.attn.in_proj_weight" in self._text_sd else 12
                self._text_enc = MixerTextEncoder(self._text_sd, clip_heads=heads, head_heads=8, device=self.device, model2=self.cfg_model2.NAME,
                                                  model1=self.cfg_model1.NAME)
            return self._text_enc.generate_cond(batch)
        if self.text_encoder is None:
            raise NotImplementedError("text encoding (CLIP ViT-L/14 tower + clipTransEncoder, mixermdm.py:283-356) is upstream of the HIP path: "
                                      "pass batch['cond'] [B, 8*768] or set model.text_encoder")
        return self.text_encoder(batch)

    # ---- sampling -----------------------------------------------------------------------------------
    def _run_loop(self, diffusion, cond, x_T, cfg_scale=None):
        B, T = x_T.shape[:2]
        smp = self._sampler_for(B, T, cfg_scale=cfg_scale)
        sch = diffusion.schedule
        # the facade builds a new MixerDiffusion per call (as the reference does, mixermdm.py:515-522): compare schedules by CONTENT so
        # that the tables are uploaded, the time-embedding tables rebuilt and (through the (B, T, S) graph cache) nothing re-captured
        # only when the strategy really changed
        if getattr(smp, "_strategy", None) != sch.key():
            import ctypes as C
            from ._lib import check
            tmap = np.ascontiguousarray(np.array(sch.timestep_map, dtype=np.int32))
            coef = np.ascontiguousarray(sch.device_coefficients())
            with torch.cuda.device(smp.device):
                check(smp.lib.mmdm_set_schedule(smp.h, tmap.ctypes.data_as(C.c_void_p), coef.ctypes.data_as(C.c_void_p), sch.num_timesteps, smp._s()), smp.h)
            smp.schedule, smp._strategy = sch, sch.key()
        m = self.mixing
        names = []
        if m.store_influence:
            names += ["influence_i1", "influence_i2"]
        if m.mode == "eval":
            names += ["out1", "out2", "out_influenced"]
        slots = (sch.num_timesteps + self.history_every - 1) // self.history_every
        wi = 262 if m.mixing_mode >= 3 else 1          # modes 1-2 keep the un-expanded [2B, T, 1] influence (mixermdm.py:739-745)
        need = slots * 2 * B * T * 4 * sum(wi if n.startswith("influence") else 524 for n in names)
        if need > HISTORY_BUDGET_BYTES:
            raise MemoryError(f"history side outputs need {need / 2**30:.1f} GiB for {sch.num_timesteps} steps (the reference keeps every step: "
                              "mixermdm.py:794-808); set model.history_every = k to keep every k-th step, or store_influence=False / forward_test")
        smp.begin(cond, x_T)
        hist = smp.set_history(names, self.history_every) if names else {}
        smp.run(None, use_graph=True)
        out = smp.state()["pred_xstart2"].clone()
        torch.cuda.current_stream(smp.device).wait_stream(smp.stream)
        as_list = lambda n: list(hist[n].unbind(0)) if n in hist else []
        m.history_influence_i1, m.history_influence_i2 = as_list("influence_i1"), as_list("influence_i2")
        m.history_out1, m.history_out2, m.history_out_influenced = as_list("out1"), as_list("out2"), as_list("out_influenced")
        return out

    # ---- many calls at once: the reference's callers, redesigned --------------------------------------------------
    def _set_schedule_on(self, smp, sch):
        if getattr(smp, "_strategy", None) != sch.key():
            import ctypes as C
            from ._lib import check
            tmap = np.ascontiguousarray(np.array(sch.timestep_map, dtype=np.int32))
            coef = np.ascontiguousarray(sch.device_coefficients())
            with torch.cuda.device(smp.device):
                check(smp.lib.mmdm_set_schedule(smp.h, tmap.ctypes.data_as(C.c_void_p), coef.ctypes.data_as(C.c_void_p), sch.num_timesteps, smp._s()), smp.h)
            smp.schedule, smp._strategy = sch, sch.key()

    def _pool(self, k, B, T):
        """k samplers over ONE weight set: the facade's own handle plus k - 1 handles that borrow its weights (mmdm_create_shared), each with its
        own stream, workspace (B, T), schedule tables and graph cache."""
        main = self._sampler_for(B, T)
        pool = getattr(self, "_shared", None)
        if pool is None or pool["parent"] is not main or pool["B"] < B or pool["T"] < T:
            if pool is not None:
                for c in pool["kids"]:
                    c.close()
            pool = {"parent": main, "B": B, "T": T, "kids": []}
            self._shared = pool
        while len(pool["kids"]) < k - 1:
            pool["kids"].append(main.share(max_batch=B, max_frames=max(T, self.num_frames)))
        return [main] + pool["kids"][:k - 1]

    def sample_many(self, batches, mode="eval_intermediate", batching="ragged", inflight=2, max_rows=4800, max_items=64, keep_history=None):
        """Many sampling calls in one go -- what the reference's callers do one after the other: the inference script calls ``model(batch)`` ten
        times at B = 1 (src/scripts/infer/mixermdm.py:184-188), the evaluation datasets call ``forward_test`` once per item with that item's
        own length (src/evaluation/datasets.py:100-116).  `batches`: reference-style batch dicts (text lists or a precomputed 'cond' [b, 8*768];
        'motion_lens'[0] = T; optionally 'x_T' [b, T, 524]).  Returns one result dict per batch, as ``forward`` (mode="eval") /
        ``forward_test`` (mode="eval_intermediate") would -- every motion BIT-IDENTICAL to that call's on the same x_T.

        batching: "sequential" = the reference's loop (one call after the other on one handle);
                  "inflight"   = `inflight` handles over one weight set, calls dealt round-robin, each on its own stream.  Bit-identical, and
                                 measured worth 1.00-1.03 x (the GPU co-schedules two such streams hardly at all: LAB_NOTES.md round 5) --
                                 kept for callers whose requests arrive at different times, not as a throughput lever;
                  "ragged"     = the calls' motions packed into ragged batches of <= max_rows frames / <= max_items motions
                                 (mmdm_begin_ragged: per-sequence lengths as device data; GEMMs at the efficiency of a full batch).
        keep_history: None = what the mode says (store_influence -> influence lists; "eval" -> out1 / out2 / out_influenced), False = outputs only."""
        if batching not in ("sequential", "inflight", "ragged"):
            raise ValueError(f"batching {batching!r} not recognized")
        m = self.mixing
        m.mode = mode
        names = []
        if keep_history is None or keep_history:
            if m.store_influence:
                names += ["influence_i1", "influence_i2"]
            if mode == "eval":
                names += ["out1", "out2", "out_influenced"]
        with torch.no_grad():
            conds = [self.generate_cond(b) for b in batches]
        Ts = [int(b["motion_lens"][0]) for b in batches]
        Bs = [int(c.shape[0]) for c in conds]
        dev = self.device
        xs = [b["x_T"].to(dev, torch.float32) if "x_T" in b else torch.randn(nb, T, self.nfeats * 2, device=dev) for b, nb, T in zip(batches, Bs, Ts)]
        sch = MixerDiffusion(use_timesteps=space_timesteps(self.diffusion_steps, self.sampling_strategy), betas=self.betas).schedule
        slots = (sch.num_timesteps + self.history_every - 1) // self.history_every
        keys = ("influence_i1", "influence_i2") + (("out1", "out2", "out_influenced") if mode == "eval" else ())
        results = [{"output": None, **{k: [] for k in keys}} for _ in batches]
        if batching == "sequential":
            for i, (b, c, x) in enumerate(zip(batches, conds, xs)):
                bb = dict(b)
                bb["cond"], bb["x_T"] = c, x
                keep = m.store_influence
                if not names:
                    m.store_influence = False
                try:
                    results[i] = self.forward(bb) if mode == "eval" else self.forward_test(bb)
                finally:
                    m.store_influence = keep
            return results
        if batching == "inflight":
            pool = self._pool(max(1, int(inflight)), max(Bs), max(Ts))
            for smp in pool:
                self._set_schedule_on(smp, sch)
            # ONE host thread drives every handle (calls dealt round-robin, nothing synchronised in between).  One thread per handle was built
            # and measured (tools/inflight_probe.py: 4 handles 3.65 vs 4.27 ms per item-step) and is not used: graph launches from two host
            # threads crash inside this runtime's hipGraphLaunch (hip::Graph::UpdateStreams) as soon as the threads also capture new shapes,
            # whatever the locking around captures (LAB_NOTES.md, round 5).  Buffers are allocated up front; the calls are C-ABI calls only.
            conds = [c.to(dev, torch.float32).contiguous() for c in conds]
            xs = [x.contiguous() for x in xs]
            wi = 262 if m.mixing_mode >= 3 else 1
            per_row = slots * 2 * 4 * sum(wi if n.startswith("influence") else 524 for n in names)      # history bytes per (motion, frame)
            need = per_row * sum(nb * T for nb, T in zip(Bs, Ts))
            # the history side outputs of EVERY call are the caller's to keep (as in the sequential and ragged paths): the same budget, summed
            if need > HISTORY_BUDGET_BYTES:
                raise MemoryError(f"history side outputs of {len(Bs)} in-flight calls need {need / 2**30:.1f} GiB; pass fewer batches per call, set "
                                  "model.history_every, or pass keep_history=False")
            cur = torch.cuda.current_stream(dev)
            # calls are queued in windows of a few multiples of `inflight`: output and history buffers are allocated per window (the caching
            # allocator is not touched while a window's calls are being queued), so lazily built inputs of later windows are not needed yet
            win = 4 * len(pool)
            for w0 in range(0, len(conds), win):
                idx = range(w0, min(w0 + win, len(conds)))
                outs = {i: torch.empty_like(xs[i]) for i in idx}
                hists = {i: {nm: torch.empty(slots, 2 * Bs[i], Ts[i], (wi if nm.startswith("influence") else 524), device=dev) for nm in names} for i in idx}
                for smp in pool:                     # inputs and buffers were produced on the caller's stream: order every handle's stream behind it
                    smp.stream.wait_stream(cur)
                for i in idx:
                    pool[i % len(pool)].enqueue(conds[i], xs[i], outs[i], hists[i], self.history_every)
                for smp in pool:
                    cur.wait_stream(smp.stream)
                    smp.stream.synchronize()
                for i in idx:
                    results[i]["output"] = outs[i]
                    for k, v in hists[i].items():
                        results[i][k] = list(v.unbind(0))
            return results
        # ragged: motions in call order, cut into groups of <= max_rows frames and <= max_items motions
        motions = [(i, j) for i, nb in enumerate(Bs) for j in range(nb)]
        groups, cur, rows = [], [], 0
        for (i, j) in motions:
            if cur and (rows + Ts[i] > max_rows or len(cur) >= max_items):
                groups.append(cur)
                cur, rows = [], 0
            cur.append((i, j))
            rows += Ts[i]
        if cur:
            groups.append(cur)
        gmax_items = max(len(g) for g in groups)
        gmax_rows = max(sum(Ts[i] for i, _ in g) for g in groups)
        Tm = max(max(Ts), self.num_frames)
        smp = self._sampler_for(max(gmax_items, -(-gmax_rows // Tm)), Tm)
        self._set_schedule_on(smp, sch)
        wi = 262 if m.mixing_mode >= 3 else 1
        outs = [[None] * nb for nb in Bs]
        hists = [[None] * nb for nb in Bs]
        pend = []
        for g in groups:
            lens = [Ts[i] for i, _ in g]
            need = slots * 2 * (sum(lens) + 128) * 4 * sum(wi if n.startswith("influence") else 524 for n in names)
            if need > HISTORY_BUDGET_BYTES:
                raise MemoryError(f"history side outputs of a ragged batch of {sum(lens)} frames need {need / 2**30:.1f} GiB; lower max_rows, set "
                                  "model.history_every, or pass keep_history=False")
            cond_g = torch.cat([conds[i][j:j + 1] for i, j in g], 0)
            x_g = [xs[i][j] for i, j in g]
            items, hist, ev = smp.sample_ragged_async(cond_g, x_g, lens, history=names or None, history_every=self.history_every)
            pend.append((g, items, hist, ev, smp.item_slices(), smp.rows))
        for g, items, hist, ev, slices, rows in pend:
            ev.synchronize()
            for (i, j), it, (o, t) in zip(g, items, slices):
                outs[i][j] = it
                if hist:
                    # the reference's [2B, T, C] history rows of this motion: its cond row and its uncond row
                    hists[i][j] = {k: (v[:, 0, o:o + t], v[:, 1, o:o + t]) for k, v in hist.items()}
        torch.cuda.current_stream(dev).wait_stream(smp.stream)
        for i, nb in enumerate(Bs):
            results[i]["output"] = torch.stack(outs[i], 0)
            if names:
                for k in names:
                    cond_rows = torch.stack([hists[i][j][k][0] for j in range(nb)], 1)      # [slots, b, T, C]
                    unc_rows = torch.stack([hists[i][j][k][1] for j in range(nb)], 1)
                    results[i][k] = list(torch.cat([cond_rows, unc_rows], 1).unbind(0))     # [2b, T, C] per kept step
        return results

    def _sample(self, batch, mode):
        self.mixing.mode = mode
        cond = self.generate_cond(batch)
        B = cond.shape[0]
        T = int(batch["motion_lens"][0])
        self.diffusion_test = MixerDiffusion(use_timesteps=space_timesteps(self.diffusion_steps, self.sampling_strategy), betas=self.betas)
        self.cfg_model = ClassifierFreeSampleModelX2(self.mixing, self.cfg_mixing_weight)
        return self.diffusion_test.ddim_sample_loop(self.cfg_model, (B, T, self.nfeats * 2), noise=batch.get("x_T"), clip_denoised=False,
                                                    progress=True, model_kwargs={"mask": None, "cond": cond}, x_start=None)

    def forward(self, batch):
        """mixermdm.py:490-548."""
        output = self._sample(batch, "eval")
        m = self.mixing
        return {"output": output, "influence_i1": m.history_influence_i1, "influence_i2": m.history_influence_i2,
                "out1": m.history_out1, "out2": m.history_out2, "out_influenced": m.history_out_influenced}

    def forward_test(self, batch):
        """mixermdm.py:550-602."""
        output = self._sample(batch, "eval_intermediate")
        m = self.mixing
        return {"output": output, "influence_i1": m.history_influence_i1, "influence_i2": m.history_influence_i2}


class in2INDiffusion(nn.Module):
    """Stand-alone sub-model sampler: in2INDiffusion.forward (src/models/in2in.py:285-356), modes "individual"
    (ClassifierFreeSampleModel, [B,T,262]), "interaction" (ClassifierFreeSampleModelMultiple, [B,T,524]) and "dual"
    (ClassifierFreeSampleDualMDM over net_individual + net_interaction, [B,T,524]); output stays in the model's normalised
    space, as in the reference (callers de-normalise: src/scripts/infer/in2IN.py:101-105)."""

    def __init__(self, cfg, mode, sampling_strategy="ddim50"):
        super().__init__()
        if mode not in ("individual", "interaction", "dual"):
            raise ValueError(f"in2IN mode {mode!r} not recognized")
        self.cfg, self.mode = cfg, mode
        self.nfeats = cfg.INPUT_DIM
        self.dims = dict(d_latent=cfg.LATENT_DIM, d_ff=cfg.FF_SIZE, d_layers=cfg.NUM_LAYERS)
        self.num_heads = cfg.NUM_HEADS
        self.cfg_weight = cfg.CFG_WEIGHT if "CFG_WEIGHT" in cfg else 0.0
        self.cfg_weight_interaction = cfg.CFG_WEIGHT_INTERACTION if "CFG_WEIGHT_INTERACTION" in cfg else 0.0
        self.cfg_weight_individual = cfg.CFG_WEIGHT_INDIVIDUAL if "CFG_WEIGHT_INDIVIDUAL" in cfg else 0.0
        if mode == "dual":                          # in2in.py:158-162
            self.cfg_composition_weight_func, self.cfg_composition_weight_value = cfg.W_FUNC, cfg.W_VALUE
        self.diffusion_steps = cfg.DIFFUSION_STEPS
        self.betas = get_named_beta_schedule(cfg.BETA_SCHEDULER, self.diffusion_steps)
        self.sampling_strategy = sampling_strategy
        self._nets = {"individual": [("net_individual", "denoiser1.")], "interaction": [("net_interaction", "denoiser2.")],
                      "dual": [("net_individual", "denoiser1."), ("net_interaction", "denoiser2.")]}[mode]
        from .synthetic import denoiser_shapes
        for net, _ in self._nets:
            for k, shp in denoiser_shapes(net + ".", self.dims["d_latent"], self.dims["d_ff"], self.dims["d_layers"]).items():
                _holder_tree(self, k, torch.zeros(shp))
        self._sampler, self._dirty = None, True

    def _apply(self, fn, recurse=True):
        self._dirty = True
        return super()._apply(fn, recurse)

    def load_state_dict(self, state_dict, strict=True):
        sd = {k: v for k, v in state_dict.items() if not k.endswith("sequence_pos_encoder.pe")}
        self._dirty = True
        return super().load_state_dict(sd, strict=strict)

    def _get_sampler(self, B, T):
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("in2IN runs on an MI355X only: call .to('cuda:N') first (no CPU path)")
        s = self._sampler
        if s is None or B > s.cfg.max_batch or T > s.cfg.max_frames or s.device != dev:
            if s is not None:
                s.close()
            kind = {"individual": 1, "interaction": 2, "dual": 3}[self.mode]
            s = Sampler(d_heads=self.num_heads, single_only=kind, cfg_scale=self.cfg_weight, cfg_scale_interaction=self.cfg_weight_interaction,
                        cfg_scale_individual=self.cfg_weight_individual, max_batch=B, max_frames=max(T, 300), device=dev, **self.dims)
            self._sampler, self._dirty = s, True
        if self._dirty:
            sd = {}
            for net, pfx in self._nets:
                sd.update({pfx + k[len(net) + 1:]: p.data for k, p in self.named_parameters() if k.startswith(net + ".")})
            s.load_state_dict(sd)
            s.prepare()
            s._strategy = None
            self._dirty = False
        return s

    def forward(self, batch):
        if self.mode == "dual":                      # in2in.py:299-305
            cond = torch.cat([batch["cond_interaction"], batch["cond_interaction_individual1"], batch["cond_interaction_individual2"],
                              batch["cond_individual_individual1"], batch["cond_individual_individual2"]], dim=1)
        elif self.mode == "interaction":
            cond = torch.cat([batch["cond_interaction"], batch["cond_interaction_individual1"], batch["cond_interaction_individual2"]], dim=1)
        else:
            cond = torch.cat([batch["cond_individual_individual1"]], dim=1)
        B, T = cond.shape[0], int(batch["motion_lens"][0])
        s = self._get_sampler(B, T)
        if s._strategy != self.sampling_strategy:
            s.set_schedule(self.sampling_strategy, self.cfg.BETA_SCHEDULER, self.diffusion_steps)
            if self.mode == "dual":
                s.set_dual_weights(self.cfg_composition_weight_func, self.cfg_composition_weight_value)
            s._strategy = self.sampling_strategy
        width = self.nfeats * (1 if self.mode == "individual" else 2)
        x_T = batch["x_T"] if "x_T" in batch else torch.randn(B, T, width, device=s.device)
        return {"output": s.sample(cond, x_T)}


class in2IN(nn.Module):
    """in2IN(cfg, mode) facade (src/models/in2in.py:14-135) around the stand-alone sampler.  Text conditioning: when the loaded checkpoint
    carries the CLIP tower and the clipTransEncoder_{individual,interaction} heads (in2in.py:24-69), ``text_process`` runs them on the GPU
    (mixermdm_amd.text); otherwise pass the encoded ``cond_*`` entries in the batch or register ``text_encoder(batch, mode, text_name, out_name)``."""

    def __init__(self, cfg, mode):
        super().__init__()
        self.cfg, self.mode = cfg, mode
        self.decoder = in2INDiffusion(cfg, mode, sampling_strategy=cfg.STRATEGY)
        self.text_encoder = None
        self.text_num_heads = 8               # nhead of the clipTransEncoder layers (hard-coded in the reference: in2in.py:27, 43)
        self._text_sd, self._text = None, None

    def load_state_dict(self, state_dict, strict=True):
        """in2IN checkpoint (raw state dict, in2in.py keys): ``decoder.*`` -> the sampler; tower + text heads -> the GPU text stage."""
        dec = {k[len("decoder."):]: v for k, v in state_dict.items() if k.startswith("decoder.")}
        res = self.decoder.load_state_dict(dec, strict=strict)
        pre = ("token_embedding.", "positional_embedding", "clip_transformer.", "ln_final.", "clipTransEncoder_", "clip_ln_")
        txt = {k: v for k, v in state_dict.items() if k.startswith(pre)}
        if "token_embedding.weight" in txt:
            self._text_sd, self._text = txt, None
        return res

    def _text_stage(self, mode):
        from .text import ClipTextTower, TextHead
        dev = next(self.decoder.parameters()).device
        if self._text is None or self._text["tower"].table.device != dev:
            sd = self._text_sd
            heads = sd["clip_transformer.resblocks.0.attn.in_proj_weight"].shape[1] // 64 if "clip_transformer.resblocks.0.attn.in_proj_weight" in sd else 12
            self._text = {"tower": ClipTextTower(sd, "", heads, dev)}
            for m in ("individual", "interaction"):
                if f"clip_ln_{m}.weight" in sd:
                    self._text[m] = TextHead(sd, f"clipTransEncoder_{m}.", f"clip_ln_{m}", self.text_num_heads, dev)
        if mode not in self._text:
            raise ValueError("Mode not recognized")
        return self._text["tower"], self._text[mode]

    def text_process(self, batch, mode=None, text_name="text", out_name="cond"):
        if out_name in batch:
            return batch
        if self.text_encoder is not None:
            return self.text_encoder(batch, mode, text_name, out_name)
        if self._text_sd is None:
            raise NotImplementedError("text encoding (CLIP tower + clipTransEncoder, in2in.py:109-135) is upstream of the HIP path: "
                                      f"put the encoded '{out_name}' in the batch, load a checkpoint that carries the text modules, or set model.text_encoder")
        from .text import tokenize
        tower, head = self._text_stage(mode)
        tok = torch.as_tensor(batch["tokens_" + text_name]) if "tokens_" + text_name in batch else tokenize(batch[text_name])
        batch[out_name] = head(tower(tok), tok)
        return batch

    def decode_motion(self, batch):
        batch.update(self.decoder(batch))
        return batch

    def forward_test(self, batch):
        if self.mode in ("interaction", "dual"):
            batch = self.text_process(batch, "interaction", out_name="cond_interaction")
            batch = self.text_process(batch, "interaction", "text_individual1", "cond_interaction_individual1")
            batch = self.text_process(batch, "interaction", "text_individual2", "cond_interaction_individual2")
        if self.mode == "dual":
            batch = self.text_process(batch, "individual", "text_individual1", "cond_individual_individual1")
            batch = self.text_process(batch, "individual", "text_individual2", "cond_individual_individual2")
        elif self.mode == "individual":
            batch = self.text_process(batch, "individual", out_name="cond_individual_individual1")
        batch.update(self.decode_motion(batch))
        return batch
