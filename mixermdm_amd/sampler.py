"""Python owner of an mmdm_handle (include/mmdm.h section 2): weights, schedule, the DDIM host loop.

The loop itself is one C call per chunk of steps (``mmdm_run``), optionally replaying a captured hipGraph; Python only
decides how many steps to run and where the history buffers live.
"""
import ctypes as C
import numpy as np
import torch

from ._lib import Config, load_library, check
from .schedule import make_schedule

STATS_ORDER = ("mean_hml", "std_hml", "mean_ih", "std_ih")


def pe_table(d_model, max_len=5000):
    """PositionalEncoding buffer, generated exactly like the reference (src/models/utils/utils.py:24-35) with torch on the
    host, then handed to the library as the ``sequence_pos_encoder.pe`` entry of the state dict."""
    pe = torch.zeros(max_len, d_model)
    position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-np.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe


class Sampler:
    """One handle = one device, one (max_batch, max_frames) workspace."""

    def __init__(self, *, d_latent, d_ff, d_layers, d_heads, m_latent=0, m_ff=0, m_layers=0, m_heads=1, mixing_mode=4, align=True,
                 xstart_align=True, model2_kind=0, force_influence_val=None, cfg_scale=3.5, max_batch=1, max_frames=300,
                 single_only=False, text_dim=768, device=None, cfg_scale_interaction=0.0, cfg_scale_individual=0.0, precision="fp32",
                 model1_kind=0, d1_latent=0, d1_ff=0, d1_layers=0, d1_heads=0):
        """single_only: False/0 = two-chain MixerMDM; True/1 = individual denoiser alone (2-way CFG);
        2 = interaction denoiser alone with the 4-way CFG of ClassifierFreeSampleModelMultiple;
        3 = in2IN "dual": both denoisers composed by ClassifierFreeSampleDualMDM (call set_dual_weights after set_schedule).
        model1_kind: 0 = in2IN individual, 1 = MDMDenoiser; d1_*: denoiser1's own sizes (0 = same as d_*).
        precision: "fp32" (native fp32 MFMA), "fp32_split" (fp32 results from six bf16 MFMAs per product on exactly split operands:
        same accuracy, the bf16 matrix rate), "bf16" (bf16 GEMM operands), "bf16_fp8" (BASELINE configs[4]: as "bf16" with the QKV /
        cross-attention input projections and both FFN GEMMs on fp8 e4m3 operands)."""
        if not torch.cuda.is_available():
            raise RuntimeError("mixermdm_amd.Sampler needs an MI355X (HIP device); there is no CPU path")
        self.lib = load_library()
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.cfg = Config(d_latent, d_ff, d_layers, d_heads, m_latent, m_ff, m_layers, m_heads, 262, text_dim, mixing_mode, int(align),
                          int(xstart_align), model2_kind, int(force_influence_val is not None), float(force_influence_val or 0.0),
                          float(cfg_scale), max_batch, max_frames, int(single_only), float(cfg_scale_interaction), float(cfg_scale_individual),
                          {"fp32": 0, "bf16": 1, "fp32_split": 2, "bf16_fp8": 3}[precision], int(model1_kind), d1_latent, d1_ff, d1_layers, d1_heads)
        self.single_only = int(single_only)
        self.h = C.c_void_p()
        with torch.cuda.device(self.device):
            check(self.lib.mmdm_create(C.byref(self.cfg), C.byref(self.h)))
            # graph capture is not allowed on the legacy default stream: the handle works on its own stream
            self.stream = torch.cuda.Stream(device=self.device)
        self.schedule = None
        self._strategy = None
        self._hist = None
        self._parent = None
        self.lens, self.rows = None, 0

    def share(self, max_batch=None, max_frames=None):
        """A second Sampler over THIS sampler's weights (mmdm_create_shared): own workspace, stream, schedule tables and graph cache, one
        copy of the parameters.  K shared samplers keep K sampling calls in flight on K streams -- the reference's callers sample one
        item at a time (src/scripts/infer/mixermdm.py:184-188, src/evaluation/datasets.py:100-116), which leaves the GPU 40 % empty at B = 1.
        The weights must be loaded and prepared on `self`; load_state_dict / set_norm_stats / prepare are the parent's."""
        s = object.__new__(Sampler)
        s.lib, s.device, s.single_only = self.lib, self.device, self.single_only
        cfg = Config.from_buffer_copy(self.cfg)
        cfg.max_batch = int(max_batch or self.cfg.max_batch)
        cfg.max_frames = int(max_frames or self.cfg.max_frames)
        s.cfg = cfg
        s.h = C.c_void_p()
        with torch.cuda.device(self.device):
            check(self.lib.mmdm_create_shared(self.h, cfg.max_batch, cfg.max_frames, C.byref(s.h)), self.h)
            s.stream = torch.cuda.Stream(device=self.device)
        s.schedule, s._strategy, s._hist, s._parent = None, None, None, self
        s.lens, s.rows = None, 0
        return s

    def close(self):
        """Frees the handle (mmdm_destroy selects the handle's own device before synchronising and freeing).  Tensors returned by
        state() are views of handle memory and are invalid afterwards."""
        if getattr(self, "h", None) and self.h.value:
            self.lib.mmdm_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _s(self):
        return C.c_void_p(self.stream.cuda_stream)

    # ---- weights ----------------------------------------------------------------------------------
    def load_state_dict(self, sd):
        """sd: Mixer state_dict (reference key names; src/models/mixermdm.py:134-148); strict: unknown keys raise here, missing
        keys raise in prepare().  pe buffers are regenerated if absent."""
        sd = dict(sd)
        D, Dm = self.cfg.d_latent, self.cfg.m_latent
        pes = []
        if self.single_only != 2:
            pes.append(("denoiser1.sequence_pos_encoder.pe", self.cfg.d1_latent or D))
        if self.single_only != 1:
            pes.append(("denoiser2.sequence_pos_encoder.pe", D))
        if self.single_only == 0:
            pes.append(("sequence_pos_encoder.pe", Dm))
        for k, d in pes:
            if k not in sd:
                sd[k] = pe_table(d)
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        keep = []
        with torch.cuda.device(self.device):
            for k, v in sd.items():
                t = v.detach().to(device=self.device, dtype=torch.float32).contiguous()
                keep.append(t)
                rows, cols = (t.shape[0], 1) if t.dim() == 1 else (t.shape[0], t.shape[1])
                check(self.lib.mmdm_set_weight(self.h, k.encode(), C.c_void_p(t.data_ptr()), rows, cols, self._s()), self.h)
            self.stream.synchronize()
        return self

    def set_norm_stats(self, mean_hml, std_hml, mean_ih, std_ih):
        arr = np.concatenate([np.asarray(a, dtype=np.float32).reshape(262) for a in (mean_hml, std_hml, mean_ih, std_ih)])
        arr = np.ascontiguousarray(arr)
        with torch.cuda.device(self.device):
            check(self.lib.mmdm_set_norm_stats(self.h, arr.ctypes.data_as(C.c_void_p)), self.h)
        return self

    def prepare(self):
        with torch.cuda.device(self.device):
            check(self.lib.mmdm_prepare(self.h), self.h)
        return self

    # ---- schedule ---------------------------------------------------------------------------------
    def set_schedule(self, sampling_strategy="ddim50", beta_scheduler="cosine", diffusion_steps=1000):
        sch = make_schedule(beta_scheduler, diffusion_steps, sampling_strategy)
        tmap = np.ascontiguousarray(np.array(sch.timestep_map, dtype=np.int32))
        coef = np.ascontiguousarray(sch.device_coefficients())
        with torch.cuda.device(self.device):
            check(self.lib.mmdm_set_schedule(self.h, tmap.ctypes.data_as(C.c_void_p), coef.ctypes.data_as(C.c_void_p), sch.num_timesteps, self._s()), self.h)
        self.schedule = sch
        self._strategy = sch.key()
        return sch

    def set_dual_weights(self, func, value):
        """Composition weight of ClassifierFreeSampleDualMDM (cfg_sampler.py:113-125) evaluated in float64 on the model-side
        timestep of every respaced step, exactly as the reference's lambdas do, then handed over as a float32 table."""
        t = np.asarray(self.schedule.timestep_map, dtype=np.int64)
        if func == "exp":
            w = np.exp(-value * (1000 - t))
        elif func == "exp-inv":
            w = 1 - np.exp(-value * (1000 - t))
        elif func == "lin":
            w = 1 - ((1000 - t) / 1000)
        elif func == "const":
            w = np.full(t.shape, value, dtype=np.float64)
        else:
            raise ValueError("Unknown function")
        w = np.ascontiguousarray(w.astype(np.float32))
        with torch.cuda.device(self.device):
            check(self.lib.mmdm_set_dual_weights(self.h, w.ctypes.data_as(C.c_void_p), len(w)), self.h)
        return w

    # ---- sampling ---------------------------------------------------------------------------------
    def begin(self, cond, x_T):
        cond = cond.to(self.device, torch.float32).contiguous()
        x_T = x_T.to(self.device, torch.float32).contiguous()
        B, T = x_T.shape[:2]
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        # the copies into the handle run on the sampler's stream, possibly long after this call returns (several calls in flight):
        # keep the allocator from handing the inputs' memory out again before that stream has passed this point
        cond.record_stream(self.stream)
        x_T.record_stream(self.stream)
        with torch.cuda.device(self.device):
            check(self.lib.mmdm_begin(self.h, C.c_void_p(cond.data_ptr()), C.c_void_p(x_T.data_ptr()), B, T, self._s()), self.h)
        self._keep = (cond, x_T)
        self.B, self.T = B, T
        self.lens, self.rows = None, B * T
        self._hist = None
        return self

    def begin_ragged(self, cond, x_T, lens):
        """Begin a RAGGED call (mmdm_begin_ragged): B items of different lengths in one batch.  cond [B, .]; lens: B ints; x_T: the items' frames
        back to back [sum(lens), C], or a list of B tensors [T_i, C].  Every item's result is bit-identical to sampling it alone."""
        lens = [int(v) for v in lens]
        if isinstance(x_T, (list, tuple)):
            x_T = torch.cat([t.to(self.device, torch.float32).reshape(-1, t.shape[-1]) for t in x_T], 0)
        cond = cond.to(self.device, torch.float32).contiguous()
        x_T = x_T.to(self.device, torch.float32).contiguous()
        B = len(lens)
        if cond.shape[0] != B or x_T.dim() != 2 or x_T.shape[0] != sum(lens):
            raise ValueError(f"begin_ragged: cond rows {cond.shape[0]}, x_T {tuple(x_T.shape)} do not match {B} items of {sum(lens)} frames in all")
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        cond.record_stream(self.stream)
        x_T.record_stream(self.stream)
        arr = (C.c_int * B)(*lens)
        with torch.cuda.device(self.device):
            check(self.lib.mmdm_begin_ragged(self.h, C.c_void_p(cond.data_ptr()), C.c_void_p(x_T.data_ptr()), B, arr, self._s()), self.h)
        rows, real, rag = C.c_int(), C.c_int(), C.c_int()
        check(self.lib.mmdm_call_rows(self.h, C.byref(rows), C.byref(real), C.byref(rag)), self.h)
        self._keep = (cond, x_T)
        self.B, self.T = B, max(lens)
        self.lens, self.rows = lens, rows.value
        self._hist = None
        return self

    def item_slices(self):
        """(first row, length) of every item inside a group of a ragged call's buffers."""
        out, o = [], 0
        for t in self.lens:
            out.append((o, t))
            o += t
        return out

    def set_history(self, names=("influence_i1", "influence_i2", "out1", "out2", "out_influenced"), every=1):
        """Allocate history buffers [slots, 2B, T, C] for the requested side outputs (mixermdm.py:794-796, 805-808).  C = 524 for
        out1 / out2 / out_influenced; for the influences 262 in mixing modes 3-4 and 1 in modes 1-2 (the reference's shapes).
        Ragged call: [slots, 2, rows, C] -- cond half, uncond half, each a group of the call's frame rows (item_slices())."""
        S = self.schedule.num_timesteps
        slots = (S + every - 1) // every
        n = 2 * self.B
        bufs = {}
        rag = getattr(self, "lens", None) is not None
        for nm in ("influence_i1", "influence_i2", "out1", "out2", "out_influenced"):
            if nm in names:
                Cc = (262 if self.cfg.mixing_mode >= 3 else 1) if nm.startswith("influence") else 524
                bufs[nm] = torch.empty((slots, 2, self.rows, Cc) if rag else (slots, n, self.T, Cc), device=self.device, dtype=torch.float32)
        for b in bufs.values():
            b.record_stream(self.stream)
        ptr = lambda nm: C.c_void_p(bufs[nm].data_ptr() if nm in bufs else 0)
        with torch.cuda.device(self.device):
            check(self.lib.mmdm_set_history(self.h, ptr("influence_i1"), ptr("influence_i2"), ptr("out1"), ptr("out2"), ptr("out_influenced"), every), self.h)
        self._hist = bufs
        return bufs

    def run(self, nsteps=None, use_graph=True):
        if nsteps is None:
            nsteps = self.schedule.num_timesteps
        with torch.cuda.device(self.device):
            check(self.lib.mmdm_run(self.h, nsteps, int(use_graph), self._s()), self.h)
        return self

    def seek(self, step_index):
        """Continue the begun call from respaced step `step_index` (S-1 = first, 0 = last) with the chains as they stand."""
        with torch.cuda.device(self.device):
            check(self.lib.mmdm_seek(self.h, int(step_index), self._s()), self.h)
        return self

    def synchronize(self):
        self.stream.synchronize()

    def state(self, sync=True):
        """Views (no copy) of the handle's x, x2, pred_xstart, pred_xstart2, model_out -- after the queued work completes (sync=True), or
        at once for use ON the sampler's stream (sync=False: sample_async)."""
        ptrs = [C.c_void_p() for _ in range(5)]
        check(self.lib.mmdm_get_state(self.h, *[C.byref(p) for p in ptrs]), self.h)
        if sync:
            self.stream.synchronize()
        Cc = 262 if self.single_only == 1 else 524
        shape = (self.rows, Cc) if getattr(self, "lens", None) is not None else (self.B, self.T, Cc)      # ragged: the group's frame rows (first sum(lens) real)
        out = {}
        for nm, p in zip(("x", "x2", "pred_xstart", "pred_xstart2", "model_out"), ptrs):
            out[nm] = _from_ptr(p.value, shape, self.device) if p.value else None
        return out

    def sample(self, cond, x_T, use_graph=True, history=None, history_every=1):
        """Full loop: MixerDiffusion.ddim_sample_loop (gaussian_diffusion.py:1769-1820) -> last pred_xstart2 (or pred_xstart, single)."""
        self.begin(cond, x_T)
        hist = self.set_history(history, history_every) if history else None
        self.run(None, use_graph)
        st = self.state()
        res = st["pred_xstart"] if self.single_only else st["pred_xstart2"]
        torch.cuda.current_stream(self.device).wait_stream(self.stream)
        return (res.clone(), hist) if history else res.clone()

    def sample_async(self, cond, x_T, use_graph=True, history=None, history_every=1):
        """sample() without a host synchronisation: the whole loop and the copy of its result are queued on the sampler's stream and the call
        returns at once -> (result tensor, history buffers or None, event recorded behind them).  Wait for the event (or synchronise the
        stream) before reading either from the host or another stream.  Several shared samplers (share()) driven this way overlap."""
        self.begin(cond, x_T)
        hist = self.set_history(history, history_every) if history else None
        self.run(None, use_graph)
        st = self.state(sync=False)
        with torch.cuda.stream(self.stream):
            res = (st["pred_xstart"] if self.single_only else st["pred_xstart2"]).clone()
            ev = torch.cuda.Event()
            ev.record(self.stream)
        return res, hist, ev

    def enqueue(self, cond, x_T, out, hist=None, history_every=1, use_graph=True):
        """A whole uniform sampling call through the C ABI ONLY (mmdm_begin, mmdm_set_history, mmdm_run, mmdm_copy_result on the sampler's
        stream): no tensor is created, no torch stream or allocator call is made -- safe to call from a worker thread while other threads
        capture step graphs (the caching allocator's event queries are not allowed beside a capture).  Everything is the caller's: cond
        [B, .] and x_T [B, T, C] contiguous fp32 on the device and ordered before the sampler's stream, `out` [B, T, C] preallocated, `hist`
        {name: preallocated [slots, 2B, T, C] buffer}; all must stay alive until the stream has passed the call."""
        B, T = x_T.shape[:2]
        h = hist or {}
        ptr = lambda nm: C.c_void_p(h[nm].data_ptr() if nm in h else 0)
        # the C side launches on whatever device is current: select the handle's (torch.cuda.device is a hipSetDevice pair, no allocator call)
        with torch.cuda.device(self.device):
            check(self.lib.mmdm_begin(self.h, C.c_void_p(cond.data_ptr()), C.c_void_p(x_T.data_ptr()), B, T, self._s()), self.h)
            if h:
                check(self.lib.mmdm_set_history(self.h, ptr("influence_i1"), ptr("influence_i2"), ptr("out1"), ptr("out2"), ptr("out_influenced"), history_every), self.h)
            check(self.lib.mmdm_run(self.h, self.schedule.num_timesteps, int(use_graph), self._s()), self.h)
            check(self.lib.mmdm_copy_result(self.h, C.c_void_p(out.data_ptr()), self._s()), self.h)
        self.B, self.T, self.lens, self.rows = B, T, None, B * T

    def sample_ragged_async(self, cond, x_T, lens, use_graph=True, history=None, history_every=1):
        """A whole ragged sampling call queued on the sampler's stream (no host synchronisation): -> (list of per-item results [T_i, C], history
        buffers [slots, 2, rows, C] or None, event).  Wait for the event before reading from the host or another stream."""
        self.begin_ragged(cond, x_T, lens)
        hist = self.set_history(history, history_every) if history else None
        self.run(None, use_graph)
        st = self.state(sync=False)
        with torch.cuda.stream(self.stream):
            res = (st["pred_xstart"] if self.single_only else st["pred_xstart2"])[:sum(self.lens)].clone()
            ev = torch.cuda.Event()
            ev.record(self.stream)
        return [res[o:o + t] for o, t in self.item_slices()], hist, ev

    def sample_ragged(self, cond, x_T, lens, use_graph=True):
        """Ragged MixerDiffusion.ddim_sample_loop: list of B results [T_i, C]."""
        items, _, ev = self.sample_ragged_async(cond, x_T, lens, use_graph)
        ev.synchronize()
        torch.cuda.current_stream(self.device).wait_stream(self.stream)
        return items

    # ---- teacher-forced module forwards (tests / swappable inner protocol) ----------------------------
    def module_forward(self, which, x, cond, t, x2=None):
        """which: 0 denoiser1, 1 denoiser2, 2 Mixer.forward, 3 denoiser1 in "dual_individual" mode (inputs are the CFG-doubled batch);
        4 ClassifierFreeSampleModelX2.forward (B un-doubled rows in, B combined rows out).  The schedule survives; a begun call does not."""
        x = x.to(self.device, torch.float32).contiguous()
        cond = cond.to(self.device, torch.float32).contiguous()
        x2c = x2.to(self.device, torch.float32).contiguous() if x2 is not None else None
        n, T = x.shape[:2]
        out = torch.empty(n, T, 262 if which == 0 else 524, device=self.device, dtype=torch.float32)   # which 1 with single_only=2: n = 4B rows; 3 = dual_individual
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.device(self.device):
            check(self.lib.mmdm_module_forward(self.h, which, C.c_void_p(x.data_ptr()), C.c_void_p(x2c.data_ptr() if x2c is not None else 0),
                                               C.c_void_p(cond.data_ptr()), int(t), C.c_void_p(out.data_ptr()), n, T, self._s()), self.h)
        self.stream.synchronize()
        return out

    def graph_stats(self):
        """(steps captured, steps replayed, graphs cached) of the handle's (B, T, S)-keyed hipGraph cache."""
        cap, rep, n = C.c_int64(), C.c_int64(), C.c_int()
        check(self.lib.mmdm_graph_stats(self.h, C.byref(cap), C.byref(rep), C.byref(n)), self.h)
        return cap.value, rep.value, n.value

    def graphs_parked(self):
        """Graph execs of the PROCESS kept past their cache entry's life (mmdm_graph_parked: include/mmdm.h, mmdm_create_shared)."""
        return int(self.lib.mmdm_graph_parked())

    # ---- profiling ----------------------------------------------------------------------------------
    def profile(self, on=True):
        check(self.lib.mmdm_profile_enable(self.h, int(on)), self.h)

    def profile_read(self, which):
        ms, n, fl, by = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
        check(self.lib.mmdm_profile_read(self.h, which, C.byref(ms), C.byref(n), C.byref(fl), C.byref(by)), self.h)
        return ms.value, n.value, fl.value, by.value


def _from_ptr(ptr, shape, device):
    """Wrap a device pointer owned by the handle as a torch tensor (no copy) via __cuda_array_interface__."""
    class _Holder:
        pass
    hld = _Holder()
    hld.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<f4", "data": (int(ptr), False), "version": 3, "strides": None}
    return torch.as_tensor(hld, device=device)
