"""GPU: INTEGRATION.md option B executed -- the reference-side ctypes binding of include/mmdm.h, with NOTHING from mixermdm_amd's Python
host layer between the test and the library except the schedule tables (which the reference computes itself in MixerDiffusion):
the `Cfg` structure is taken verbatim from the document's code block, the call sequence is the document's, and the result must equal the
Sampler's (the Python mirror of the same ABI) bit for bit."""
import ctypes as C
import os
import re
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cfg_class_from_integration_md():
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"class Cfg\(C\.Structure\):.*?\n\n", md, re.S)
    assert m, "INTEGRATION.md no longer holds the Cfg structure of option B"
    ns = {"C": C}
    exec(m.group(0), ns)
    return ns["Cfg"]


def test_option_b_ctypes_binding_equals_the_python_mirror():
    from mixermdm_amd.sampler import Sampler, pe_table
    from mixermdm_amd.schedule import make_schedule
    from mixermdm_amd.synthetic import synthetic_state_dict, synthetic_stats, synthetic_inputs
    dims = dict(d_latent=512, d_ff=1024, d_layers=2, m_latent=256, m_ff=512, m_layers=2)
    sd = synthetic_state_dict(seed=4, std=0.03, bias_std=0.02, **dims)
    st = synthetic_stats()
    B, T, STEPS = 2, 40, 4
    cond, xT = synthetic_inputs(B, T, seed_cond=71, seed_x=72)
    dev = torch.device("cuda:0")

    # ---- the document's binding -------------------------------------------------------------------
    Cfg = _cfg_class_from_integration_md()
    lib = C.CDLL(os.path.join(ROOT, "mixermdm_amd", "libmmdm_hip.so"))          # after `import torch`: one HIP runtime per process
    lib.mmdm_handle_error.restype = C.c_char_p
    vp = C.c_void_p
    cfg = Cfg(512, 1024, 2, 4, 256, 512, 2, 4, 262, 768, 4, 1, 1, 0, 0, 0.0, 3.5, B, T, 0, 0.0, 0.0, 0, 0, 0, 0, 0, 0)
    h = vp()
    assert lib.mmdm_create(C.byref(cfg), C.byref(h)) == 0
    stream = torch.cuda.Stream()
    s = vp(stream.cuda_stream)
    weights = dict(sd)
    weights["denoiser1.sequence_pos_encoder.pe"] = pe_table(512)
    weights["denoiser2.sequence_pos_encoder.pe"] = pe_table(512)
    weights["sequence_pos_encoder.pe"] = pe_table(256)
    keep = []
    for k, v in weights.items():
        v = v.float().contiguous().to(dev)
        keep.append(v)
        r, c = (v.shape[0], 1) if v.dim() == 1 else v.shape
        assert lib.mmdm_set_weight(h, k.encode(), vp(v.data_ptr()), r, c, s) == 0, lib.mmdm_handle_error(h)
    stream.synchronize()
    stats = np.ascontiguousarray(np.concatenate([np.asarray(st[k], dtype=np.float32).reshape(262) for k in ("mean_hml", "std_hml", "mean_ih", "std_ih")]))
    assert lib.mmdm_set_norm_stats(h, stats.ctypes.data_as(vp)) == 0
    assert lib.mmdm_prepare(h) == 0, lib.mmdm_handle_error(h)
    sch = make_schedule("cosine", 1000, "ddim50")                                # the reference builds these tables in MixerDiffusion.__init__
    tmap = np.ascontiguousarray(np.array(sch.timestep_map, dtype=np.int32))
    coef = np.ascontiguousarray(sch.device_coefficients())
    assert lib.mmdm_set_schedule(h, tmap.ctypes.data_as(vp), coef.ctypes.data_as(vp), len(tmap), s) == 0
    cd, xd = cond.to(dev).contiguous(), xT.to(dev).contiguous()
    assert lib.mmdm_begin(h, vp(cd.data_ptr()), vp(xd.data_ptr()), B, T, s) == 0, lib.mmdm_handle_error(h)
    assert lib.mmdm_run(h, STEPS, 1, s) == 0, lib.mmdm_handle_error(h)
    px, px2 = vp(), vp()
    assert lib.mmdm_get_state(h, C.byref(px), C.byref(px2), None, None, None) == 0
    stream.synchronize()
    class _View:                                                                 # handle-owned device memory as a tensor (no copy)
        def __init__(self, ptr):
            self.__cuda_array_interface__ = {"shape": (B, T, 524), "typestr": "<f4", "data": (int(ptr), False), "version": 3, "strides": None}
    got = {nm: torch.as_tensor(_View(p.value), device=dev).clone() for nm, p in (("x", px), ("x2", px2))}
    torch.cuda.synchronize()
    lib.mmdm_destroy.restype = None
    lib.mmdm_destroy(h)

    # ---- the Python mirror ------------------------------------------------------------------------
    smp = Sampler(d_heads=4, m_heads=4, max_batch=B, max_frames=T, **dims)
    smp.load_state_dict(sd)
    smp.set_norm_stats(st["mean_hml"], st["std_hml"], st["mean_ih"], st["std_ih"])
    smp.prepare()
    smp.set_schedule("ddim50")
    smp.begin(cond, xT)
    smp.run(STEPS, use_graph=True)
    ref = smp.state()
    assert torch.isfinite(got["x"]).all()
    assert torch.equal(got["x"], ref["x"]) and torch.equal(got["x2"], ref["x2"])
    smp.close()
