"""A/B of dispatch rules inside the bf16_fp8 step on one box (mmdm_diag_set switches, graphs re-captured per setting):
   bf16_cfg 13 = the narrow fp8 tile everywhere (no 128 x 256 tile for the GELU epilogue), fp8p 0 / 1 / 2 = the persistent kernel nowhere / where shipped / wherever it covers."""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mixermdm_amd.sampler import Sampler
from mixermdm_amd._lib import diag
from mixermdm_amd.synthetic import synthetic_state_dict, synthetic_stats, synthetic_inputs, FULL_DIMS
B = int(os.environ.get("AB_BATCH", "16"))
sd = synthetic_state_dict(seed=0, std=0.02, bias_std=0.0, **FULL_DIMS); st = synthetic_stats()
cond, xT = [t.cuda() for t in synthetic_inputs(B, 300)]
def run(tag, sets):
    for k, v in sets: diag(k, v)
    s = Sampler(d_heads=8, m_heads=8, max_batch=B, max_frames=300, precision="bf16_fp8", **FULL_DIMS)
    s.load_state_dict(sd); s.set_norm_stats(st["mean_hml"], st["std_hml"], st["mean_ih"], st["std_ih"]); s.prepare(); s.set_schedule("ddim1000")
    s.begin(cond, xT); s.run(5); torch.cuda.synchronize()
    t0 = time.perf_counter(); s.run(30); torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 30 * 1e3
    x = s.state()["x"].clone(); s.close()
    print(f"{tag:60s} {ms:7.3f} ms/step", flush=True)
    return x
ref = None
for rnd in range(2):
    for tag, sets in [("persistent kernel on the cross-attention projections (fp8p 1)", [("fp8p", 1), ("bf16_cfg", -1), ("attn_kc32", 1)]), ("narrow fp8 tile everywhere (bf16_cfg 13)", [("fp8p", 1), ("bf16_cfg", 13), ("attn_kc32", 1)]),
                      ("shipped: no persistent kernel (fp8p 0), 32-key attention", [("fp8p", 0), ("bf16_cfg", -1), ("attn_kc32", 1)]), ("persistent kernel wherever it covers (fp8p 2)", [("fp8p", 2), ("bf16_cfg", -1), ("attn_kc32", 1)]),
                      ("16-key attention chunks (attn_kc32 0; other bits: not compared)", [("fp8p", 1), ("bf16_cfg", -1), ("attn_kc32", 0)])]:
        x = run(tag, sets)
        if "attn_kc32 0" in tag: continue
        if ref is None: ref = x
        else: assert torch.equal(x, ref), "a dispatch rule changed the step's bits: " + tag
diag("fp8p", 0); diag("bf16_cfg", -1); diag("attn_kc32", 1)
print("every GEMM dispatch setting: bitwise the same chains after 35 steps")
