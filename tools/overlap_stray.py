"""scratch: does one low-precision handle WRITE into (or READ from) another handle's device memory?  Needs the debug build of tools/mk_debug_lib.py
(MMDM_LIB=build/libmmdm_debug.so): snapshot every allocation of handle A, run handle B alone, list A's allocations that changed; then poison
B's memory and see whether A's result moves."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mixermdm_amd.sampler import Sampler
from mixermdm_amd._lib import diag
from mixermdm_amd.synthetic import synthetic_state_dict, synthetic_stats, synthetic_inputs, FULL_DIMS
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
sd = synthetic_state_dict(seed=0, std=0.02, bias_std=0.0, **FULL_DIMS); st = synthetic_stats()
def fresh():
    t = Sampler(d_heads=8, m_heads=8, max_batch=1, max_frames=300, precision=prec, **FULL_DIMS)
    t.load_state_dict(sd); t.set_norm_stats(st["mean_hml"], st["std_hml"], st["mean_ih"], st["std_ih"]); t.prepare(); t.set_schedule("ddim50")
    return t
A, B = fresh(), fresh()
ia = tuple(t.cuda() for t in synthetic_inputs(1, 181, seed_cond=181, seed_x=182))
ib = tuple(t.cuda() for t in synthetic_inputs(1, 263, seed_cond=263, seed_x=264))
refA = A.sample(*ia); refB = B.sample(*ib)
for nm, V, iv, W, iw in (("A", A, ia, B, ib), ("B", B, ib, A, ia)):
    V.begin(*iv); V.run(3); V.synchronize()
    print("== snapshot of", nm, "mid-call; the other handle samples alone; what changed in", nm, ":", flush=True)
    diag("snap_handle", V.h.value); diag("snap_weights", V.h.value)
    W.sample(*iw)
    torch.cuda.synchronize()
    diag("diff_handle", V.h.value); diag("diff_weights", V.h.value)
print("== A alone again:", "eq" if torch.equal(A.sample(*ia), refA) else "DIFF", flush=True)
diag("poison_handle", B.h.value)
o = A.sample(*ia)
print("== A with every allocation of B poisoned (0x7f bytes):", "eq" if torch.equal(o, refA) else "DIFF max|d| %.3e" % (o - refA).abs().max().item(), flush=True)
