"""Per-workgroup timeline of one fp32 attention launch (GPU box): entry, loop start, loop end, kernel end, CU placement, and wave 0's time in the
three phases of a chunk (Q K^T, softmax, P V).  Uses the diagnostic stamps of attn_mfma_kernel (mmdm_diag_set "attn_stamps")."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from mixermdm_amd import ops, load_library
lib = load_library()
d = torch.device("cuda:0")
ops.attention(torch.zeros(1,16,64,device=d),torch.zeros(1,16,64,device=d),torch.zeros(1,16,64,device=d),1)
lib.mmdm_diag_set(b"attn_ablate", int(os.environ.get("ABL","0")))
for nseq, T, H, dh in [(64, 300, 8, 128), (64, 300, 8, 64)]:
    D = H * dh
    qkv = torch.randn(nseq, T, 3 * D, device=d)
    for _ in range(3):
        ops.attention(qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:], H)
    stamps = torch.zeros(4096 * 8, dtype=torch.int64, device=d)
    lib.mmdm_diag_set(b"attn_stamps", stamps.data_ptr())
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); ops.attention(qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:], H); e1.record(); torch.cuda.synchronize()
    lib.mmdm_diag_set(b"attn_stamps", 0)
    s = stamps.cpu().numpy().reshape(-1, 8)
    s = s[s[:, 0] != 0]
    t0 = s[:, 0].min()
    ent, lp, le, ke = [(s[:, i] - t0) / 100.0 for i in range(4)]
    qk, sm, pv = s[:, 5] / 100.0, s[:, 6] / 100.0, s[:, 7] / 100.0
    hw = s[:, 4] & 0xFFFFFFFF; xcc = s[:, 4] >> 32
    cu = ((hw >> 8) & 0xF) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5) | (xcc << 8)
    ucu, per = np.unique(cu, return_counts=True)
    print(f"== nseq={nseq} T={T} H={H} dh={dh}: event time {e0.elapsed_time(e1) * 1e3:.0f} us; {len(s)} workgroups on {len(ucu)} CUs (per CU {per.min()}..{per.max()})")
    print(f"   prologue (entry -> loop) mean {np.mean(lp - ent):.2f} us | loop mean {np.mean(le - lp):.1f} (min {np.min(le - lp):.1f} max {np.max(le - lp):.1f}) us | "
          f"epilogue mean {np.mean(ke - le):.2f} us | last end {ke.max():.1f} us")
    print(f"   wave 0 per workgroup: Q K^T {np.mean(qk):.1f} us, softmax {np.mean(sm):.1f} us, P V {np.mean(pv):.1f} us (sum {np.mean(qk + sm + pv):.1f} of loop {np.mean(le - lp):.1f})")
    grid = np.linspace(0, ke.max(), 21)
    print("   workgroups in their loop at 0,5,..100 %: " + " ".join(str(int(((lp <= t) & (le > t)).sum())) for t in grid))
    for c in ucu[:2]:
        idx = np.where(cu == c)[0]
        print(f"   CU {c}: " + "; ".join(f"{ent[i]:.0f}/{lp[i]:.0f}->{le[i]:.0f}/{ke[i]:.0f}" for i in idx[np.argsort(ent[idx])]))
