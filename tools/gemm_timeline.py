"""Per-workgroup timeline of one fp32 GEMM launch (GPU box): when does each workgroup start, enter its K loop, leave it; how many share a CU.
Uses the diagnostic stamps of gemm_glds_kernel (mmdm_diag_set "gemm_stamps"); production launches carry a null stamp pointer."""
import os, sys, math, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from mixermdm_amd import ops, load_library
lib = load_library()
d = torch.device("cuda:0")
shapes = {"out": (19200, 1024, 1024, "resid"), "ffn2": (19200, 1024, 2048, "resid"), "qkv": (19200, 3072, 1024, "bias"), "ffn1": (19200, 2048, 1024, "gelu")}
ops.linear(torch.zeros(8, 64, device=d), torch.zeros(8, 64, device=d))     # lazy init first (it resets the forced configuration)
lib.mmdm_diag_set(b"gemm_cfg", int(os.environ.get("CFG", "-1")))
lib.mmdm_diag_set(b"gemm_ablate", int(os.environ.get("ABL", "0")))
for name in (sys.argv[1:] or ["out", "qkv"]):
    M, N, K, epi = shapes[name]
    x = torch.randn(M, K, device=d); w = torch.randn(N, K, device=d) / math.sqrt(K); b = torch.randn(N, device=d)
    out = torch.empty(M, N, device=d)
    extra = out if epi == "resid" else None
    for _ in range(3):
        ops.linear(x, w, b, epi, extra, out=out)
    stamps = torch.zeros(8192 * 8, dtype=torch.int64, device=d)
    lib.mmdm_diag_set(b"gemm_stamps", stamps.data_ptr())
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); ops.linear(x, w, b, epi, extra, out=out); e1.record(); torch.cuda.synchronize()
    lib.mmdm_diag_set(b"gemm_stamps", 0)
    kern = lib.mmdm_last_gemm_kernel().decode()
    raw = stamps.cpu().numpy()
    import re
    tm_, tn_ = [int(v) for v in re.search(r"<(\d+),(\d+)", kern).groups()]
    bm, bn = 32 * (tm_ // 10) * (tm_ % 10), 32 * (tn_ // 10) * (tn_ % 10)
    n = ((M + bm - 1) // bm) * ((N + bn - 1) // bn)                 # workgroups of the launch
    s = raw[:8 * n].reshape(n, 8)
    ck = raw[8 * n:10 * n].reshape(n, 2)                             # s_memtime at loop start / loop end
    mhz = 100.0 * (ck[:, 1] - ck[:, 0]) / np.maximum(s[:, 2] - s[:, 1], 1)
    print(f"{name}: shader clock inside the K loop (s_memtime / s_memrealtime): median {np.median(mhz):.0f} MHz, min {mhz.min():.0f}, max {mhz.max():.0f}"
          f"  -> the fp32 matrix peak at that clock is {157.3 * np.median(mhz) / 2400:.1f} TFLOP/s")
    t0 = s[:, 0].min()
    t0 = s[:, 5].min()
    st, lp, en = (s[:, 0] - t0) / 100.0, (s[:, 1] - t0) / 100.0, (s[:, 2] - t0) / 100.0      # us
    ke, ent = (s[:, 4] - t0) / 100.0, (s[:, 5] - t0) / 100.0                                  # kernel end (after the epilogue), kernel entry (before the stagger wait)
    hw = s[:, 3] & 0xFFFFFFFF; xcc = s[:, 3] >> 32
    cu = ((hw >> 8) & 0xF) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5) | (xcc << 8)       # cu_id, sh_id, se_id, xcc
    ucu, per_cu = np.unique(cu, return_counts=True)
    print(f"== {name} {M}x{N}x{K} {epi}: {kern}; event time {e0.elapsed_time(e1)*1e3:.0f} us; {n} workgroups on {len(ucu)} CUs "
          f"(per CU min {per_cu.min()} max {per_cu.max()}; hist {np.bincount(per_cu).tolist()})")
    print(f"   start: p50 {np.median(st):.1f} p99 {np.percentile(st,99):.1f} max {st.max():.1f} us | prologue (start->loop) mean {np.mean(lp-st):.2f} us | "
          f"loop mean {np.mean(en-lp):.1f} min {np.min(en-lp):.1f} max {np.max(en-lp):.1f} us | last loop end {en.max():.1f} us")
    ideal_tile = 2.0 * 128 * 128 * K / (157.3e12 / 256) * 1e6   # us if one 128x128 tile had a whole CU at peak
    print(f"   a 128x128xK tile alone on a CU at the MFMA peak: {ideal_tile:.1f} us; sum over a 5-tile CU {5*ideal_tile:.1f} us")
    # finish-time distribution and concurrency over time (how many workgroups are inside their loop)
    grid = np.linspace(0, en.max(), 21)
    act = [(int(((lp <= t) & (en > t)).sum())) for t in grid]
    print("   workgroups in their K loop at 0,5,..100 % of the kernel: " + " ".join(str(a) for a in act))
    for c in ucu[:2]:
        idx = np.where(cu == c)[0]
        print(f"   CU {c} (entry/start->loop end/kernel end): " + "; ".join(f"wg{idx[k]} {ent[idx[k]]:.0f}/{st[idx[k]]:.0f}->{en[idx[k]]:.0f}/{ke[idx[k]]:.0f}" for k in np.argsort(st[idx])))
    rvw, sti = (s[:, 6] - t0) / 100.0, (s[:, 7] - t0) / 100.0
    print(f"   loop end -> residual tile landed: mean {np.mean(rvw-en):.2f} us | -> stores issued: mean {np.mean(sti-rvw):.2f} us | -> stores acknowledged: mean {np.mean(ke-sti):.2f} us")
    print(f"   epilogue (loop end -> kernel end) mean {np.mean(ke-en):.2f} max {np.max(ke-en):.2f} us")
