"""Where a wave of the packed fp32-split GEMM spends its K step: s_memtime sums per phase (mmdm_diag_set "split_timeline").
Phases of one step (two-way fp16 form, TM x TN = 4 tiles per wave): 0 = 16 MFMAs (k-block 0 and ah*wl of k-block 1; + A fragment reads, next
step's B loads, LDS-DMA requests); 2 = counted vmcnt wait; 3 = barrier; 4 = last 8 MFMAs (+ next stage's A reads)."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, math, ctypes as C
from mixermdm_amd import ops, load_library
lib = load_library()
d = torch.device("cuda:0")
vp = lambda t: C.c_void_p(t.data_ptr() if t is not None else 0)
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
def split(x):
    out = torch.empty(2, *x.shape, device=d, dtype=torch.float16); n = x.numel()
    assert lib.mmdm_f32_split(vp(x), vp(out), n, n, st()) == 0
    return out
for cfg in [int(c) for c in os.environ.get("CFGS", "-1,5").split(",")]:
  lib.mmdm_diag_set(b"split_cfg", cfg)
  print("split cfg", cfg)
  for M, N, K in [(19200, 3072, 1024), (4096, 4096, 4096)]:
      x = torch.randn(M, K, device=d); w = torch.randn(N, K, device=d) / math.sqrt(K); b = torch.randn(N, device=d)
      out = torch.empty(M, N, device=d)
      xs, ws = split(x), split(w)
      wp = torch.empty_like(ws)
      assert lib.mmdm_split_pack_weight(vp(ws), K, N * K, vp(wp), N * K, N, K, st()) == 0
      call = lambda: lib.mmdm_linear_split_packed(vp(xs), K, M * K, vp(wp), N * K, vp(b), vp(out), N, 0, 0, M, N, K, 0, None, 0, 0, st())
      for _ in range(int(os.environ.get("WARM", "30"))): call()
      nwg = ((M + 127) // 128) * (N // 128)
      tl = torch.zeros(nwg * 4 * 8, device=d, dtype=torch.int64)
      lib.mmdm_diag_set(b"split_timeline", tl.data_ptr()); assert call() == 0; torch.cuda.synchronize(); lib.mmdm_diag_set(b"split_timeline", 0)
      t = tl.view(nwg, 4, 8).double().cpu()
      nkt = t[0, 0, 5].item()
      per = t[:, :, :5] / nkt                     # cycles per step and phase
      mean = per.mean(dim=(0, 1)); tot = mean.sum().item()
      ideal = 24 * 32 * 2                          # 24 MFMAs x 32 cycles (8 passes x 4), two waves per SIMD
      mhz = (t[:, :, 7] / t[:, :, 6].clamp(min=1)).median().item() * 100
      print(f"s_memtime / s_memrealtime over the loop: {mhz:.0f} MHz; loop {t[:, :, 6].median().item() / 100:.1f} us per tile")
      print(f"{M}x{N}x{K}: {tot:.0f} shader clocks per step (MFMA-bound: {ideal}); phases " + " ".join(f"{v:.0f}" for v in mean.tolist())
            + f" | per-wave spread of the total: min {per.sum(-1).min():.0f} max {per.sum(-1).max():.0f}")
