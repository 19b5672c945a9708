"""Persistent fp8 GEMM (gemm_fp8p.hip, diagnostic build: tools/mk_variant.sh p_tl gemm_fp8p.hip -DFP8P_TL; MMDM_LIB=variants/libmmdm_p_tl.so): shader clocks per tile
of a workgroup's walk -- first tile (prologue exposed), steady tiles (epilogue of the previous tile inside), drain -- and the clock."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, math
from mixermdm_amd import ops, load_library
lib = load_library(); d = torch.device("cuda:0")
grid = int(os.environ.get("FP8P_GRID", "512")); lib.mmdm_diag_set(b"fp8p_grid", grid); lib.mmdm_diag_set(b"fp8p", 2)
for M, N, K, epi, od in [(19200, 3072, 1024, "bias", torch.bfloat16), (19200, 1024, 1024, "bias", torch.bfloat16), (19200, 2048, 1024, "gelu", torch.float8_e4m3fn), (76800, 3072, 1024, "bias", torch.bfloat16)]:
    x = torch.randn(M, K, device=d); w = torch.randn(N, K, device=d) / math.sqrt(K); b = torch.randn(N, device=d)
    xq, xs = ops.quantize_rows_fp8(x); wq, ws = ops.quantize_rows_fp8(w); wp = ops.pack_weight_frag(wq)
    call = lambda: ops.linear_fp8(xq, xs, wp, ws, b, epi, None, out_dtype=od, packed=True)
    for _ in range(int(os.environ.get("WARM", "200"))): call()
    tl = torch.zeros(grid * 32, device=d, dtype=torch.int64)
    lib.mmdm_diag_set(b"fp8p_timeline", tl.data_ptr()); call(); torch.cuda.synchronize(); lib.mmdm_diag_set(b"fp8p_timeline", 0)
    t = tl.view(grid, 32).cpu()
    ntl = (t[:, 31] & 0xffffffff); ticks = (t[:, 31] >> 32).double()
    live = ntl > 0
    t, ntl, ticks = t[live], ntl[live], ticks[live]
    first = (t[:, 1] - t[:, 0]).double()
    steady = []
    for r in range(t.shape[0]):
        n = int(ntl[r])
        steady += [float(t[r, k + 1] - t[r, k]) for k in range(1, min(n, 29))]
    steady = torch.tensor(steady) if steady else torch.zeros(1)
    last_body = torch.tensor([float(t[r, min(int(ntl[r]), 29)]) for r in range(t.shape[0])])
    drain = (t[:, 30].double() - last_body)
    total = (t[:, 30] - t[:, 0]).double()
    mhz = (100.0 * total / ticks.clamp(min=1)).median().item()
    print(f"{M}x{N}x{K} {epi} grid {grid}: tiles per workgroup {int(ntl.min())}-{int(ntl.max())}; first tile {first.median():.0f} clk, steady tile {steady.median():.0f} clk (p10 {steady.quantile(.1):.0f}, p90 {steady.quantile(.9):.0f}; "
          f"matrix pipe alone: 4096 per wave), drain {drain.median():.0f} clk; walk {total.median():.0f} clk = {ticks.median() / 100:.1f} us at {mhz:.0f} MHz", flush=True)
lib.mmdm_diag_set(b"fp8p", 0)
