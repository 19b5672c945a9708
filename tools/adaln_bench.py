"""AdaLN pass at the layer shapes, every output mode: us per launch and HBM-side TB/s.
usage: python tools/adaln_bench.py   (GPU)"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ctypes as C
from mixermdm_amd import load_library
lib = load_library()
d = torch.device("cuda:0")
vp = lambda t: C.c_void_p(t.data_ptr() if t is not None else 0)
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
def timeit(f, n=200):
    for _ in range(20): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for nseq, T, D in [(64, 300, 1024), (64, 300, 512), (256, 300, 1024)]:
    rows = nseq * T
    h = torch.randn(rows, D, device=d); ss = torch.randn(nseq, 2 * D, device=d)
    for mode, name, ob in [(0, "fp32", 4), (1, "bf16", 2), (2, "fp16x2", 4), (3, "fp8", 1)]:
        out = torch.empty(rows * D * 4, dtype=torch.uint8, device=d); sc = torch.empty(rows, device=d)
        f = (lambda: lib.mmdm_adaln_fp8(vp(h), vp(ss), 2 * D, nseq, vp(out), vp(sc), nseq, T, D, st())) if mode == 3 else \
            (lambda: lib.mmdm_adaln_ex(vp(h), vp(ss), 2 * D, nseq, vp(out), mode, nseq, T, D, st()))
        assert f() == 0
        us = timeit(f)
        mb = rows * D * (4 + ob) / 1e6
        print(f"rows {rows:6d} D {D:5d} {name:7s}: {us:6.1f} us ({mb / us:5.2f} TB/s)")
