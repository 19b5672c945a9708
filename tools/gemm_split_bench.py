import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, math, statistics, ctypes as C
from mixermdm_amd import ops, load_library
lib = load_library()
lib.mmdm_diag_set(b"split_ablate", int(os.environ.get('ABL','0')))
d = torch.device("cuda:0")
vp = lambda t: C.c_void_p(t.data_ptr() if t is not None else 0)
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
def split(x):
    n = x.numel()
    out = torch.empty(2, *x.shape, device=d, dtype=torch.float16)
    assert lib.mmdm_f32_split(vp(x), vp(out), n, n, st()) == 0
    return out
def lin_split(xs, ws, b, epi, extra, out, M, N, K):
    rc = lib.mmdm_linear_split(vp(xs), K, M * K, vp(ws), K, N * K, vp(b), vp(out), N, 0, 0, M, N, K, ops.EPI[epi], vp(extra), N if extra is not None else 0, 0, st())
    assert rc == 0, lib.mmdm_last_error()
shapes = [(19200,3072,1024,"qkv","bias"),(19200,1024,1024,"out","resid"),(19200,2048,1024,"ffn1","gelu"),(19200,1024,2048,"ffn2","resid"),
          (19200,1536,512,"m.qkv","bias"),(19200,512,512,"m.out","resid"),(19200,512,1024,"m.ffn2","resid"),(4096,4096,4096,"sq4k","bias")]
# accuracy vs float64
M,N,K = 512, 384, 1024
x = torch.randn(M,K,device=d); w = torch.randn(N,K,device=d)/math.sqrt(K); b = torch.randn(N,device=d)
ref = (x.double() @ w.double().T + b.double())
out = torch.empty(M,N,device=d)
xs, ws = split(x), split(w)
assert ((xs[0].double()+xs[1].double()/2048 - x.double()).abs() <= 2.0**-22 * x.double().abs() + 2.0**-36).all(), "split beyond 2^-22"
lin_split(xs, ws, b, "bias", None, out, M, N, K)
nat = ops.linear(x, w, b)
sc = ref.abs().mean()
print(f"err/mean|ref|: split mean {(out.double()-ref).abs().mean()/sc:.3e} max {(out.double()-ref).abs().max()/sc:.3e} | native fp32 MFMA mean {(nat.double()-ref).abs().mean()/sc:.3e} max {(nat.double()-ref).abs().max()/sc:.3e}")
cfgs = [int(c) for c in os.environ.get("CFGS","-1,1").split(",")]
# clock ramp: the first ~30 launches after an idle period run at lower clocks; warm up, or the first configuration reads ~10 % low
_w = torch.randn(4096, 4096, device=d)
for _ in range(60): ops.linear(_w, _w)
torch.cuda.synchronize()
for M,N,K,name,epi in shapes:
    x = torch.randn(M,K,device=d); w = torch.randn(N,K,device=d)/math.sqrt(K); b = torch.randn(N,device=d)
    out = torch.empty(M,N,device=d); extra = out if epi=="resid" else None
    xs, ws = split(x), split(w)
    line = f"{name:7s} {M}x{N}x{K} {epi:5s}"
    wp = torch.empty_like(ws)
    assert lib.mmdm_split_pack_weight(vp(ws), K, N * K, vp(wp), N * K, N, K, st()) == 0, lib.mmdm_last_error()
    def lin_packed(xs, wp, b, epi, extra, out, M, N, K):
        rc = lib.mmdm_linear_split_packed(vp(xs), K, M * K, vp(wp), N * K, vp(b), vp(out), N, 0, 0, M, N, K, ops.EPI[epi], vp(extra), N if extra is not None else 0, 0, st())
        assert rc == 0, lib.mmdm_last_error()
    if epi != "resid":
        o1 = torch.empty(M, N, device=d); o2 = torch.empty(M, N, device=d)
        lib.mmdm_diag_set(b"split_cfg", -1)
        lin_split(xs, ws, b, epi, None, o1, M, N, K); lin_packed(xs, wp, b, epi, None, o2, M, N, K)
        line += " | packed==planes: " + str(torch.equal(o1, o2))
        for c in os.environ.get("PCFGS", "p-1,p5").split(","):
            lib.mmdm_diag_set(b"split_cfg", int(c[1:])); lin_packed(xs, wp, b, epi, None, o2, M, N, K); line += "/" + str(torch.equal(o1, o2))
    pk = os.environ.get("PCFGS", "p-1,p5").split(",")
    for c in cfgs + pk:
        if isinstance(c, str):
            lib.mmdm_diag_set(b"split_cfg", int(c[1:]))
            ts = []
            for r in range(5):
                lin_packed(xs, wp, b, epi, extra, out, M, N, K)
                e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(4): lin_packed(xs, wp, b, epi, extra, out, M, N, K)
                e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1)/4)
            ms = statistics.median(ts); line += f" | {c}: {ms*1e3:7.1f}us {2*M*N*K/ms/1e9:6.1f}TF"
            continue
        lib.mmdm_diag_set(b"split_cfg", c)
        ts = []
        for r in range(5):
            lin_split(xs, ws, b, epi, extra, out, M, N, K)
            e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4): lin_split(xs, ws, b, epi, extra, out, M, N, K)
            e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1)/4)
        ms = statistics.median(ts); line += f" | cfg{c}: {ms*1e3:7.1f}us {2*M*N*K/ms/1e9:6.1f}TF"
    print(line, flush=True)
