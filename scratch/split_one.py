import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, math, ctypes as C
from mixermdm_amd import ops, load_library
lib = load_library(); lib.mmdmx_set_split_cfg.argtypes = [C.c_int]
lib.mmdmx_set_split_cfg(int(os.environ.get("CFG", "1")))
d = torch.device("cuda:0")
M, N, K = [int(v) for v in os.environ.get("SHAPE", "19200,3072,1024").split(",")]
x = torch.randn(M, K, device=d); w = torch.randn(N, K, device=d) / math.sqrt(K); b = torch.randn(N, device=d)
xs, ws = ops.split3(x), ops.split3(w)
for _ in range(6):
    out = ops.linear_split(xs, ws, b)
torch.cuda.synchronize()
