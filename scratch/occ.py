import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mixermdm_amd import load_library
torch.cuda.init(); torch.zeros(1,device="cuda")
lib=load_library()
for dh in (128,64):
    print(dh, [lib.mmdmx_attn_occupancy(dh, e) for e in (0, 8192, 16384)])
