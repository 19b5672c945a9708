"""scratch: host cost of a graph replay at B = 1 and whether two handles overlap when driven from one / two host threads."""
import sys, os, time, threading, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
GRAPH = os.environ.get("PROBE_EAGER") != "1"
from mixermdm_amd.sampler import Sampler
from mixermdm_amd.synthetic import synthetic_state_dict, synthetic_stats, synthetic_inputs, FULL_DIMS
sd = synthetic_state_dict(seed=0, std=0.02, bias_std=0.0, **FULL_DIMS)
st = synthetic_stats()
s = Sampler(d_heads=8, m_heads=8, max_batch=1, max_frames=300, **FULL_DIMS)
s.load_state_dict(sd); s.set_norm_stats(st["mean_hml"], st["std_hml"], st["mean_ih"], st["std_ih"]); s.prepare()
pool = [s] + [s.share() for _ in range(3)]
T = int(sys.argv[1]) if len(sys.argv) > 1 else 180
cond, x = synthetic_inputs(1, T); cond, x = cond.cuda(), x.cuda()
for p in pool:
    p.set_schedule("ddim50"); p.begin(cond, x); p.run(2)      # capture, single-threaded
torch.cuda.synchronize()
# host cost of enqueueing 40 replays vs the GPU time
p = pool[0]; p.begin(cond, x); torch.cuda.synchronize()
t0 = time.perf_counter(); p.run(40, use_graph=GRAPH); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("one handle: host enqueue %.2f ms/step, GPU %.2f ms/step" % ((t1 - t0) / 40 * 1e3, (t2 - t0) / 40 * 1e3), flush=True)
def timed(K, threads):
    for q in pool[:K]: q.begin(cond, x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if threads:
        def w(q):
            with torch.cuda.device(0): q.run(40, use_graph=GRAPH)
        th = [threading.Thread(target=w, args=(q,)) for q in pool[:K]]
        [t.start() for t in th]; [t.join() for t in th]
    else:
        for k in range(8):
            for q in pool[:K]: q.run(5, use_graph=GRAPH)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 40 * 1e3
for K in (1, 2, 4):
    print("K=%d one host thread: %.2f ms per step-of-all (%.2f per item-step)" % (K, timed(K, False), timed(K, False) / K), flush=True)
for K in (2, 4):
    v = timed(K, True)
    print("K=%d host threads:    %.2f ms per step-of-all (%.2f per item-step)" % (K, v, v / K), flush=True)
# capture on one thread while another replays
def cap():
    c2, x2 = synthetic_inputs(1, T - 7); 
    with torch.cuda.device(0):
        pool[1].begin(c2.cuda(), x2.cuda()); pool[1].run(10)
def rep():
    with torch.cuda.device(0): pool[0].run(10)
pool[0].begin(cond, x)
a, b = threading.Thread(target=cap), threading.Thread(target=rep)
a.start(); b.start(); a.join(); b.join(); torch.cuda.synchronize()
print("capture beside replay: ok", flush=True)

# two threads that both have to capture new shapes
def cap2(q, t):
    c2, x2 = synthetic_inputs(1, t)
    with torch.cuda.device(0):
        for k in range(3):
            q.begin(c2.cuda(), x2[:, :t - 3 * k].contiguous().cuda()); q.run(5)
ths = [threading.Thread(target=cap2, args=(pool[i], 100 + 11 * i)) for i in range(4)]
[t.start() for t in ths]; [t.join() for t in ths]; torch.cuda.synchronize()
print("concurrent captures: ok", flush=True)
