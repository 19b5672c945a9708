"""Concurrency accounting of the captured step from a rocprofv3 kernel trace (GPU box):
   python tools/step_timeline.py <dir with *kernel_trace.csv>   -> time with 0 / 1 / >=2 kernels in flight, per-kernel totals."""
import sys, glob, csv, collections, re
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# window = the last `nsteps` whole steps: from the end of one step_dec_kernel (last kernel of a step) to the end of the last one
decs = [e for s, e, k in rows if "step_dec_kernel" in k]
nsteps = min(4, len(decs) - 1)
t_lo, t_hi = decs[-1 - nsteps], decs[-1]
rows = [r for r in rows if r[0] >= t_lo and r[1] <= t_hi]
print("steps in window: %d, %.2f ms per step" % (nsteps, (t_hi - t_lo) / nsteps / 1e6))
ev = []
for s, e, _ in rows:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
lvl = collections.Counter(); cur = 0; last = ev[0][0]
for t, d in ev:
    lvl[min(cur, 2)] += t - last
    last = t; cur += d
tot = sum(lvl.values())
print("window %.1f ms: idle %.1f %%, one kernel %.1f %%, two or more %.1f %%" % (tot / 1e6, 100 * lvl[0] / tot, 100 * lvl[1] / tot, 100 * lvl[2] / tot))
agg = collections.defaultdict(lambda: [0, 0])
for s, e, k in rows:
    k = re.sub(r"\(anonymous namespace\)::", "", k)
    k = re.sub(r"^void ", "", k).split("(")[0][:72]
    agg[k][0] += e - s; agg[k][1] += 1
for k, (ns, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:14]:
    print("  %-72s %8.2f ms  x%-5d avg %7.1f us" % (k, ns / 1e6, n, ns / n / 1e3))
