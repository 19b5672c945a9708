"""Copy what tools/bench_round.sh left under gpurun_out/bench_<tag>/ into profiles/<tag>_bench_*.json (tracked) and print the figures.
usage: python tools/collect_bench.py [round-tag, default r04]"""
import json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
src, dst = os.path.join(ROOT, "gpurun_out", f"bench_{tag}"), os.path.join(ROOT, "profiles")
for n in ("default", "single", "bf16", "bf16_fp8", "fp8_b64", "b32", "b32_split", "infer_b1", "infer_b15", "configs0"):
    f = os.path.join(src, n + ".json")
    line = next((l for l in open(f) if l.startswith("{")), None) if os.path.exists(f) else None
    if not line:
        print(n, "missing"); continue
    d = json.loads(line)
    json.dump(d, open(os.path.join(dst, f"{tag}_bench_{n}.json"), "w"), indent=1)
    r = d["roofline"]
    extra = ""
    if d.get("fp32_split"):
        s = d["fp32_split"]; extra += f"  | fp32_split {s['ms_per_step']} ms/step frac {s['roofline']['frac']}"
    if d.get("full_loop"):
        extra += f"  | full loop {d['full_loop']}"
    if d.get("cpu_baseline"):
        extra += f"  | cpu {d['cpu_baseline'].get('value')} {d['cpu_baseline'].get('unit')}"
        if d["cpu_baseline"].get("whole_host"):
            extra += f" (whole host: {d['cpu_baseline']['whole_host'].get('motions_per_s')})"
    if d.get("facade"):
        extra += f"  | facade {d['facade']}"
    print(f"{n:10s} {d['ms_per_step']:8.3f} ms/step  {d['value']:.5f} {d['unit']}  roofline {r['achieved']} / {r['peak']} = {r['frac']}  traffic {r['traffic']}  attention {r.get('attention')}{extra}")
for n in ("eval64", "eval64_split"):        # bench.py --eval-items: items/s per strategy
    f = os.path.join(src, n + ".json")
    line = next((l for l in open(f) if l.startswith("{")), None) if os.path.exists(f) else None
    if not line:
        print(n, "missing"); continue
    d = json.loads(line)
    json.dump(d, open(os.path.join(dst, f"{tag}_bench_{n}.json"), "w"), indent=1)
    print(n, {k: (v["items_per_s"], v["frac_of_peak"], v.get("speedup_vs_sequential_forward_test")) for k, v in d["strategies"].items()}, "bit-identical:", d["bit_identical_to_sequential"],
          "roofline", d["roofline"] and d["roofline"]["frac"], "cpu", d["cpu_baseline"] and d["cpu_baseline"]["value"])
if os.path.exists(os.path.join(src, "full_loops.txt")):
    shutil.copy(os.path.join(src, "full_loops.txt"), os.path.join(dst, f"{tag}_full_loops.txt"))
    print(open(os.path.join(dst, f"{tag}_full_loops.txt")).read())
