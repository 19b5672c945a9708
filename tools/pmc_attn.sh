#!/bin/bash
# PMC pass over the stand-alone attention bench (GPU box)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_attn
rm -rf $OUT; mkdir -p $OUT
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAVE_CYCLES"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/$tag -- python3 $GRAFT_REPO_ROOT/tools/attn_bench.py > $OUT/$tag.log 2>&1
done
python3 - "$OUT" <<'PY'
import sys, glob, csv, collections
out = sys.argv[1]
cnt = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "attn" not in k: continue
        key = k.split("(")[0][-40:]
        cnt[key][r["Counter_Name"]] += float(r["Counter_Value"]); n[key][r["Counter_Name"]] += 1
for key, c in cnt.items():
    a = {k: v / n[key][k] for k, v in c.items()}
    wc = a["SQ_WAVE_CYCLES"]
    print(key)
    print("  MFMA busy %.1f %% of SIMD cycles; clock-cycles (GRBM/8) %.0f" % (100 * a["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * a["GRBM_GUI_ACTIVE"] / 8), a["GRBM_GUI_ACTIVE"] / 8))
    print("  wave cycles: wait_any %.0f %%  wait_inst_any %.0f %% (lds %.0f %%)  active %.0f %% | active VALU %.0f %%  active LDS %.0f %%" % tuple(100 * a[k] / wc for k in
          ["SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"]))
    print("  insts per MFMA: VALU %.2f  LDS %.2f | LDS bank conflict cycles / LDS active %.2f" % (a["SQ_INSTS_VALU"] / a["SQ_INSTS_MFMA"], a["SQ_INSTS_LDS"] / a["SQ_INSTS_MFMA"], a["SQ_LDS_BANK_CONFLICT"] / max(1, a["SQ_LDS_IDX_ACTIVE"])))
PY
