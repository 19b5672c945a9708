#!/bin/bash
# PMC pass over the stand-alone GEMM bench (GPU box): MFMA busy, clock.  usage: tools/pmc_gemm.sh <cfg> <shape...>
cd /tmp && export TMPDIR=/tmp
CFG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_gemm_$CFG
rm -rf $OUT; mkdir -p $OUT
CFGS=$CFG rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/gemm_bench.py "$@" > $OUT/log.txt 2>&1
python3 - "$OUT" <<'PY'
import sys, glob, csv, collections
out = sys.argv[1]
cnt = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
dur = {}
for f in glob.glob(out + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gemm" not in k: continue
        key = (k[:90], r["Grid_Size"])
        cnt[key][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            n[key] += 1
            cnt[key]["ns"] += dur.get(r["Dispatch_Id"], ("", 0))[1]
for key, c in cnt.items():
    m = n[key]
    ns = c["ns"] / m
    clk = c["GRBM_GUI_ACTIVE"] / m / 8 / ns        # GHz
    busy = c["SQ_VALU_MFMA_BUSY_CYCLES"] / m / (1024 * c["GRBM_GUI_ACTIVE"] / m / 8)
    wc = c["SQ_WAVE_CYCLES"]; 
    print(f"{key[0][-60:]} grid {key[1]} x{m}: {ns/1e3:.0f} us, clock {clk:.2f} GHz, MFMA busy {busy*100:.1f} %, wave cycles: wait_any {c['SQ_WAIT_ANY']/wc*100:.0f} % wait_inst {c['SQ_WAIT_INST_ANY']/wc*100:.0f} % active {c['SQ_ACTIVE_INST_ANY']/wc*100:.0f} %")
PY
