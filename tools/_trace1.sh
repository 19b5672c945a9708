#!/bin/bash
# scratch: one serial kernel trace of a precision mode -> gpurun_out/tr_$1/
P=${1:-fp32_split}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/tr_$P; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
MMDM_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 $R/bench.py --no-cpu-baseline --no-alt --no-full-loop --no-clock --precision $P --steps 6 --warmup 2 ${@:2} > $O/b.json 2> $O/b.err
cd $R
f=$(find $O/t -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:16]:
    print(f"  {r['Name'][:80]:80s} x{int(r['Calls']):5d} {float(r['AverageNs'])/1e3:8.1f} us {float(r['TotalDurationNs'])/tot*100:6.2f} %")
print('  total ms/step', tot/1e6/10)
PY
