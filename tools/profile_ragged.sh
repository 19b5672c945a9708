#!/bin/bash
# rocprofv3 evidence for the RAGGED workload (VERDICT r5 missing 4): kernel trace + PMC passes of one ragged batch of the evaluation caller
# (tools/ragged_step.py: 28 items, 4740 frames in 4864 rows, fp32), counters in their own runs.  Output: gpurun_out/prof_ragged_$TAG/
#   summary/kernel_stats_serial.csv, summary_pmc/pmc_summary.json, summary_pmc/gemm_traffic.json (-> profiles/gemm_traffic_ragged.json)
# usage: tools/profile_ragged.sh [tag]      (GPU box, from the repo root; ~4 minutes)
TAG=${1:-r06}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_ragged_$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
MMDM_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial -- python3 $R/tools/ragged_step.py 6 > $O/serial.json 2> $O/serial.err
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  MMDM_NO_OVERLAP=1 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc/pmc_$tag -- python3 $R/tools/ragged_step.py 2 > $O/pmc_$tag.json 2> $O/pmc_$tag.err
done
cd $R
ROWS=$(python3 -c "import json; print(json.load(open('$O/serial.json'))['rows'])")
PMC_ROWS=$ROWS python3 tools/pmc_summary.py $O/pmc $O/summary_pmc fp32
python3 tools/pmc_summary.py $O $O/summary fp32 traces-only
