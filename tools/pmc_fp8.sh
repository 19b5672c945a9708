#!/bin/bash
# PMC passes over the stand-alone fp8 GEMM bench (GPU box): what bounds gemm_bf16w_kernel<1, .> -- matrix-pipe busy and the wave-cycle
# breakdown, the L1 -> L2 request rate (DESIGN 10.2's "L2 roof" as a number), LDS array cycles / bank conflicts.
# usage: tools/pmc_fp8.sh [tag]      -> gpurun_out/pmc_fp8_$TAG/summary.json  (counters in their own runs beside --kernel-trace only)
TAG=${1:-r06}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_fp8_$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_F8 GRBM_GUI_ACTIVE" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  FP8_ONLY=${FP8_ONLY:-packed-t} FP8_SHAPES=${FP8_SHAPES:-qkv,ffn2} FP8_ITERS=1 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pass$i -- python3 $R/tools/gemm_fp8_bench.py > $O/pass$i.log 2>&1
done
cd $R
python3 tools/pmc_fp8_summary.py $O > $O/summary.txt; cat $O/summary.txt
