#!/bin/bash
# BASELINE configs[4]'s per-GPU shard under the profiler (GPU box): bf16_fp8 at B = 64, T = 300 -- kernel trace (one stream) + PMC passes ("rocprof MFMA-util + HBM
# GB/s vs roofline" at the shard size the config names: 512 motions on 8 GPUs).  Output: gpurun_out/prof_fp8_b64_$TAG/
# usage: tools/profile_fp8_b64.sh [tag]     (then: python tools/collect_profiles.py picks the files up as profiles/${TAG}_*_fp8_b64*)
TAG=${1:-r06}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_fp8_b64_$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --precision bf16_fp8 --batch 64 --no-cpu-baseline --no-alt --no-side --no-full-loop --no-clock"
MMDM_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial -- python3 $B --steps 6 --warmup 2 > $O/serial.json 2> $O/serial.err
P="$B --steps 2 --warmup 1 --no-graph --profile-steps 0"
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  MMDM_NO_OVERLAP=1 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_bf16_fp8/pmc_$tag -- python3 $P > $O/pmc_$tag.json 2> $O/pmc_$tag.err
done
cd $R
PMC_BATCH=64 PMC_FRAMES=300 python3 tools/pmc_summary.py $O/pmc_bf16_fp8 $O/summary_bf16_fp8 bf16_fp8 > $O/summary.log 2>&1
python3 tools/pmc_summary.py $O $O/summary bf16_fp8 traces-only >> $O/summary.log 2>&1
head -12 $O/summary/kernel_stats_serial.csv; tail -5 $O/summary.log
