#!/bin/bash
# The reference's own call shape under the profiler (GPU box): B = 1, T = 299, ddim50 through the MixerMDM facade (src/scripts/infer/mixermdm.py:73,117-124) --
# kernel trace (one stream: clean per-kernel durations) + PMC passes (matrix-pipe busy, HBM-side bytes).  Output: gpurun_out/prof_b1_$TAG/
# usage: tools/profile_b1.sh [tag]     (then: python tools/collect_profiles.py picks the files up as profiles/${TAG}_*_infer_b1.*)
TAG=${1:-r06}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_b1_$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --batch 1 --frames 299 --sampler ddim50 --facade --no-cpu-baseline --no-alt --no-side --no-full-loop --no-clock"
MMDM_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial -- python3 $B --steps 20 --warmup 3 > $O/serial.json 2> $O/serial.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/overlap -- python3 $B --steps 20 --warmup 3 > $O/overlap.json 2> $O/overlap.err
# (PMC passes: the same sampler without the facade's two whole 50-step calls -- counter collection serialises every launch at milliseconds each)
P="$R/bench.py --batch 1 --frames 299 --sampler ddim50 --no-cpu-baseline --no-alt --no-side --no-full-loop --no-clock --steps 2 --warmup 1 --no-graph --profile-steps 0"
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  MMDM_NO_OVERLAP=1 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_fp32/pmc_$tag -- python3 $P > $O/pmc_$tag.json 2> $O/pmc_$tag.err
done
cd $R
PMC_BATCH=1 PMC_FRAMES=299 python3 tools/pmc_summary.py $O/pmc_fp32 $O/summary_fp32 fp32 > $O/summary.log 2>&1
python3 tools/pmc_summary.py $O $O/summary fp32 traces-only >> $O/summary.log 2>&1
head -14 $O/summary/kernel_stats_serial.csv; cat $O/summary_fp32/pmc_summary.json | head -60
