#!/bin/bash
# Round profile (GPU box): kernel traces + PMC passes of the default bench workload at HEAD, for the headline fp32 mode and for the
# fp32_split / bf16 / bf16_fp8 modes.  Output: gpurun_out/prof_$TAG/*  (then: python tools/collect_profiles.py $TAG, here).
# usage: tools/profile_round.sh [tag, default r05]   (run from the repo root on the GPU box; ~10 minutes)
TAG=${1:-r06}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --no-cpu-baseline --no-alt --no-side --no-full-loop --no-clock"
MMDM_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial -- python3 $B --steps 6 --warmup 2 > $O/serial.json 2> $O/serial.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/overlap -- python3 $B --steps 6 --warmup 2 > $O/overlap.json 2> $O/overlap.err
for p in fp32_split bf16 bf16_fp8; do
  MMDM_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${p}_serial -- python3 $B --precision $p --steps 6 --warmup 2 > $O/${p}_serial.json 2> $O/${p}_serial.err
done
MMDM_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/single_serial -- python3 $B --workload single --steps 6 --warmup 2 > $O/single_serial.json 2> $O/single_serial.err
# PMC passes: counters in their own runs (kernel trace only beside them), one set per pass (TCC: FETCH_SIZE and WRITE_SIZE do not fit together)
for p in fp32 fp32_split bf16_fp8; do
  P="$B --precision $p --steps 2 --warmup 1 --no-graph --profile-steps 0"
  for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    tag=$(echo $set | cut -d' ' -f1)
    MMDM_NO_OVERLAP=1 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_${p}/pmc_$tag -- python3 $P > $O/pmc_${p}_$tag.json 2> $O/pmc_${p}_$tag.err
  done
done
cd $R
for p in fp32 fp32_split bf16_fp8; do python3 tools/pmc_summary.py $O/pmc_$p $O/summary_$p $p; done
python3 tools/pmc_summary.py $O $O/summary fp32 traces-only
