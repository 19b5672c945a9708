#!/bin/bash
# Round profile (GPU box): kernel traces + PMC passes of the default bench workload at HEAD.  Output: gpurun_out/prof_r02/*
# usage: tools/profile_round.sh   (run from the repo root on the GPU box; ~6 minutes)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r02
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --no-cpu-baseline --no-alt --no-full-loop --no-clock"
MMDM_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial -- python3 $B --steps 6 --warmup 2 > $O/serial.json 2> $O/serial.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/overlap -- python3 $B --steps 6 --warmup 2 > $O/overlap.json 2> $O/overlap.err
MMDM_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/split_serial -- python3 $B --precision fp32_split --steps 6 --warmup 2 > $O/split_serial.json 2> $O/split_serial.err
MMDM_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bf16_serial -- python3 $B --precision bf16 --steps 6 --warmup 2 > $O/bf16_serial.json 2> $O/bf16_serial.err
MMDM_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fp8_serial -- python3 $B --precision bf16_fp8 --steps 6 --warmup 2 > $O/fp8_serial.json 2> $O/fp8_serial.err
P="$B --steps 2 --warmup 1 --no-graph --profile-steps 0"
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  MMDM_NO_OVERLAP=1 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_$tag -- python3 $P > $O/pmc_$tag.json 2> $O/pmc_$tag.err
done
cd $R
python3 tools/pmc_summary.py $O
