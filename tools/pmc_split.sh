R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_split_pmc
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P="$R/bench.py --no-cpu-baseline --no-alt --no-full-loop --no-clock --precision fp32_split --steps 2 --warmup 1 --no-graph --profile-steps 0"
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  MMDM_NO_OVERLAP=1 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_$tag -- python3 $P > $O/pmc_$tag.json 2> $O/pmc_$tag.err
done
cd $R
python3 tools/pmc_summary.py $O
