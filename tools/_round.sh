#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r05 > gpurun_out/round_profile.log 2>&1
cd $GRAFT_REPO_ROOT
bash tools/profile_b1.sh r05 > gpurun_out/round_b1.log 2>&1
cd $GRAFT_REPO_ROOT
bash tools/bench_round.sh r05 > gpurun_out/round_bench.log 2>&1
tail -30 gpurun_out/round_bench.log
