#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 240 python tools/adaln_victim.py > gpurun_out/adaln_victim.log 2>&1
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
grep -E "passed|failed" gpurun_out/pytest_gpu.log | tail -2
bash tools/bench_round.sh r05 > gpurun_out/round_bench.log 2>&1
cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r05 > gpurun_out/round_profile.log 2>&1
cd $GRAFT_REPO_ROOT
bash tools/profile_b1.sh r05 > gpurun_out/round_b1.log 2>&1
cd $GRAFT_REPO_ROOT
grep -v amdgpu.ids gpurun_out/adaln_victim.log | tail -14
tail -22 gpurun_out/round_bench.log | cut -c1-200
