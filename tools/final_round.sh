#!/bin/bash
# Everything profiles/ quotes for a round, in ONE gpurun call (one box, one build): kernel traces + PMC passes (headline, the other modes, the B = 1 call, a ragged
# batch, configs[4]'s B = 64 shard, the stand-alone fp8 GEMM in both forms), the sustained-rate and in-step A/B probes of round 6, then every bench line.
# usage (GPU box, repo root): tools/final_round.sh [tag, default r06]     (~45 minutes; then here: python tools/collect_profiles.py $TAG; python tools/collect_bench.py $TAG)
TAG=${1:-r06}
cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh $TAG > gpurun_out/profile_round_$TAG.log 2>&1
bash tools/profile_b1.sh $TAG > gpurun_out/profile_b1_$TAG.log 2>&1
bash tools/profile_ragged.sh $TAG > gpurun_out/profile_ragged_$TAG.log 2>&1
bash tools/profile_fp8_b64.sh $TAG > gpurun_out/profile_fp8_b64_$TAG.log 2>&1
FP8_ONLY=packed-t FP8_SHAPES=qkv,ffn2 bash tools/pmc_fp8.sh $TAG > gpurun_out/pmc_fp8_$TAG.log 2>&1
FP8_ONLY=persist FP8_SHAPES=qkv bash tools/pmc_fp8.sh ${TAG}_persist > gpurun_out/pmc_fp8_${TAG}_persist.log 2>&1
O=gpurun_out/probes_$TAG; mkdir -p $O
[ -x variants/pk8 ] && ./variants/pk8 > $O/mfma_sustained.txt 2>&1
python tools/gemm_fp8_bench.py 2>&1 | grep -v amdgpu.ids > $O/gemm_fp8_bench.txt
FP8_M=76800 FP8_SHAPES=qkv,caq,cakv,ffn1 python tools/gemm_fp8_bench.py 2>&1 | grep -v amdgpu.ids > $O/gemm_fp8_bench_b64.txt
python tools/fp8_step_ab.py 2>&1 | grep -v amdgpu.ids > $O/fp8_step_ab.txt
AB_BATCH=64 python tools/fp8_step_ab.py 2>&1 | grep -v amdgpu.ids > $O/fp8_step_ab_b64.txt
python tools/attn_planes_bench.py 2>&1 | grep -v amdgpu.ids > $O/attn_planes_bench.txt
FP8=1 WARM=100 python tools/bf16w_timeline.py 2>&1 | grep -v amdgpu.ids > $O/bf16w_timeline_fp8.txt
bash tools/bench_round.sh $TAG > gpurun_out/bench_round_$TAG.log 2>&1
tail -25 gpurun_out/bench_round_$TAG.log
