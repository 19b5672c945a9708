#!/bin/bash
# Round bench lines (GPU box): every bench.py line profiles/README.md quotes, at HEAD.  Output: gpurun_out/bench_$TAG/*.json
# (then: python tools/collect_bench.py $TAG, here).  usage: tools/bench_round.sh [tag, default r05]   (~22 minutes)
TAG=${1:-r06}
O=gpurun_out/bench_$TAG; rm -rf $O; mkdir -p $O
# roofline.traffic needs the PMC traffic files of THESE sources: take them from a profile round that ran in the same call (tools/profile_round.sh)
for pm in fp32: fp32_split:_fp32_split bf16_fp8:_bf16_fp8; do
  f=gpurun_out/prof_$TAG/summary_${pm%%:*}/gemm_traffic.json
  [ -f $f ] && cp $f profiles/gemm_traffic${pm##*:}.json
done
f=gpurun_out/prof_b1_$TAG/summary_fp32/gemm_traffic.json; [ -f $f ] && cp $f profiles/gemm_traffic_b1t299.json       # tools/profile_b1.sh of the same call
f=gpurun_out/prof_ragged_$TAG/summary_pmc/gemm_traffic.json; [ -f $f ] && cp $f profiles/gemm_traffic_ragged.json        # tools/profile_ragged.sh of the same call
Q="--no-cpu-baseline --no-alt --no-side --no-full-loop"
python bench.py > $O/default.json 2> $O/default.err                                    # the driver's command: headline + fp32_split + full loop + CPU port
python bench.py --workload single $Q > $O/single.json 2> $O/single.err                  # configs[1]
python bench.py --precision bf16 $Q > $O/bf16.json 2> $O/bf16.err
python bench.py --precision bf16_fp8 $Q > $O/bf16_fp8.json 2> $O/bf16_fp8.err           # configs[4] arithmetic at B = 16
python bench.py --precision bf16_fp8 --batch 64 $Q > $O/fp8_b64.json 2> $O/fp8_b64.err  # configs[4] per-GPU shard
python bench.py --batch 32 $Q > $O/b32.json 2> $O/b32.err                               # configs[3] per-GPU shard
python bench.py --batch 32 --precision fp32_split $Q > $O/b32_split.json 2> $O/b32_split.err
# the reference's own call shapes (src/scripts/infer/mixermdm.py:73,117-124; src/evaluation/datasets.py:58,100-116): B = 1 and B = 15, T = 299, ddim50, through the
# facade (one whole MixerMDM.forward / forward_test each), the CPU port timing the SAME loop in full; and configs[0] (single-person, B = 1, T = 120, ddim50)
python bench.py --batch 1 --frames 299 --sampler ddim50 --facade --steps 20 --warmup 3 > $O/infer_b1.json 2> $O/infer_b1.err
python bench.py --batch 15 --frames 299 --sampler ddim50 --facade --steps 20 --warmup 3 --no-cpu-baseline > $O/infer_b15.json 2> $O/infer_b15.err
python bench.py --workload single --batch 1 --frames 120 --sampler ddim50 --steps 20 --warmup 3 > $O/configs0.json 2> $O/configs0.err
# the reference's evaluation caller (src/evaluation/datasets.py:100-116): 64 items of their own lengths as the sequential loop, in-flight handles and ragged batches
python bench.py --eval-items 64 > $O/eval64.json 2> $O/eval64.err
python bench.py --eval-items 64 --precision fp32_split --no-cpu-baseline > $O/eval64_split.json 2> $O/eval64_split.err
python tools/full_loop.py > $O/full_loops.txt 2> $O/full_loops.err
for f in $O/*.json; do echo $(basename $f) $(grep -o '"ms_per_step": [0-9.]*' $f | head -1) $(grep -o '"value": [0-9.]*' $f | head -1); done
tail -6 $O/full_loops.txt
