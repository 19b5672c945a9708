#!/bin/bash
# What bounds the packed fp8 GEMM (round 6): variant builds of gemm_bf16.hip side by side on one box, stand-alone at M = 19 200.
#   base   the in-tree library                      exp1  -DMMDM_W_EXP=1: no epilogue (the K loops alone)
#   exp2   -DMMDM_W_EXP=2: no K loop (prologue + epilogue alone)     nopk  gemm_bf16.hip without packed-fp32 instructions
#   base + FP8_LDS_PAD: two workgroups per CU instead of three (what a 2 x accumulator design would have to live with)
# usage: tools/fp8_levers.sh      (GPU box, repo root; variants/libmmdm_{exp1,exp2,nopk}.so built in the container)
O=gpurun_out/fp8_levers; rm -rf $O; mkdir -p $O
S=${FP8_SHAPES:-qkv,caq,cakv,ffn1,ffn2}
for v in base exp1 exp2 nopk; do
  if [ $v = base ]; then unset MMDM_LIB; else export MMDM_LIB=$PWD/variants/libmmdm_$v.so; fi
  [ $v != base ] && [ ! -f "$MMDM_LIB" ] && continue
  FP8_ONLY=packed-t FP8_SHAPES=$S python tools/gemm_fp8_bench.py > $O/$v.txt 2>&1
  echo "== $v"; cat $O/$v.txt | grep -v amdgpu.ids
done
unset MMDM_LIB
FP8_LDS_PAD=30000 FP8_ONLY=packed-t FP8_SHAPES=$S python tools/gemm_fp8_bench.py > $O/base_2wg.txt 2>&1; echo "== base, 2 workgroups per CU"; grep -v amdgpu.ids $O/base_2wg.txt
FP8_CFG=12 FP8_ONLY=packed-t FP8_SHAPES=$S python tools/gemm_fp8_bench.py > $O/base_wide.txt 2>&1; echo "== base, 128 x 256 tiles"; grep -v amdgpu.ids $O/base_wide.txt
FP8=1 WARM=100 python tools/bf16w_timeline.py > $O/timeline.txt 2>&1; echo "== timeline"; grep -v amdgpu.ids $O/timeline.txt
