#!/bin/bash
export MMDM_LIB=$PWD/build/libmmdm_noserial.so
mkdir -p gpurun_out
{
for pr in "fp32 fp32_split" "bf16 bf16" "fp32_split fp32_split" "bf16_fp8 bf16_fp8" "fp32 bf16_fp8"; do
  echo "== steps $pr"; NSTEP=24 timeout 400 python tools/overlap_bisect.py $pr 2>&1 | grep -v amdgpu.ids | cut -c1-250 | tail -6
done
echo "== whole calls, 2 handles, bf16"; NPOOL=2 PRECAPTURE=1 timeout 300 python tools/handle_overlap_bits.py bf16 2>&1 | grep -v amdgpu.ids
echo "== whole calls, 4 handles, fp32_split"; NPOOL=4 PRECAPTURE=1 timeout 300 python tools/handle_overlap_bits.py fp32_split 2>&1 | grep -v amdgpu.ids
} > gpurun_out/ob20.log 2>&1
cat gpurun_out/ob20.log | tail -60
