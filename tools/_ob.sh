#!/bin/bash
mkdir -p gpurun_out
{
for v in plain nop0 nop1 nop3 nop7; do
  echo "== canary code object: $v"
  CANARY_HSACO=$PWD/build/canary_$v.hsaco CANARY_ONLY="split packed, default" CANARY_VARIANTS=1 CANARY_AGGR=1 CANARY_TRANS=1 MODE=canary NSTEP=3 timeout 60 python tools/overlap_bisect.py fp32 fp32 2>&1 | grep -v amdgpu.ids | sed 's/mismatches LDS.*checks; //; s/canary (512 workgroups, 32 KB LDS, 5000 us) //'
done
} > gpurun_out/ob21.log 2>&1
cat gpurun_out/ob21.log | cut -c1-220
