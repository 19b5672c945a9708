#!/bin/bash
mkdir -p gpurun_out
CANARY_PK=1 CANARY_ONLY="split packed, default" CANARY_VARIANTS=1 CANARY_AGGR=1 MODE=canary NSTEP=4 timeout 70 python tools/overlap_bisect.py fp32 fp32 2>&1 | grep -v amdgpu.ids > gpurun_out/ob22.log
cat gpurun_out/ob22.log | cut -c1-260
