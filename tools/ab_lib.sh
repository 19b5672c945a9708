#!/bin/bash
# A/B of two builds of the library ON ONE BOX (GPU box, from the repo root): bench.py lines alternating another build of the same ABI (MMDM_LIB selects
# it; variants/*.so travel with the snapshot, build/ does not) and the in-tree library.  The boxes of the pool differ by 1-4 %; this is how 'the box or the
# change?' is answered.   usage: tools/ab_lib.sh variants/libmmdm_other.so [precision ...]     (default: fp32, two-stream and one-stream)
OTHER=${1:-variants/libmmdm_old.so}; shift
MODES=${@:-fp32}
Q="--no-cpu-baseline --no-alt --no-side --no-full-loop --no-clock --steps 20 --warmup 5"
for m in $MODES; do
  for i in 1 2; do
    for lib in other tree; do
      if [ $lib = other ]; then export MMDM_LIB=$PWD/$OTHER; else unset MMDM_LIB; fi
      echo "$m $lib $(python bench.py $Q --precision $m 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["frac"], d["lib"])')"
      [ $m = fp32 ] && echo "$m $lib one-stream $(MMDM_NO_OVERLAP=1 python bench.py $Q --precision $m 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')"
    done
  done
done
unset MMDM_LIB
