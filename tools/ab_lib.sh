#!/bin/bash
# A/B of two builds of the library ON ONE BOX (GPU box, from the repo root): the default bench line, two-stream and one-stream, alternating
# build/libmmdm_old.so (any other build of the same ABI, e.g. a previous commit's: MMDM_LIB selects it) and the in-tree library.
# Round 5: the boxes of the pool differ by 1-4 %; this is how 'the box or the change?' was answered (59.82 / 60.06 vs 59.91 / 59.90 ms/step).
Q="--no-cpu-baseline --no-alt --no-full-loop --no-clock --steps 20 --warmup 5"
for i in 1 2; do
  for lib in old new; do
    if [ $lib = old ]; then export MMDM_LIB=$PWD/build/libmmdm_old.so; else unset MMDM_LIB; fi
    echo "$lib $(python bench.py $Q 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["frac"])')"
    echo "$lib one-stream $(MMDM_NO_OVERLAP=1 python bench.py $Q 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')"
  done
done
