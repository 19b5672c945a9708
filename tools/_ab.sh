#!/bin/bash
Q="--no-cpu-baseline --no-alt --no-full-loop --no-clock --steps 20 --warmup 5"
for i in 1 2; do
  for lib in old new; do
    if [ $lib = old ]; then export MMDM_LIB=$PWD/build/libmmdm_old.so; else unset MMDM_LIB; fi
    echo "$lib $(python bench.py $Q 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["frac"])')"
    echo "$lib one-stream $(MMDM_NO_OVERLAP=1 python bench.py $Q 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')"
  done
done
