#!/bin/bash
# tools/sweep_env.sh VAR v1 v2 ... : the benchmark step under each value of an environment switch, two rounds
VAR=$1; shift
for round in 1 2; do
  for v in "$@"; do
    echo -n "$VAR=$v  "; env $VAR=$v python3 tools/step_bench.py --grid 192 64 48 --steps 20 --warmup 5 2>/dev/null | tail -1
  done
done
