#!/bin/bash
# same-box A/B of an environment switch on the benchmark step: tools/ab_step.sh VAR v0 v1 [pairs]
VAR=$1; A=$2; B=$3; N=${4:-3}
for i in $(seq $N); do
  for v in $A $B; do
    echo -n "$VAR=$v  "; env $VAR=$v python3 tools/step_bench.py --grid 192 64 48 --steps 20 --warmup 5 2>/dev/null | tail -1
  done
done
