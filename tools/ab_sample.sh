#!/bin/bash
# same-box A/B of an environment switch on the sampling step: tools/ab_sample.sh VAR v0 v1 [pairs] [extra sample_bench args]
VAR=$1; A=$2; B=$3; N=${4:-3}; shift 4 2>/dev/null
for i in $(seq $N); do
  for v in $A $B; do
    echo -n "$VAR=$v  "; env $VAR=$v python3 tools/sample_bench.py --steps 200 "$@" 2>/dev/null | tail -1
  done
done
