#!/bin/bash
# Potential of overlapping matrix-bound and memory-bound phases of INDEPENDENT half-batches: two processes, each a
# training loop at half the batch, with the persistent kernels capped to CUS CUs, against one process at the full batch.
CUS=${1:-128}; B=${2:-3}; STEPS=${3:-80}
echo "== one process, B=$((2*B)), all CUs"
python3 tools/step_bench.py --grid 192 64 48 --batch $((2*B)) --steps 30 --warmup 5 2>/dev/null | tail -1
echo "== one process, B=$B, all CUs"
python3 tools/step_bench.py --grid 192 64 48 --batch $B --steps 30 --warmup 5 2>/dev/null | tail -1
for c in $CUS 256; do
  echo "== two concurrent processes, B=$B each, TDX_PERSISTENT_CUS=$c"
  TDX_PERSISTENT_CUS=$c python3 tools/step_bench.py --grid 192 64 48 --batch $B --steps $STEPS --warmup 20 2>/dev/null | tail -1 &
  TDX_PERSISTENT_CUS=$c python3 tools/step_bench.py --grid 192 64 48 --batch $B --steps $STEPS --warmup 20 2>/dev/null | tail -1 &
  wait
done
