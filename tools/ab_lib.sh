#!/bin/bash
# same-box A/B of two builds of the library on the benchmark step: tools/ab_lib.sh <old .so> [pairs]
# (TDX_LIB selects the library a process loads; TDX_LIB_LAX=1 tolerates symbols the old build lacks)
OLD=$1; N=${2:-3}
for i in $(seq $N); do
  echo -n "old  "; TDX_LIB=$OLD TDX_LIB_LAX=1 python3 tools/step_bench.py --grid 192 64 48 --steps 20 --warmup 5 2>/dev/null | tail -1
  echo -n "new  "; python3 tools/step_bench.py --grid 192 64 48 --steps 20 --warmup 5 2>/dev/null | tail -1
done
