#!/bin/bash
# rocprofv3 per-kernel table of tools/step_bench.py on the GPU box: tools/prof_step.sh <tag> <step_bench args...>
# -> gpurun_out/<tag>_kernel_stats.csv (+ the un-profiled ms/step in gpurun_out/<tag>.txt)
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
export TDX_WGRAD_STREAM=${TDX_WGRAD_STREAM:-0}
python3 $R/tools/step_bench.py "$@" > $R/gpurun_out/$TAG.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/step_bench.py "$@" > $OUT/log.txt 2>&1
cp $(ls $OUT/*/*kernel_stats.csv | head -1) $R/gpurun_out/${TAG}_kernel_stats.csv
cat $R/gpurun_out/$TAG.txt
