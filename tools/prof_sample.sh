#!/bin/bash
# rocprofv3 per-kernel table of tools/sample_bench.py on the GPU box: tools/prof_sample.sh <tag> <sample_bench args...>
# (eager launches -- --no-graph -- so that every kernel of the reverse step is attributed by name)
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
python3 $R/tools/sample_bench.py --steps 50 "$@" > $R/gpurun_out/$TAG.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/sample_bench.py --steps 50 --no-graph "$@" > $OUT/log.txt 2>&1
cp $(ls $OUT/*/*kernel_stats.csv | head -1) $R/gpurun_out/${TAG}_kernel_stats.csv
cat $R/gpurun_out/$TAG.txt
