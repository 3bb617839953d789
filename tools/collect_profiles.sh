#!/bin/bash
# Run on the GPU box (via gpurun): collects the rocprofv3 evidence that bench.py's roofline
# object refers to.  Output goes to gpurun_out/prof_<tag>/ ; summarise with
# tools/summarize_profiles.py and commit the summaries under profiles/.
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
# Per-kernel attribution needs kernels that run one after the other: the weight gradients go back onto the launching stream
# for the profiled runs (in the product they run on a side stream beside the data-gradient chain -- ops._WgradSide -- and a
# kernel trace then charges each of two concurrent kernels the whole overlapped span).  The forward launches that
# bench.py's roofline is quoted on have nothing beside them either way.
# TDX_PROFILE_SIDE_STREAM=1 collects the same set with the product's default (side stream on) for comparison; the
# summariser notes which of the two a set is.
export TDX_PROFILE_SIDE_STREAM=${TDX_PROFILE_SIDE_STREAM:-0}
export TDX_WGRAD_STREAM=${TDX_WGRAD_STREAM:-$TDX_PROFILE_SIDE_STREAM}
BENCH="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra ${TDX_BENCH_ARGS:-}"
# (1) per-kernel time
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH > $OUT/stats.log 2>&1
# (2)+(3) HBM traffic counters, one pass each (FETCH_SIZE and WRITE_SIZE do not fit one pass)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $BENCH > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $BENCH > $OUT/write.log 2>&1
# (4) SQ counters for the matrix-core kernels
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- $BENCH > $OUT/sq.log 2>&1
# (5) an un-profiled reference run of the same command
$BENCH > $OUT/bench_unprofiled.json 2> $OUT/bench_unprofiled.err
ls -R $OUT | head -40
