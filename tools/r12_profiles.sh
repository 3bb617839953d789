#!/bin/bash
# Round-5 evidence on the GPU box: rocprofv3 stats + PMC of the bf16 step (kernels one after the other, and with the
# product's side stream), of the f32s step, and the summaries copied under gpurun_out/ (commit them under profiles/).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
tools/collect_profiles.sh r12bf16 > gpurun_out/r12bf16_collect.log 2>&1
TDX_PROFILE_SIDE_STREAM=1 python3 tools/summarize_profiles.py r12bf16 4 bf16 > /dev/null 2>&1 || true
TDX_PROFILE_SIDE_STREAM=0 python3 tools/summarize_profiles.py r12bf16 4 bf16 > gpurun_out/r12bf16_summary_head.txt 2>&1
TDX_PROFILE_SIDE_STREAM=1 tools/collect_profiles.sh r12bf16ss > gpurun_out/r12bf16ss_collect.log 2>&1
TDX_PROFILE_SIDE_STREAM=1 python3 tools/summarize_profiles.py r12bf16ss 4 bf16 > gpurun_out/r12bf16ss_summary_head.txt 2>&1
TDX_BENCH_ARGS="--dtype f32s" tools/collect_profiles.sh r12f32s > gpurun_out/r12f32s_collect.log 2>&1
python3 tools/summarize_profiles.py r12f32s 4 f32s > gpurun_out/r12f32s_summary_head.txt 2>&1
mkdir -p gpurun_out/profiles_r12
cp profiles/r12* gpurun_out/profiles_r12/ 2>/dev/null
ls gpurun_out/profiles_r12
