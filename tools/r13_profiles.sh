#!/bin/bash
# Round-6 evidence on the GPU box: rocprofv3 stats + PMC (separate passes) of the bf16 step (kernels one after the other), of the
# fp16 step (new mode), of the f32s step and -- for its roofline.traffic, null until now -- of the f32 step; summaries land under
# profiles/ of the box's copy and are copied to gpurun_out/profiles_r13/ (commit them under profiles/).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
tools/collect_profiles.sh r13bf16 > gpurun_out/r13bf16_collect.log 2>&1
python3 tools/summarize_profiles.py r13bf16 4 bf16 > gpurun_out/r13bf16_summary_head.txt 2>&1
TDX_BENCH_ARGS="--dtype fp16" tools/collect_profiles.sh r13fp16 > gpurun_out/r13fp16_collect.log 2>&1
python3 tools/summarize_profiles.py r13fp16 4 fp16 > gpurun_out/r13fp16_summary_head.txt 2>&1
TDX_BENCH_ARGS="--dtype f32s" tools/collect_profiles.sh r13f32s > gpurun_out/r13f32s_collect.log 2>&1
python3 tools/summarize_profiles.py r13f32s 4 f32s > gpurun_out/r13f32s_summary_head.txt 2>&1
TDX_BENCH_ARGS="--dtype f32" tools/collect_profiles.sh r13f32 > gpurun_out/r13f32_collect.log 2>&1
python3 tools/summarize_profiles.py r13f32 4 f32 > gpurun_out/r13f32_summary_head.txt 2>&1
mkdir -p gpurun_out/profiles_r13
cp profiles/r13* gpurun_out/profiles_r13/ 2>/dev/null
ls gpurun_out/profiles_r13
