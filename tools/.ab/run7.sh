cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r7; R=$GRAFT_REPO_ROOT
(timeout 1200 python -m pytest tests/test_hip_model.py tests/test_hip_ops.py tests/test_data_pipeline.py -x -q -m gpu -k "not full_size and not benchmark_batch and not 194x50x50 and not config3" 2>&1 | tail -4) > gpurun_out/r7/tests.log 2>&1; tail -4 gpurun_out/r7/tests.log
for i in 1 2 3; do python3 tools/step_bench.py --grid 192 64 48 --steps 20 --warmup 5 2>/dev/null | tail -1; done
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r7/tl -- python3 $R/tools/step_bench.py --grid 192 64 48 --steps 6 --warmup 3 > $R/gpurun_out/r7/tl.log 2>&1
cd $R; python3 tools/timeline_gaps.py gpurun_out/r7/tl > gpurun_out/r7/timeline.txt 2>&1
head -16 gpurun_out/r7/timeline.txt
rm -rf gpurun_out/r7/tl
