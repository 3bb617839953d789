cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
(timeout 600 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "gn or groupnorm or resnet" 2>&1 | tail -3) > gpurun_out/r6/tests.log 2>&1; tail -3 gpurun_out/r6/tests.log
for i in 1 2 3; do
echo -n "batched  "; TDX_GN_BWD_PER_SAMPLE=0 python3 tools/step_bench.py --grid 192 64 48 --steps 20 --warmup 5 2>/dev/null | tail -1
echo -n "per-samp "; TDX_GN_BWD_PER_SAMPLE=1 python3 tools/step_bench.py --grid 192 64 48 --steps 20 --warmup 5 2>/dev/null | tail -1
echo -n "auto     "; python3 tools/step_bench.py --grid 192 64 48 --steps 20 --warmup 5 2>/dev/null | tail -1
done > gpurun_out/r6/ab_gn_per_sample.log 2>&1
cat gpurun_out/r6/ab_gn_per_sample.log
