cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2
(timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_model.py -x -q -m gpu -k "attention or attn" 2>&1 | tail -15) > gpurun_out/r2/tests.log 2>&1
for d in f16 bf16; do python tools/attn_bench.py --dtype $d; TDX_ATTN_BOUND=0 python tools/attn_bench.py --dtype $d | grep fwd | sed 's/^/   [always checking] /'; done > gpurun_out/r2/attn.log 2>&1
tail -20 gpurun_out/r2/tests.log; cat gpurun_out/r2/attn.log
