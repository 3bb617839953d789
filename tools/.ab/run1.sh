cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r1
(timeout 1500 python -m pytest tests/test_hip_model.py -x -q -m gpu -k "config3 or 194x50x50_benchmark or graph_samplers or constructor_options or default_path or eager_switch" 2>&1 | tail -15) > gpurun_out/r1/tests.log 2>&1
(timeout 600 python -m pytest tests/test_parallel_gloo.py tests/test_data_pipeline.py tests/test_hip_ops.py -x -q -m gpu -k "stager or conv1 or fused or attention" 2>&1 | tail -8) >> gpurun_out/r1/tests.log 2>&1
(timeout 300 python tools/host_profile.py 2>&1 | head -60) > gpurun_out/r1/host.log 2>&1
(bash tools/ab_lib.sh tools/.ab/libtdx_r4.so 3) > gpurun_out/r1/ab_m0.log 2>&1
tail -30 gpurun_out/r1/tests.log; head -5 gpurun_out/r1/host.log; cat gpurun_out/r1/ab_m0.log
