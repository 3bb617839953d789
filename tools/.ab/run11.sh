cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r11
(timeout 1500 python -m pytest tests/test_hip_ops.py tests/test_hip_model.py tests/test_hip_fuzz.py -q -m gpu -k "not full_size and not benchmark_batch and not 194x50x50_bench and not config3" 2>&1 | tail -6) > gpurun_out/r11/tests.log 2>&1; tail -6 gpurun_out/r11/tests.log
bash tools/ab_sample.sh TDX_SMALL_STATS 0 1 3 --trajectories 1 2>&1 | grep -o 'TDX_SMALL_STATS=[0-9]\|"ms_per_reverse_step": [0-9.]*' | paste - - > gpurun_out/r11/ab_b1.log; cat gpurun_out/r11/ab_b1.log
bash tools/ab_sample.sh TDX_SMALL_STATS 0 1 2 --trajectories 8 2>&1 | grep -o 'TDX_SMALL_STATS=[0-9]\|"ms_per_reverse_step": [0-9.]*' | paste - - > gpurun_out/r11/ab_b8.log; cat gpurun_out/r11/ab_b8.log
bash tools/ab_step.sh TDX_SMALL_STATS 0 1 3 > gpurun_out/r11/ab_step.log 2>&1; cat gpurun_out/r11/ab_step.log
