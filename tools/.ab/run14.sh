cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r14
for i in 1 2 3; do
echo -n "slp    "; python3 tools/step_bench.py --grid 192 64 48 --steps 20 --warmup 5 2>/dev/null | tail -1
echo -n "noslp  "; TDX_LIB=tools/.ab/libtdx_noslp.so python3 tools/step_bench.py --grid 192 64 48 --steps 20 --warmup 5 2>/dev/null | tail -1
done > gpurun_out/r14/ab_noslp_step.log 2>&1; cat gpurun_out/r14/ab_noslp_step.log
for i in 1 2; do
echo -n "slp   B8 "; python3 tools/sample_bench.py --steps 200 --trajectories 8 2>/dev/null | grep -o '"ms_per_reverse_step": [0-9.]*'
echo -n "noslp B8 "; TDX_LIB=tools/.ab/libtdx_noslp.so python3 tools/sample_bench.py --steps 200 --trajectories 8 2>/dev/null | grep -o '"ms_per_reverse_step": [0-9.]*'
echo -n "slp   B1 "; python3 tools/sample_bench.py --steps 200 --trajectories 1 2>/dev/null | grep -o '"ms_per_reverse_step": [0-9.]*'
echo -n "noslp B1 "; TDX_LIB=tools/.ab/libtdx_noslp.so python3 tools/sample_bench.py --steps 200 --trajectories 1 2>/dev/null | grep -o '"ms_per_reverse_step": [0-9.]*'
done > gpurun_out/r14/ab_noslp_sample.log 2>&1; cat gpurun_out/r14/ab_noslp_sample.log
echo -n "slp   f32s "; python3 tools/step_bench.py --grid 192 64 48 --steps 8 --warmup 3 --mode f32s 2>/dev/null | tail -1
echo -n "noslp f32s "; TDX_LIB=tools/.ab/libtdx_noslp.so python3 tools/step_bench.py --grid 192 64 48 --steps 8 --warmup 3 --mode f32s 2>/dev/null | tail -1
