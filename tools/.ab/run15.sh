cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r15
for i in 1 2 3; do
echo -n "slp   f32s "; python3 tools/step_bench.py --grid 192 64 48 --steps 8 --warmup 3 --mode f32s 2>/dev/null | tail -1
echo -n "noslp f32s "; TDX_LIB=tools/.ab/libtdx_noslp.so python3 tools/step_bench.py --grid 192 64 48 --steps 8 --warmup 3 --mode f32s 2>/dev/null | tail -1
done > gpurun_out/r15/ab_noslp_f32s.log 2>&1; cat gpurun_out/r15/ab_noslp_f32s.log
echo -n "slp   f32 "; python3 tools/step_bench.py --grid 192 64 48 --steps 4 --warmup 2 --mode f32 2>/dev/null | tail -1
echo -n "noslp f32 "; TDX_LIB=tools/.ab/libtdx_noslp.so python3 tools/step_bench.py --grid 192 64 48 --steps 4 --warmup 2 --mode f32 2>/dev/null | tail -1
TDX_LIB=tools/.ab/libtdx_noslp.so timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -4
