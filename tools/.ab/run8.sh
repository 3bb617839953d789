cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r8
for cfg in "" "160,96" "192,64" "128,128" "224,32" "256,128" "256,64" "160,64"; do echo -n "split [$cfg]  "; TDX_EXP_SPLIT_CUS=$cfg python3 tools/step_bench.py --grid 192 64 48 --steps 20 --warmup 5 2>/dev/null | tail -1; done > gpurun_out/r8/split.log 2>&1
echo -n "split []  " >> gpurun_out/r8/split.log; python3 tools/step_bench.py --grid 192 64 48 --steps 20 --warmup 5 2>/dev/null | tail -1 >> gpurun_out/r8/split.log
cat gpurun_out/r8/split.log
