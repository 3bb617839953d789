cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r17
for v in prev new; do
  echo "== $v"
  if [ $v = new ]; then unset TDX_LIB; else export TDX_LIB=tools/.ab/libtdx_$v.so; fi
  python3 tools/conv_bench.py --dtype f32 --impl split --no-wgrad --layers down.1.b1,down.1.b2,up.2.b1,up.2.b2,up.3.b1,up.3.b2 2>&1 | grep -v "^$\|amdgpu.ids"
done > gpurun_out/r17/split_v2.log 2>&1; cat gpurun_out/r17/split_v2.log
unset TDX_LIB
timeout 1500 python -m pytest tests -q -m gpu -x -k "split or f32s or conv3" 2>&1 | tail -5
for i in 1 2; do
echo -n "prev f32s "; TDX_LIB=tools/.ab/libtdx_prev.so python3 tools/step_bench.py --grid 192 64 48 --steps 8 --warmup 3 --mode f32s 2>/dev/null | tail -1
echo -n "new  f32s "; python3 tools/step_bench.py --grid 192 64 48 --steps 8 --warmup 3 --mode f32s 2>/dev/null | tail -1
done
