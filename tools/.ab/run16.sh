cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r16
for v in base abl1 abl2 abl3; do
  echo "== $v"
  if [ $v = base ]; then unset TDX_LIB; else export TDX_LIB=tools/.ab/libtdx_$v.so; fi
  python3 tools/conv_bench.py --dtype f32 --impl split --no-wgrad --layers down.1.b1,down.1.b2,up.2.b1,up.2.b2,up.3.b1,up.3.b2 2>&1 | grep -v "^$"
done > gpurun_out/r16/split_ablation.log 2>&1; cat gpurun_out/r16/split_ablation.log
