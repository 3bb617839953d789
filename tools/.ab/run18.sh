cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r18
for v in new n1 n2 n3 n7; do
  echo "== $v"
  if [ $v = new ]; then unset TDX_LIB; else export TDX_LIB=tools/.ab/libtdx_$v.so; fi
  python3 tools/conv_bench.py --dtype f32 --impl split --no-wgrad --layers down.1.b2,up.2.b1,up.3.b1,up.3.b2 2>&1 | grep -v "^$\|amdgpu.ids"
done > gpurun_out/r18/split_v2_abl.log 2>&1; cat gpurun_out/r18/split_v2_abl.log
