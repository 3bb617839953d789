cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3
for v in NOEXP NOMAX NOLS; do echo -n "$v: "; TDX_LIB=tools/.ab/libtdx_$v.so python tools/attn_bench.py --dtype bf16 2>/dev/null | grep fwd; done > gpurun_out/r3/abl.log 2>&1
echo -n "full: " >> gpurun_out/r3/abl.log; python tools/attn_bench.py --dtype bf16 2>/dev/null | grep fwd >> gpurun_out/r3/abl.log
cat gpurun_out/r3/abl.log
