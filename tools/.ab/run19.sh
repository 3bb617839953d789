R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r19; cd /tmp; export TMPDIR=/tmp
for v in ${VARIANTS:-prev new n1 n7}; do
  if [ $v = new ]; then unset TDX_LIB; else export TDX_LIB=$R/tools/.ab/libtdx_$v.so; fi
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d /tmp/pmc_$v -- python3 $R/tools/conv_bench.py --dtype f32 --impl split --no-wgrad --layers down.1.b2,up.3.b1 > /tmp/pmc_$v.log 2>&1
  echo "== $v"; python3 $R/tools/kernel_counters.py $(ls /tmp/pmc_$v/*/*counter_collection.csv | head -1) split
done > $R/gpurun_out/r19/split_counters.log 2>&1
cat $R/gpurun_out/r19/split_counters.log
