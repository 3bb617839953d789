#!/bin/bash
# two bench.py ranks on ONE device over gloo, started by hand with faulthandler: where does a hang sit?
export TDX_BENCH_BACKEND=gloo TDX_BENCH_ONE_DEVICE=1 HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=${PORT:-29611} WORLD_SIZE=2
ARGS="--gpus 2 --steps 2 --warmup 1 --batch 1 --no-cpu-baseline --sample-steps 3 --sample-batch 1"
for r in 0 1; do
  RANK=$r LOCAL_RANK=$r timeout -s ABRT ${TMO:-240} python3 -X faulthandler bench.py $ARGS > gpurun_out/dbg_rank$r.out 2> gpurun_out/dbg_rank$r.err &
done
wait
for r in 0 1; do echo "== rank $r"; tail -c 300 gpurun_out/dbg_rank$r.out; grep -v "amdgpu.ids" gpurun_out/dbg_rank$r.err | tail -40; done
