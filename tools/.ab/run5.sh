cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5
(timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_model.py tests/test_data_pipeline.py tests/test_parallel_gpu.py -x -q -m gpu -k "radam or RAdam or optim or three_training or graphed or trainer or two_ranks" 2>&1 | tail -5) > gpurun_out/r5/tests.log 2>&1
tail -5 gpurun_out/r5/tests.log
timeout 1500 python bench.py --no-cpu-baseline --no-torch-baseline --no-parity-modes > gpurun_out/r5/bench.json 2> gpurun_out/r5/bench.err
tail -3 gpurun_out/r5/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5/bench.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "frac", d.get("roofline", {}).get("frac"))
e = d["extra"]
print("captured_step", json.dumps(e.get("captured_step")))
print("cfg1", json.dumps({m: r["captured_step"] for m, r in e["cfg1"]["modes"].items()})[:1500])
PY
