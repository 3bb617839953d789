cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r13
(timeout 300 python tools/host_profile.py 2>&1 | head -12) > gpurun_out/r13/host.log 2>&1; head -3 gpurun_out/r13/host.log
(timeout 900 python -m pytest tests/test_hip_model.py tests/test_parallel_gpu.py tests/test_data_pipeline.py -x -q -m gpu -k "sampler or scratch or stream or two_ranks or graphed or golden" 2>&1 | tail -3) > gpurun_out/r13/tests.log 2>&1; tail -3 gpurun_out/r13/tests.log
timeout 900 python bench.py --no-cpu-baseline --no-torch-baseline --no-parity-modes > gpurun_out/r13/bench.json 2>/dev/null
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r13/bench.json").read().strip().splitlines()[-1])
e = d["extra"]
print("ms_per_step", d["ms_per_step"])
print("captured_step", e["captured_step"]["eager"], e["captured_step"]["graph"])
for m, r in e["cfg1"]["modes"].items(): print("cfg1", m, r["ms_per_step"], r["captured_step"]["eager"]["ms_per_step"], r["captured_step"]["graph"]["ms_per_step"])
print("sampling", e["sampling"]["ms_per_reverse_step"], e["sampling"]["public_p_sample_loop"]["B1_ms_per_reverse_step"], e["sampling"]["public_p_sample_loop"]["B1_eager_ms_per_reverse_step"])
PY
