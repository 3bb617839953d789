cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r12
timeout 1200 python bench.py > gpurun_out/r12/bench_default.json 2> gpurun_out/r12/bench_default.err; tail -2 gpurun_out/r12/bench_default.err
bash tools/r12_profiles.sh > gpurun_out/r12/profiles.log 2>&1; tail -5 gpurun_out/r12/profiles.log
timeout 900 python bench.py --no-cpu-baseline --no-torch-baseline --no-extra > gpurun_out/r12/bench_headline_after_profiles.json 2>/dev/null
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r12/bench_default.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "roofline", d.get("roofline"))
print("cpu", d.get("cpu_baseline"))
e = d["extra"]
print({k: (v if not isinstance(v, dict) else "...") for k, v in e.items()})
print("parity", {m: (v["ms_per_step"], v.get("rel_l2_vs_cpu_oracle")) for m, v in e["parity_modes"].items()})
print("sampling", e["sampling"]["ms_per_reverse_step"], e["sampling"]["public_p_sample_loop"])
print("real_grid", e["real_grid"]["modes"])
print("torch", e.get("torch_rocm_reference"))
PY
