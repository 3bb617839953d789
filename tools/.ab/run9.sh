cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r9
bash tools/ab_sample.sh TDX_CONV3_RING 1 2 2 --trajectories 1 > gpurun_out/r9/ab_ring_b1.log 2>&1
python3 - <<'PY' >> gpurun_out/r9/ab_ring_b1.log
PY
grep -o 'TDX_CONV3_RING=[0-9]\|"ms_per_reverse_step": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/r9/ab_ring_b1.log | paste - - | head; head -c 600 gpurun_out/r9/ab_ring_b1.log
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r9/st -- python3 $GRAFT_REPO_ROOT/tools/sample_bench.py --trajectories 1 --steps 50 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r9/st/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print("total kernel ms per step ~", tot / 52 / 1e6)
for r in rows[:32]:
    print(f'{int(r["Calls"])/52:6.1f} {int(r["TotalDurationNs"])/52/1e3:8.1f} us/step {float(r["AverageNs"])/1e3:7.1f} avg  {r["Name"][:84]}')
PY
rm -rf gpurun_out/r9/st
