#!/bin/bash
# round-6 evidence: fp32-MFMA conv path and the ingress/egress kernels.  GPU box: bash tools/r06_profiles.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06; mkdir -p $O
cd $R
python bench.py --steps 5 --warmup 2 > $O/bench.log 2>&1; tail -1 $O/bench.log > $O/r06_bench.json
python bench.py --dtype f32 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_f32.log 2>&1; tail -1 $O/bench_f32.log > $O/r06_bench_f32.json
python bench.py --dtype f32s --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_f32s.log 2>&1; tail -1 $O/bench_f32s.log > $O/r06_bench_f32s.json
python tools/conv_bench.py --dtype f32 > $O/r06_conv3_f32_table.txt 2>&1
python tools/conv_bench.py --dtype f32 --impl split > $O/r06_conv3_f32s_table.txt 2>&1
python tests/report_accuracy.py > $O/r06_accuracy.txt 2>&1
python tools/metrics_bench.py > $O/r06_metrics.txt 2>&1
python tools/gridio_bench.py > $O/r06_gridio.txt 2>&1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_gridio -- python3 $R/tools/gridio_bench.py > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_f32 -- python3 $R/bench.py --dtype f32 --steps 3 --warmup 1 --no-cpu-baseline --no-extra > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
def table(pattern, title, div, only=None, top=40):
    f = sorted(glob.glob(pattern, recursive=True))[-1]
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"][:96]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    out = [title]
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:top]:
        if only and not any(o in k for o in only): continue
        v = sorted(v)
        out.append(f"{len(v)/div:8.1f} calls  {sum(v)/div:10.1f} us total  median {v[len(v)//2]:9.1f} us  {k}")
    return "\n".join(out) + "\n"
open("$O/r06_gridio.txt", "a").write("\n" + table("$O/prof_gridio/**/*kernel_trace.csv", "# rocprofv3 --kernel-trace of the same command (kernel-only durations)", 1, ["grid_", "cell_embed", "sum_and_scatter", "krn_partial", "index_elementwise", "compute_grad_weight"], 400))
open("$O/r06_f32_kernel_table.txt", "w").write(table("$O/prof_f32/**/*kernel_trace.csv", "# rocprofv3 --kernel-trace --stats -- python3 bench.py --dtype f32 --steps 3 --warmup 1 (per step = /4)", 4))
PY
cp $(ls $O/prof_f32/*/*kernel_stats.csv | tail -1) $O/r06_f32_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_f32s -- python3 $R/bench.py --dtype f32s --steps 3 --warmup 1 --no-cpu-baseline --no-extra > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
f = sorted(glob.glob("$O/prof_f32s/**/*kernel_trace.csv", recursive=True))[-1]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[r["Kernel_Name"][:96]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = ["# rocprofv3 --kernel-trace --stats -- python3 bench.py --dtype f32s --steps 3 --warmup 1 (per step = /4)"]
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:40]:
    v = sorted(v); out.append(f"{len(v)/4:8.1f} calls  {sum(v)/4:10.1f} us total  median {v[len(v)//2]:9.1f} us  {k}")
open("$O/r06_f32s_kernel_table.txt", "w").write("\n".join(out) + "\n")
PY
cp $(ls $O/prof_f32s/*/*kernel_stats.csv | tail -1) $O/r06_f32s_kernel_stats.csv
ls -la $O
