#!/bin/bash
# kernel-time table of the bench step (stats pass only); usage on the GPU box: bash tools/quick_profile.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-q}; OUT=$R/gpurun_out/prof_$TAG
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra $TDX_BENCH_ARGS > $OUT.log 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("$OUT/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
d=collections.defaultdict(lambda:[0,0.0])
for r in rows:
    k=r["Kernel_Name"][:86]; d[k][0]+=1; d[k][1]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
tot=sum(v[1] for v in d.values())/4
print(f"total kernel time {tot/1e3:.2f} ms/step")
for k,(c,t) in sorted(d.items(),key=lambda kv:-kv[1][1])[:42]:
    print(f"{c/4:7.1f} {t/4:8.1f} us/step {t/c:8.1f} avg  {k}")
PY
