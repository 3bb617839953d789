#!/bin/bash
# kernel-time table of the sampling loop (eager, no graph, 12 reverse steps at B = 8)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/prof_sample
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/sample_bench.py --steps 12 --no-graph > $OUT.log 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("$OUT/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
d=collections.defaultdict(lambda:[0,0.0])
for r in rows:
    k=r["Kernel_Name"][:86]; d[k][0]+=1; d[k][1]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
n=14.0  # 2 warm-up + 12 timed steps
tot=sum(v[1] for v in d.values())/n
print(f"total kernel time {tot/1e3:.2f} ms/reverse step")
for k,(c,t) in sorted(d.items(),key=lambda kv:-kv[1][1])[:30]:
    print(f"{c/n:7.1f} {t/n:8.1f} us/step {t/c:8.1f} avg  {k}")
PY
