#!/usr/bin/env python3
"""Where does a step's wall time go that is NOT kernel time?  From a rocprofv3 --kernel-trace csv of tools/step_bench.py:
per step (delimited by the optimiser's opt_radam_kernel launches) the span, the union of all kernel intervals (device
busy), the idle remainder, and which kernels the longest idle gaps precede.
    rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/step_bench.py --grid 192 64 48 --steps 6 --warmup 3
    python3 tools/timeline_gaps.py OUT"""
import csv, glob, sys, collections

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))))
marks = [i for i, r in enumerate(rows) if r[2].startswith("void opt_radam_kernel") or "opt_radam_kernel" in r[2][:40]]
steps = [(marks[i] + 1, marks[i + 1] + 1) for i in range(len(marks) - 1)]
steps = steps[len(steps) // 2:]  # the timed (warm) half
tot_span = tot_busy = 0.0
gap_before = collections.defaultdict(lambda: [0, 0.0])
small = collections.defaultdict(lambda: [0, 0.0])
for a, b in steps:
    seg = rows[a:b]
    span = seg[-1][1] - rows[a - 1][1]
    busy, cur_end = 0, rows[a - 1][1]
    for s, e, name in seg:
        if s > cur_end:
            g = gap_before[name[:70]]
            g[0] += 1; g[1] += (s - cur_end) / 1e3
        if e > cur_end:
            busy += e - max(s, cur_end)
            cur_end = e
        if e - s < 8000:
            k = small[name[:70]]; k[0] += 1; k[1] += (e - s) / 1e3
    tot_span += span / 1e6; tot_busy += busy / 1e6
n = len(steps)
print(f"{n} steps: span {tot_span / n:.3f} ms/step, device busy (union of kernels) {tot_busy / n:.3f}, idle {(tot_span - tot_busy) / n:.3f}")
print(f"launches per step {sum(b - a for a, b in steps) / n:.0f}; kernels shorter than 8 us: "
      f"{sum(v[0] for v in small.values()) / n:.0f} per step, {sum(v[1] for v in small.values()) / n / 1e3:.3f} ms")
print("idle time by the kernel that FOLLOWS the gap (us per step, gaps per step):")
for k, (c, t) in sorted(gap_before.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"  {t / n:8.1f} us {c / n:6.1f}  {k}")
