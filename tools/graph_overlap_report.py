"""Reads the rocpd database of `rocprofv3 --kernel-trace -- python3 tools/graph_overlap_probe.py`: per half (eager / captured),
the last 4 steps' span, the sum of kernel durations, the time during which >= 2 kernels ran, and the same restricted to the
weight-gradient kernels."""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end from kernels order by start").fetchall()
mark = [i for i, r in enumerate(rows) if "cumsum" in r[0].lower() or "scan" in r[0].lower()]
assert mark, "marker kernel not found"
halves = {"eager": rows[: mark[0]], "captured": rows[mark[-1] + 1:]}
for name, seg in halves.items():
    opt = [i for i, r in enumerate(seg) if "opt_radam" in r[0]]
    seg = seg[opt[-5] + 1: opt[-1] + 1]  # the last 4 steps
    t0, t1 = seg[0][1], max(r[2] for r in seg)
    ev = sorted([(r[1], 1) for r in seg] + [(r[2], -1) for r in seg])
    depth, last, both, busy = 0, t0, 0, 0
    for t, d in ev:
        if depth >= 1:
            busy += t - last
        if depth >= 2:
            both += t - last
        depth += d
        last = t
    wg = [r for r in seg if "wgrad" in r[0]]
    print(f"{name:9s}: {(t1 - t0) / 4e6:6.2f} ms per step, kernel time {sum(r[2] - r[1] for r in seg) / 4e6:6.2f}, busy {busy / 4e6:6.2f}, "
          f">= 2 kernels at once {both / 4e6:5.2f} ms; weight-gradient kernels {sum(r[2] - r[1] for r in wg) / 4e6:5.2f} ms in {len(wg) // 4} launches")
