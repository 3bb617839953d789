#!/usr/bin/env python3
"""Side-by-side per-kernel table of two rocprofv3 `--kernel-trace --stats` CSVs (tools/prof_step.sh), per step.
    python tools/compare_kernel_stats.py A.csv B.csv --steps 6 --scale-a 485000 --scale-b 589824 --labels real bench
--scale-*: work units (voxels, or batch) of each run: the last columns give B's time per unit over A's and, with
--fit Ba Bb (batch sizes), the batch-independent part f of a two-point fit t = f + s B."""
import argparse, csv

ap = argparse.ArgumentParser()
ap.add_argument("a"); ap.add_argument("b")
ap.add_argument("--steps", type=int, default=6, help="steps in each trace (warm-up included)")
ap.add_argument("--scale-a", type=float, default=1.0); ap.add_argument("--scale-b", type=float, default=1.0)
ap.add_argument("--labels", nargs=2, default=["A", "B"])
ap.add_argument("--fit", nargs=2, type=float, default=None)
ap.add_argument("--top", type=int, default=60)
a = ap.parse_args()


def load(f):
    return {r["Name"]: (int(r["Calls"]) / a.steps, float(r["TotalDurationNs"]) / 1e6 / a.steps) for r in csv.DictReader(open(f))}


A, B = load(a.a), load(a.b)
names = sorted(set(A) | set(B), key=lambda n: -(A.get(n, (0, 0))[1] + B.get(n, (0, 0))[1]))
ta, tb = sum(v[1] for v in A.values()), sum(v[1] for v in B.values())
print(f"kernel time per step: {a.labels[0]} {ta:.2f} ms ({sum(v[0] for v in A.values()):.0f} launches), "
      f"{a.labels[1]} {tb:.2f} ms ({sum(v[0] for v in B.values()):.0f} launches); per unit {a.labels[0]} / {a.labels[1]} = "
      f"{(ta / a.scale_a) / (tb / a.scale_b):.3f}")
hdr = f"{'kernel':68s} {a.labels[0]:>6s}: calls     ms | {a.labels[1]:>6s}: calls     ms | per-unit ratio"
if a.fit:
    hdr += " | fixed part (ms)"
print(hdr)
fixed = 0.0
for n in names[: a.top]:
    ca, ma = A.get(n, (0, 0)); cb, mb = B.get(n, (0, 0))
    r = (ma / a.scale_a) / (mb / a.scale_b) if mb > 0 and ma > 0 else float("nan")
    line = f"{n[:68]:68s} {ca:13.1f} {ma:6.3f} | {cb:13.1f} {mb:6.3f} | {r:6.2f}"
    if a.fit:
        s = (mb - ma) / (a.fit[1] - a.fit[0]); f = ma - a.fit[0] * s
        fixed += max(f, 0.0)
        line += f" | {f:7.3f}"
    print(line)
if a.fit:
    print(f"sum of the positive fixed parts over the kernels listed: {fixed:.2f} ms")
