#!/usr/bin/env python3
"""Summarise gpurun_out/prof_<tag>/ (tools/collect_profiles.sh) into profiles/<tag>_*.
  <tag>_kernel_stats.csv   rocprofv3 --stats table (per-kernel calls / total / average)
  <tag>_summary.md         human-readable per-step table + counters of the conv kernels
  <tag>_traffic.json       per-launch HBM bytes of the dominant kernel (read by bench.py)
FETCH_SIZE / WRITE_SIZE are reported in KB by rocprofv3; on gfx950 FETCH_SIZE counts 64 B per
128-B request for wide coalesced reads, so it is doubled (MI355X_MICROARCH.md, HBM section)."""
import csv, glob, json, os, shutil, sys, collections
from pathlib import Path

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4  # 1 warm-up + 3 timed
mode = sys.argv[3] if len(sys.argv) > 3 else "bf16"   # what TDX_BENCH_ARGS selected in collect_profiles.sh
root = Path(__file__).resolve().parent.parent
src = root / "gpurun_out" / f"prof_{tag}"
dst = root / "profiles"
dst.mkdir(exist_ok=True)

def one(pattern):
    # the newest match: gpurun_out/ on the build host accumulates the files of every collection under the same names'
    # directory (on the GPU box, where tools/r11_profiles.sh runs this, there is exactly one)
    f = glob.glob(str(src / pattern))
    return max(f, key=os.path.getmtime) if f else None

stats = one("stats/*/*kernel_stats.csv")
rows = list(csv.DictReader(open(stats)))
shutil.copy(stats, dst / f"{tag}_kernel_stats.csv")
tot = sum(float(r["TotalDurationNs"]) for r in rows)

def counter_avgs(pattern):
    f = one(pattern)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    if not f:
        return agg
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg

fetch, write, sq = counter_avgs("fetch/*/*counter_collection.csv"), counter_avgs("write/*/*counter_collection.csv"), counter_avgs("sq/*/*counter_collection.csv")
mean = lambda v: sum(v) / max(len(v), 1)
lines = [f"# rocprofv3 summary `{tag}` — `python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra` (B = 6, {mode}, 192x64x48)", "",
         f"Total kernel time {tot/1e6/steps:.2f} ms per step ({steps} steps incl. warm-up in the trace).  Collected with "
         + ("`TDX_WGRAD_STREAM=0` (tools/collect_profiles.sh): every kernel runs alone, so durations and counters are its own; the "
            "product overlaps the weight gradients with the data-gradient chain on a side stream (DESIGN 3.4) and its step is "
            "shorter than this sum." if os.environ.get("TDX_PROFILE_SIDE_STREAM", "0") == "0" else
            "the product's default `TDX_WGRAD_STREAM=1`: the weight gradients run on a side stream beside the data-gradient "
            "chain, so their durations (and those of what runs beside them) cover overlapped time and the sum exceeds the step."), "",
         "| kernel | calls/step | ms/step | avg µs | % |", "|---|---|---|---|---|"]
for r in rows[:28]:
    lines.append(f"| `{r['Name'][:80]}` | {int(r['Calls'])/steps:.1f} | {float(r['TotalDurationNs'])/1e6/steps:.3f} | {float(r['AverageNs'])/1e3:.1f} | {float(r['Percentage']):.1f} |")
lines += ["", "## HBM traffic and SQ counters of the matrix-core conv kernels (averages per launch)", "",
          "| kernel | FETCH_SIZE KB (raw) | read MB (x2 gfx950 correction) | WRITE_SIZE KB -> MB | MFMA insts | MFMA busy cyc | LDS conflict / active | clock GHz (GRBM/8/dur) |", "|---|---|---|---|---|---|---|---|"]
traffic = {}
durs = {r["Name"]: float(r["AverageNs"]) for r in rows}
for k in sorted(set(list(fetch) + list(write))):
    # every matrix-core conv kernel of the library: forward / data gradient (brick, ring, small-grid, split, fp32), the
    # halo shell, and all weight-gradient kernels (brick: conv3_wgrad_mfma*, producer / consumer: conv3_wgrad_ring_kernel,
    # packed K: conv3_wgrad_small_kernel -- the last two are not templates and were missed by the round-3 filter)
    if not any(t in k for t in ("conv3_ring_kernel", "conv3_small_kernel", "conv3_mfma_kernel", "conv3_wgrad", "conv3_mfma_split",
                                "conv3_mfma_f32", "conv1_f32_mfma", "conv1_mfma", "conv3_shell", "convg_")):
        continue
    f_kb = mean(fetch[k].get("FETCH_SIZE", [0])); w_kb = mean(write[k].get("WRITE_SIZE", [0]))
    rd, wr = 2 * f_kb * 1024, w_kb * 1024
    s = sq.get(k, {})
    dur = durs.get(k, 0)
    clk = mean(s.get("GRBM_GUI_ACTIVE", [0])) / 8 / dur if dur else 0
    conf, act = mean(s.get("SQ_LDS_BANK_CONFLICT", [0])), mean(s.get("SQ_LDS_IDX_ACTIVE", [1]))
    lines.append(f"| `{k[:60]}` | {f_kb:.0f} | {rd/1e6:.1f} | {w_kb:.0f} -> {wr/1e6:.1f} | {mean(s.get('SQ_INSTS_MFMA',[0])):.3g} | {mean(s.get('SQ_VALU_MFMA_BUSY_CYCLES',[0])):.3g} | {conf/act:.3f} | {clk:.2f} |")
    traffic[k] = {"read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "launches_sampled": len(fetch[k].get("FETCH_SIZE", []))}
fw = {k: v for k, v in traffic.items() if "conv3_mfma_kernel" in k and "false" in k.lower().replace("lb0", "false")}
(dst / f"{tag}_summary.md").write_text("\n".join(lines) + "\n")
sys.path.insert(0, str(root / "generative-turbulence_amd"))
from turbdiff_amd._lib import kernel_sources_fingerprint  # noqa: E402
import os, subprocess  # noqa: E402

head = subprocess.run(["git", "-C", str(root), "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip()
if not head and (root / "tools" / ".git_head").exists():  # the GPU box has no .git: the commit the snapshot was taken from
    head = (root / "tools" / ".git_head").read_text().strip() + " (+ uncommitted changes, if any, of the snapshot)"
json.dump({"source": f"profiles/{tag}_summary.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH_SIZE doubled per MI355X_MICROARCH.md)",
           # what the counters describe: bench.py refuses this file once the conv kernels' sources differ from these
           "git_head": head or None, "kernel_sources_sha16": kernel_sources_fingerprint(),
           "side_stream": os.environ.get("TDX_PROFILE_SIDE_STREAM", "0"),
           "kernels": traffic}, open(dst / f"{tag}_traffic.json", "w"), indent=1)
print("\n".join(lines[:40]))
