#!/usr/bin/env python3
"""Per-kernel averages of one rocprofv3 --pmc pass: tools/kernel_counters.py <counter_collection.csv> [name filter]
Columns: launches, average duration (the pass's own timestamps), matrix-pipe busy share (SQ_VALU_MFMA_BUSY_CYCLES per SIMD
cycle), clock (GRBM_GUI_ACTIVE / 8 XCDs / duration), share of wave cycles spent waiting (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES)."""
import collections, csv, sys
rows = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if len(sys.argv) > 2 and sys.argv[2] not in k: continue
    rows[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if "Start_Timestamp" in r: dur[k][r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
mean = lambda v: sum(v) / max(len(v), 1)
print(f"{'kernel':70s} {'n':>4s} {'us':>8s} {'mfma busy':>9s} {'GHz':>5s} {'wait':>5s}")
for k, c in sorted(rows.items(), key=lambda kv: -sum(dur[kv[0]].values())):
    d = mean(list(dur[k].values())) if dur[k] else 0.0
    g = mean(c.get("GRBM_GUI_ACTIVE", [0])); clk = g / 8 / d if d else 0.0
    busy = mean(c.get("SQ_VALU_MFMA_BUSY_CYCLES", [0])) / (1024 * g / 8) if g else 0.0  # 1024 SIMDs x cycles
    wait = mean(c.get("SQ_WAIT_INST_ANY", [0])) / max(mean(c.get("SQ_WAVE_CYCLES", [1])), 1)
    print(f"{k[:70]:70s} {len(dur[k]):4d} {d/1e3:8.1f} {busy:9.3f} {clk:5.2f} {wait:5.2f}")
