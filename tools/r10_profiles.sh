#!/bin/bash
# Round-3 evidence (GPU box: bash tools/r10_profiles.sh): rocprofv3 stats + PMC passes of the bench command in the bf16 and
# f32s modes (tools/collect_profiles.sh -> gpurun_out/prof_r10*; the weight gradients on the launching stream there, so that
# every kernel's duration and counters are its own), the config-5 attention profile with fp16 operands, the per-layer conv
# table, and two bench.py lines (default flags; --steps 20 --warmup 5 without the CPU / parity / PyTorch legs).
# Summaries: python tools/summarize_profiles.py r10bf16 (and r10f32s) -> profiles/.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r10p; mkdir -p $O
cd $R
bash tools/collect_profiles.sh r10bf16 > $O/collect_bf16.log 2>&1
TDX_BENCH_ARGS="--dtype f32s" bash tools/collect_profiles.sh r10f32s > $O/collect_f32s.log 2>&1
python tools/conv_bench.py > $O/r10_conv3_bf16_table.txt 2>&1
for dt in f16 bf16; do python tools/attn_bench.py --dtype $dt >> $O/r10_attention_config5.txt 2>&1; done
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_attn -- python3 $R/tools/attn_bench.py --dtype f16 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_attn -- python3 $R/tools/attn_bench.py --dtype f16 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_attn -- python3 $R/tools/attn_bench.py --dtype f16 > /dev/null 2>&1
python3 - <<PY >> $O/r10_attention_config5.txt
import csv, glob, collections
def avg(pattern):
    f = sorted(glob.glob(pattern, recursive=True))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[-1])) if f else []:
        agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg
st = sorted(glob.glob("$O/prof_attn/**/*kernel_stats.csv", recursive=True))
dur = {r["Name"][:70]: float(r["AverageNs"]) for r in csv.DictReader(open(st[-1]))} if st else {}
pm, fe = avg("$O/pmc_attn/**/*counter_collection.csv"), avg("$O/fetch_attn/**/*counter_collection.csv")
print("\n# rocprofv3 --pmc (fp16 operands, N = 73728, B = 1), averages per launch of the attention kernels")
for k, c in pm.items():
    if "attn" not in k: continue
    m = lambda n: sum(c.get(n, [0])) / max(len(c.get(n, [0])), 1)
    d = dur.get(k, 0.0)
    fetch = sum(fe.get(k, {}).get("FETCH_SIZE", [0])) / max(len(fe.get(k, {}).get("FETCH_SIZE", [0])), 1)
    clk = m("GRBM_GUI_ACTIVE") / 8 / d if d else 0
    busy = m("SQ_VALU_MFMA_BUSY_CYCLES") / (m("GRBM_GUI_ACTIVE") / 8 * 1024) if m("GRBM_GUI_ACTIVE") else 0  # 256 CUs x 4 SIMDs
    print(f"{k}: {d/1e6:.3f} ms, MFMA insts {m('SQ_INSTS_MFMA'):.3g}, VALU insts {m('SQ_INSTS_VALU'):.3g}, MFMA-busy cycles {m('SQ_VALU_MFMA_BUSY_CYCLES'):.3g} "
          f"= {100*busy:.1f} % of the SIMD-cycles, clock {clk:.2f} GHz, HBM read {2*fetch*1024/1e6:.1f} MB = {2*fetch*1024/max(d,1):.3f} GB/s ({100*2*fetch*1024/max(d,1)/8000:.3f} % of 8 TB/s)")
PY
cd $R
python bench.py > $O/r10_bench.json 2> $O/r10_bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity-modes --no-torch-baseline > $O/r10_bench_steps20_fullT.json 2> $O/r10_bench20.err
ls $O
