#!/bin/bash
# What bounds the attention backward at BASELINE configs[4] sizes ("ablation table first").
# Here (build container): tools/attn_bwd_ablation.sh build   -> tools/.ab/libtdx_fbabl<bits>.so, one per ablation
# On the GPU box:         tools/attn_bwd_ablation.sh run     -> the table (each variant through tools/attn_bench.py)
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/generative-turbulence_amd/csrc; AB=$R/tools/.ab
VARIANTS="0 1 2 4 8 16 3 7 15 31"
if [ "$1" = build ]; then
  mkdir -p $AB
  make -C $C -j8 > /dev/null || exit 1
  for v in $VARIANTS; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-slp-vectorize -DFB_ABL=$v \
      -c $C/tdx_attention_bwd_mfma.hip -o $AB/fbabl$v.o || exit 1
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $AB/libtdx_fbabl$v.so $AB/fbabl$v.o \
      $(ls $C/build/*.o | grep -v tdx_attention_bwd_mfma.o) || exit 1
  done
  rm -f $AB/fbabl*.o; ls -la $AB | grep fbabl
else
  echo "# attention backward ablations at N = 73 728, 4 heads x 32 (tools/attn_bench.py; bits: 1 no exp2, 2 no P o dP product, 4 no"
  echo "# fp32 -> 16-bit conversions, 8 no dP MFMAs, 16 no transposed LDS reads; 0 = the product kernel)"
  for dt in bf16 f16; do
    for v in $VARIANTS; do
      echo -n "$dt FB_ABL=$v  "; TDX_LIB=$AB/libtdx_fbabl$v.so python3 $R/tools/attn_bench.py --dtype $dt 2>/dev/null | grep bwd
    done
  done
fi
