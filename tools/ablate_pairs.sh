#!/bin/bash
# The 13 fused pairs of the n30 scheme on the product library and on the timing-only ablation builds (`make ablate`: no global
# memory traffic / no MFMAs / neither; wrong results by construction), one session: what a tile costs besides its MFMAs and bytes.
# usage (GPU box): tools/ablate_pairs.sh
for lib in artensor_amd/libartn_hip.so tools/libartn_hip_nomem.so tools/libartn_hip_nomfma.so tools/libartn_hip_nomem_nomfma.so; do
  echo "== $lib"
  ONLY_SCHEME=1 ARTN_LIB=$lib python3 tools/layout_probe.py 2>&1 | grep "^pair" | awk '{print $2, $3, $4, $7, $8, $9}' | tr '\n' ';'
  echo
done
