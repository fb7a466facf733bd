#!/bin/bash
# GPU box: per-pair timings of the n30 steps under knobs and diagnostic builds (A/B inside ONE session).
set -u
O=gpurun_out/diag_r03.txt
: > $O
echo "== product" >> $O; python3 tools/ablate.py >> $O 2>&1
echo "== ARTN_WG_PER_CU=1" >> $O; ARTN_WG_PER_CU=1 python3 tools/ablate.py >> $O 2>&1
echo "== ARTN_STAGE_PRIO=0" >> $O; ARTN_STAGE_PRIO=0 python3 tools/ablate.py >> $O 2>&1
echo "== ARTN_BITS_3M=0" >> $O; ARTN_BITS_3M=0 python3 tools/ablate.py >> $O 2>&1
for l in nomfma nomem nomem_nomfma; do
  if [ -f tools/libartn_hip_$l.so ]; then echo "== $l" >> $O; ARTN_LIB=tools/libartn_hip_$l.so python3 tools/ablate.py >> $O 2>&1; fi
done
if [ -f tools/libartn_hip_phases.so ]; then echo "== phases" >> $O; ARTN_LIB=tools/libartn_hip_phases.so python3 tools/phases.py >> $O 2>&1; fi
if [ -f tools/libartn_hip_phases.so ]; then echo "== phases WG_PER_CU=1" >> $O; ARTN_WG_PER_CU=1 ARTN_LIB=tools/libartn_hip_phases.so python3 tools/phases.py >> $O 2>&1; fi
grep -v "Warning\|warn" $O
