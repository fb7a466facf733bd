#!/bin/bash
# GPU box: A/B of artn_k_alt against artn_k_bits inside one session
O=gpurun_out/ab_alt.txt
: > $O
for m in 2 1 0; do
echo "== ALT=$m" >> $O; ARTN_ALT=$m python3 tools/ablate.py >> $O 2>&1
done
for m in 2 1 0; do
echo "== bench ALT=$m" >> $O; ARTN_ALT=$m python3 bench.py --no-workloads --no-cpu-baseline --steps 5 --detail gpurun_out/detail_alt$m.txt >> $O 2>&1
done
grep -v "amdgpu.ids" $O | cut -c1-330
