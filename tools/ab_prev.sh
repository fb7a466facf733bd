#!/bin/bash
# GPU box: A/B of the current library against tools/libartn_prev.so (the previous build) inside one session
O=gpurun_out/ab_prev.txt
: > $O
for rep in 1 2; do
echo "== new" >> $O; python3 bench.py --no-workloads --no-cpu-baseline --steps 5 --detail gpurun_out/detail_new.txt 2>/dev/null | python3 -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l['ms_per_step'], l['value'], l['config']['check'])" >> $O
echo "== prev" >> $O; ARTN_LIB=tools/libartn_prev.so python3 bench.py --no-workloads --no-cpu-baseline --steps 5 --detail gpurun_out/detail_prev.txt 2>/dev/null | python3 -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l['ms_per_step'], l['value'], l['config']['check'])" >> $O
done
cat $O
paste <(awk '{print $1,$3,$4,$11}' gpurun_out/detail_prev.txt) <(awk '{print $11}' gpurun_out/detail_new.txt) | awk '$4>1 || NR==1'
