#!/bin/bash
# GPU box: A/B of the current library against another build of it ($1, e.g. tools/libartn_late.so) inside one session
L=${1:-tools/libartn_prev.so}
O=gpurun_out/ab_lib.txt
: > $O
for rep in 1 2; do
echo "== default" >> $O; python3 bench.py --no-workloads --no-cpu-baseline --steps 8 --detail gpurun_out/detail_new.txt 2>/dev/null | python3 -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l['ms_per_step'], l['value'], l['config']['check'])" >> $O
echo "== $L" >> $O; ARTN_LIB=$L python3 bench.py --no-workloads --no-cpu-baseline --steps 8 --detail gpurun_out/detail_prev.txt 2>/dev/null | python3 -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l['ms_per_step'], l['value'], l['config']['check'])" >> $O
done
cat $O
paste <(awk '{print $1,$3,$4,$11}' gpurun_out/detail_new.txt) <(awk '{print $11}' gpurun_out/detail_prev.txt) | awk '$4>1 || NR==1'
