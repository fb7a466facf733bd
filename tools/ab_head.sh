#!/bin/bash
# A/B of one environment switch on the headline (n30 dense) inside one session: ARTN_AB_VAR, ARTN_AB_LIST
O=gpurun_out/ab_head.txt
: > $O
for f in ${ARTN_AB_LIST}; do
  echo "== ${ARTN_AB_VAR}=$f" >> $O
  env ${ARTN_AB_VAR}=$f python3 bench.py --no-cpu-baseline --no-workloads --steps 10 --warmup 2 2>/dev/null | python3 -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(l['value'],2), 'TF', round(l['ms_per_step'],2), 'ms', l.get('check'), 'kernel_ms', round(l['roofline'].get('kernel_ms',0),2))
" >> $O
done
cat $O
