#!/bin/bash
# generic A/B inside one session: ARTN_AB_VAR=<env name> ARTN_AB_LIST="a b a b" ARTN_AB_WORK=<bench legs>
O=gpurun_out/ab_env.txt
: > $O
for f in ${ARTN_AB_LIST}; do
  echo "== ${ARTN_AB_VAR}=$f" >> $O
  env ${ARTN_AB_VAR}=$f python3 bench.py --no-cpu-baseline --steps 2 --only-workloads ${ARTN_AB_WORK:-n53m20,n53,rand2} 2>/dev/null | python3 -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k,v in l['workloads'].items():
    if 'error' in v: print(k, v['error']); continue
    print(k, round(v['value'],1), 'TF', round(v['ms'],2), 'ms', v['check']['check'], v['check'].get('vs_c128_truth',{}).get('hip_loose'))
" >> $O
done
cat $O
