#!/bin/bash
# A/B of artn_k_gemm_deep (operand loads two chunks ahead) on the GEMM-heavy workloads
O=gpurun_out/ab_deep.txt
: > $O
for f in ${ARTN_AB_LIST:-1 0 1 0}; do
  echo "== ARTN_GEMM_DEEP=$f" >> $O
  ARTN_GEMM_DEEP=$f python3 bench.py --no-cpu-baseline --steps 2 --only-workloads ${ARTN_AB_WORK:-n53m20,n53,rand2} 2>/dev/null | python3 -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k,v in l['workloads'].items():
    if 'error' in v: print(k, v['error']); continue
    print(k, round(v['value'],1), 'TF', round(v['ms'],2), 'ms', v['check']['check'], v['check'].get('vs_c128_truth',{}).get('hip_loose'))
" >> $O
done
cat $O
