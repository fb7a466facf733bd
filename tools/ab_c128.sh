#!/bin/bash
# A/B of the complex128 state-streaming kernel (fused pairs on artn_k_bits128) against un-fused GEMM passes only
O=gpurun_out/ab_c128.txt
: > $O
for f in ${ARTN_AB_LIST:-1 0 2}; do
  echo "== ARTN_BITS128=$f" >> $O
  ARTN_BITS128=$f python3 bench.py --no-cpu-baseline --steps 2 --only-workloads n30_c128 2>/dev/null | python3 -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k,v in l['workloads'].items():
    if 'error' in v: print(k, v['error']); continue
    print(k, round(v['value'],2), 'TF', round(v['ms'],2), 'ms', v['check'], v.get('roofline'))
    for kk in v.get('kernels', [])[:12]: print('   ', kk)
" >> $O
done
cat $O
