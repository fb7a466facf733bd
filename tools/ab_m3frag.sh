#!/bin/bash
O=gpurun_out/ab_m3frag.txt
: > $O
for f in 96 80; do
  echo "== ARTN_M3_FRAG=$f" >> $O
  ARTN_M3_FRAG=$f python3 bench.py --no-cpu-baseline --steps 2 --only-workloads n53m20,n30_sparse10000,n53m20b,n53,rand2 2>/dev/null | python3 -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k,v in l['workloads'].items():
    if 'error' in v: print(k, v['error']); continue
    print(k, round(v['value'],1), 'TF', round(v['ms'],2), 'ms', v['check']['check'], v['check'].get('vs_c128_truth',{}).get('hip_loose'))
" >> $O
done
cat $O
