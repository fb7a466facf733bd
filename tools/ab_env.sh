#!/bin/bash
# generic A/B inside one session: ARTN_AB_VAR=<env name> ARTN_AB_LIST="a b a b" ARTN_AB_WORK=<bench legs>
O=gpurun_out/ab_env.txt
: > $O
for f in ${ARTN_AB_LIST}; do
  echo "== ${ARTN_AB_VAR}=$f" >> $O
  env ${ARTN_AB_VAR}=$f python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 --only-workloads ${ARTN_AB_WORK:-n53m20,n53,rand2} 2>/dev/null | python3 -c "
import sys,json
for ln in sys.stdin.read().strip().splitlines():
    if not ln.startswith('{\"leg\"'): continue
    v=json.loads(ln)
    if 'error' in v: print(v['leg'], v['error']); continue
    r=v.get('roofline') or {}
    print(v['leg'], round(v['value'],1), 'TF', round(v['ms'],2), 'ms', v['check']['check'], 'dominant', r.get('kernel'), round(r.get('kernel_ms',0),2), 'ms', 'fid', v['check'].get('slice0_fidelity_vs_reference'), 'loose', (v['check'].get('vs_c128_truth') or {}).get('hip_loose'))
" >> $O
done
cat $O
