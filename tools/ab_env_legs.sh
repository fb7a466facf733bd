#!/bin/bash
# GPU box: A/B of bench legs over the values of ONE environment switch of the library, inside one session
#   tools/ab_env_legs.sh ARTN_XG_TAIL "0 1 0 1" rand3,rand6 [tag]
V=$1; L=$2; W=${3:-rand3,rand6}; T=${4:-x}
O=gpurun_out/ab_env_legs_$T.txt
: > $O
for f in $L; do
  echo "== $V=$f" >> $O
  env $V=$f python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 --only-workloads $W 2>/dev/null | python3 -c "
import sys,json
for ln in sys.stdin.read().strip().splitlines():
    if not ln.startswith('{'): continue
    v=json.loads(ln)
    if 'leg' in v:
        if 'error' in v: print(v['leg'], v['error']); continue
        print('  leg', v['leg'], round(v['value'],1), 'TF', round(v['ms'],2), 'ms', v['check']['check'])
" >> $O
done
cat $O
