#!/bin/bash
# A/B of the HEADLINE (n30 m14) plus optional legs inside one session over env variants of one library:
#   ARTN_LIB=tools/libartn_dev.so ARTN_AB_VAR=ARTN_STAGE_PRIO ARTN_AB_LIST="3 4 3 4" [ARTN_AB_WORK=n53,rand2] tools/ab_head_env.sh [tag]
T=${1:-x}
O=gpurun_out/ab_head_env_$T.txt
: > $O
for f in ${ARTN_AB_LIST}; do
  echo "== ${ARTN_AB_VAR}=$f" >> $O
  if [ -n "${ARTN_AB_WORK:-}" ]; then W="--only-workloads ${ARTN_AB_WORK}"; else W="--no-workloads"; fi
  env ${ARTN_AB_VAR}=$f python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 $W --detail gpurun_out/detail_${T}_$f.txt 2>/dev/null | python3 -c "
import sys,json
for ln in sys.stdin.read().strip().splitlines():
    if not ln.startswith('{'): continue
    v=json.loads(ln)
    if 'leg' in v:
        if 'error' in v: print(v['leg'], v['error']); continue
        print('  leg', v['leg'], round(v['value'],1), 'TF', round(v['ms'],2), 'ms', v['check']['check'])
    elif 'ms_per_step' in v:
        print('headline', v['ms_per_step'], 'ms', v['value'], 'TF', v['config']['check'], 'loose', v['config'].get('err_loose'))
" >> $O
done
cat $O
