#!/bin/bash
# GPU box: A/B of the current library against another build ($1) on bench legs ($2, comma-separated) inside one session
L=${1:-tools/libartn_prev.so}; W=${2:-rand3,rand6}
O=gpurun_out/ab_lib_legs.txt
: > $O
for rep in 1 2; do
for lib in default $L; do
  echo "== $lib" >> $O
  if [ $lib = default ]; then E=""; else E="ARTN_LIB=$lib"; fi
  env $E python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 --only-workloads $W 2>/dev/null | python3 -c "
import sys,json
for ln in sys.stdin.read().strip().splitlines():
    if not ln.startswith('{'): continue
    v=json.loads(ln)
    if 'leg' in v:
        if 'error' in v: print(v['leg'], v['error']); continue
        print('  leg', v['leg'], round(v['value'],1), 'TF', round(v['ms'],2), 'ms', v['check']['check'])
" >> $O
done
done
cat $O
