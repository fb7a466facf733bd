#!/bin/bash
# One GPU-box session of several bounded pieces (every piece under its own `timeout`: a hung piece must not eat the call).
# usage (through gpurun): tools/gpu_session.sh <name> ; pieces are read from tools/gpu_session_<name>.txt (profile: the four profiling scripts of a round; example: tests, an A/B, the independent truth), one per line:
#   <seconds> <log name> <command ...>
set -u
N=${1:-a}
O=gpurun_out/s_$N
mkdir -p $O
export TMPDIR=/tmp
while IFS= read -r line; do
  [ -z "$line" ] && continue
  case "$line" in \#*) continue;; esac
  secs=${line%% *}; rest=${line#* }; log=${rest%% *}; cmd=${rest#* }
  echo "== [$secs s] $cmd" > $O/$log.log
  timeout $secs bash -c "$cmd" >> $O/$log.log 2>&1
  echo "== rc=$?" >> $O/$log.log
  tail -c 1500 $O/$log.log
done < tools/gpu_session_$N.txt
