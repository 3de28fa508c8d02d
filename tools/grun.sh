#!/bin/bash
# gpurun wrapper for the build container: retries only when no box / slot was free (nothing ran, nothing charged).
#   tools/grun.sh TIMEOUT 'command'
T=$1; shift
for i in 1 2 3 4 5 6; do
  out=$(/usr/local/graft/bin/gpurun --timeout $T -- "$@" 2>&1); rc=$?
  echo "$out" | tail -60
  if echo "$out" | grep -q "status=transient"; then sleep 150; continue; fi
  exit $rc
done
exit 3
