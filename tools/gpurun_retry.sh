#!/bin/bash
# gpurun with retries while no GPU slot is free: tools/gpurun_retry.sh <log> <timeout> <command...>
log=$1; shift; to=$1; shift
for attempt in 1 2 3 4 5 6 7 8 9 10; do
  /usr/local/graft/bin/gpurun --timeout $to -- "$@" > $log 2>&1
  rc=$?
  if grep -q "status=transient" $log; then sleep 60; continue; fi
  exit $rc
done
exit 3
