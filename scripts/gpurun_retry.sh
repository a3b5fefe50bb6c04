#!/bin/bash
# gpurun with retries while the pod's GPU slots are busy (exit code 3 = nothing charged).  usage: gpurun_retry.sh <timeout-seconds> '<command>' [logfile]
T=$1; CMD=$2; LOG=${3:-/dev/stdout}
for k in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$CMD" > "$LOG.tmp" 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then mv "$LOG.tmp" "$LOG" 2>/dev/null; exit $rc; fi
  sleep 45
done
mv "$LOG.tmp" "$LOG" 2>/dev/null
exit 3
