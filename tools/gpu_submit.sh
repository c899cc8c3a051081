#!/bin/bash
# tools/gpu_submit.sh <timeout-seconds> '<command>' — gpurun, re-submitted only while the pool reports
# "no box / slot free" (exit 3: nothing ran, nothing charged).  Any other outcome is final.
T=$1; shift
for attempt in 1 2 3 4 5 6 7 8 9 10; do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  echo "[gpu_submit] no slot (attempt $attempt), waiting 150 s" >&2
  sleep 150
done
exit 3
