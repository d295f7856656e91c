#!/bin/bash
# tools/scratch/gpurun_retry.sh <timeout-seconds> '<command>': gpurun, retried every two minutes while the pod has no free GPU slot (exit code 3)
T=$1; shift
for attempt in $(seq 1 20); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  echo "[retry] attempt $attempt: no slot, sleeping 120 s"
  sleep 120
done
exit 3
