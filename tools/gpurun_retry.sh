#!/bin/bash
# usage: tools/gpurun_retry.sh <timeout-seconds> <command>   — gpurun, retried while the pod's GPU slots are busy (exit code 3)
T=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@" > /tmp/gpurun_last.txt 2>&1
  rc=$?
  if grep -q "status=transient" /tmp/gpurun_last.txt; then sleep 45; continue; fi
  break
done
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" /tmp/gpurun_last.txt
exit $rc
