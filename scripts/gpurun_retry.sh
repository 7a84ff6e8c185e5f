#!/bin/bash
# gpurun with retries while the pod's GPU slots are busy (exit code 3 / "transient"): scripts/gpurun_retry.sh TIMEOUT 'command'
T=$1; shift
for try in $(seq 1 40); do
  out=$(gpurun --timeout $T -- "$@" 2>&1); rc=$?
  if echo "$out" | grep -q "status=transient\|status=refused.*already running"; then sleep 45; continue; fi
  echo "$out"; exit $rc
done
echo "gave up: GPU slots busy"; exit 3
