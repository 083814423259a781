#!/bin/bash
# tools/gpu.sh <logname> <timeout_s> <command...>: gpurun with retries while no slot is free (rc 3)
L=$1; T=$2; shift 2
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
  timeout $((T + 1500)) gpurun --timeout $T -- "$@" > gpurun_out/$L.log 2>&1
  rc=$?
  if grep -q "status=transient" gpurun_out/$L.log; then sleep 45; continue; fi
  break
done
echo "done rc=$rc" >> gpurun_out/$L.log
