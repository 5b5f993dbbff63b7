#!/bin/bash
# gpurun with a bounded retry on "no slot free right now" (exit code 3: nothing charged).  usage: gpurun_retry.sh TIMEOUT 'command'
for i in 1 2 3 4 5 6 7 8; do
  /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"; rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 90
done
exit 3
