#!/bin/bash
# gpurun with a retry when the pod has no free GPU slot (exit code 3: nothing ran, nothing was charged).
#   usage: scripts/gpurun_retry.sh <timeout seconds> '<command>'
T=$1; shift
for i in $(seq 1 30); do
	/usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
	rc=$?
	if [ $rc -ne 3 ]; then exit $rc; fi
	sleep 45
done
exit 3
