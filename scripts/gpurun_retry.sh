#!/bin/bash
# gpurun with a wait-and-retry on exit code 3 ONLY ("no box or slot free right now, nothing charged"); any other outcome ends.
# usage: scripts/gpurun_retry.sh <log file> <gpurun args...>
log="$1"; shift
for attempt in $(seq 1 40); do
    /usr/local/graft/bin/gpurun "$@" > "$log" 2>&1
    rc=$?
    if [ "$rc" != "3" ]; then
        echo "gpurun rc=$rc after $attempt attempt(s)" >> "$log"
        exit $rc
    fi
    sleep 90
done
echo "gpurun: no slot after 40 attempts" >> "$log"
exit 3
