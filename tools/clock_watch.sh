#!/bin/bash
# Development aid: sample GPU clock and power while a bench runs.
#   tools/clock_watch.sh <seconds> -- <command...>
secs=$1; shift; shift
"$@" > gpurun_out/clock_watch_cmd.log 2>&1 &
pid=$!
for i in $(seq 1 $((secs * 2))); do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)|Socket Power" | tr '\n' ' '
    echo
    sleep 0.5
    kill -0 $pid 2>/dev/null || break
done
wait $pid
tail -c 600 gpurun_out/clock_watch_cmd.log
