#!/usr/bin/env bash
# Development aid (GPU box): the container's CPU quota and how often the streamed scan of 2e7 rows runs into it
# (cgroup v2 cpu.max / cpu.stat before and after scripts/scan_trace_threads.py).
cd "$(dirname "${BASH_SOURCE[0]}")/.."
echo "nproc $(nproc); cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
echo "-- cpu.stat before"; grep -E "nr_periods|nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat 2>/dev/null
GRAFIMO_PARSE_THREADS_EXACT=1 python scripts/scan_trace_threads.py 2>&1 | grep -E "=== threads|^total|STUCK|longest" | cut -c1-260
echo "-- cpu.stat after"; grep -E "nr_periods|nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat 2>/dev/null
echo "-- with the library's own thread choice (quota-aware caps)"
python scripts/ingest_probe.py 2>/dev/null | grep rows
