#!/usr/bin/env bash
# Development aid (GPU box): the streamed scan against the parse thread count WITHOUT the library's caps
# (GRAFIMO_PARSE_THREADS_EXACT=1), TSV directories on tmpfs (/dev/shm) and on the disk-backed page cache (/tmp).
cd "$(dirname "${BASH_SOURCE[0]}")/.."
export GRAFIMO_PARSE_THREADS_EXACT=1
for base in /dev/shm /tmp; do
  echo "== TSV directories under $base"
  GRAFIMO_BENCH_TSV_BASE=$base python scripts/ingest_probe.py 2>/dev/null | grep rows
done
