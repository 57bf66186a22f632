#!/usr/bin/env bash
# Development aid (GPU box): HIP API time summary (rocprofv3 --hip-trace --stats) of scripts/extract_bench.py.
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ht
rocprofv3 --hip-trace --stats --output-format csv -d /tmp/ht -- python3 $GRAFT_REPO_ROOT/scripts/extract_bench.py > /dev/null 2>&1
f=$(ls -t /tmp/ht/*/*hip_api_stats.csv 2>/dev/null | head -1)
[ -z "$f" ] && ls /tmp/ht/*/ && exit
python3 - "$f" <<'PY'
import csv, sys
for i, r in enumerate(csv.reader(open(sys.argv[1]))):
    if i == 0 or i > 16: continue
    print("%-36s calls %6s total %10.1f us avg %8.1f us" % (r[0][:36], r[1], float(r[2]) / 1e3, float(r[3]) / 1e3))
PY
