#!/usr/bin/env bash
# Development aid (GPU box): durations of the extraction kernels by grid size (the two kinds of count-job launches apart),
# emit kernels one after the other (GRAFIMO_EXTRACT_SERIAL=1).
cd /tmp && export TMPDIR=/tmp
export GRAFIMO_EXTRACT_SERIAL=1
rocprofv3 --kernel-trace --output-format csv -d /tmp/jt -- python3 $GRAFT_REPO_ROOT/scripts/extract_bench.py > /dev/null 2>&1
f=$(ls -t /tmp/jt/*/*kernel_trace.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "graph_" in n:
        import re
        key = (re.search(r"graph_\w+", n).group(0), r.get("Grid_Size_X"))
        d[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items()):
    print("%-28s grid %-10s n=%3d  avg %8.1f us  min %8.1f" % (k[0], k[1], len(v), sum(v) / len(v), min(v)))
PY
