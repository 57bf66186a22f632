#!/usr/bin/env bash
# Development aid (GPU box): start / end of every kernel of one gfm_graph_emit (as shipped: two streams), relative to the
# first of them, from a rocprofv3 kernel trace of scripts/extract_bench.py.
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/jt2
rocprofv3 --kernel-trace --output-format csv -d /tmp/jt2 -- python3 $GRAFT_REPO_ROOT/scripts/extract_bench.py > /dev/null 2>&1
f=$(ls -t /tmp/jt2/*/*kernel_trace.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "graph_" in n:
        import re
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.search(r"graph_\w+", n).group(0), r.get("Grid_Size_X"), r.get("Queue_Id")))
rows.sort()
# emits: each starts with a graph_emit_del_kernel or graph_emit_kernel; take the emits 10..12
starts = [i for i, r in enumerate(rows) if r[2] in ("graph_emit_del_kernel", "graph_emit_kernel") and (i == 0 or rows[i - 1][2] not in ("graph_emit_del_kernel", "graph_emit_kernel"))]
for s in starts[10:13]:
    t0 = rows[s][0]
    print("-- one emit")
    for r in rows[s:s + 5]:
        print("  %-28s grid %-9s queue %-4s start %8.1f us  end %8.1f us" % (r[2], r[3], r[4], (r[0] - t0) / 1e3, (r[1] - t0) / 1e3))
PY
