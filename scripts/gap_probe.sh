#!/usr/bin/env bash
# Development aid (GPU box): kernel timeline of the bench loop - idle gaps between successive score kernels.
#   scripts/gap_probe.sh <tag> [bench args]
set -u
tag="$1"; shift
root="$GRAFT_REPO_ROOT"
out="$root/gpurun_out/gap_$tag"
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$out/t" -- python3 "$root/bench.py" --steps 50 --no-cpu-baseline --no-e2e "$@" > "$out/run.log" 2>&1
f=$(ls -t "$out"/t/*/*kernel_trace.csv | head -1)
python3 - "$f" <<'PY' | tee "$out/gaps.txt"
import csv, sys, statistics as st
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in rows))
sc = [k for k in ks if "score_quad" in k[2]]
gaps = [b[0] - a[1] for a, b in zip(sc, sc[1:]) if b[0] - a[1] < 200000]
dur = [k[1] - k[0] for k in sc]
print("score kernels", len(sc), "dur median", st.median(dur), "gap median", st.median(gaps), "min", min(gaps), "p90", sorted(gaps)[int(.9 * len(gaps))])
# one steady-state window: everything between two score kernels in the middle
mid = len(sc) // 2
t0 = sc[mid][0]
for k in ks:
    if sc[mid][0] <= k[0] <= sc[mid + 3][1]:
        print(f"{k[0]-t0:9d} {k[1]-t0:9d} {k[1]-k[0]:8d} q={k[3]:4s} {k[2].replace('(anonymous namespace)::','')[:60]}")
PY
rm -rf "$out/t"
