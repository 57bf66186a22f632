#!/usr/bin/env bash
# Development aid (GPU box): the kernel sequence of the fused path's steady state -- the last launches of scripts/fused_prof.py
# from a rocprofv3 kernel trace, with start offsets and durations.   scripts/fused_trace.sh <tag> [lib]
root="$GRAFT_REPO_ROOT"; tag="$1"; out="$root/gpurun_out/$tag"; mkdir -p "$out"
[ -n "$2" ] && export GRAFIMO_HIP_LIB="$root/$2"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/p" -- python3 "$root/scripts/fused_prof.py" > "$out/log.txt" 2>&1
f=$(ls -t "$out"/p/*/*kernel_trace.csv | head -1)
cp "$(ls -t "$out"/p/*/*kernel_stats.csv | head -1)" "$out/kernel_stats.csv"
python3 - "$f" <<'PY' | tee "$out/sequence.txt"
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "graph_" in r["Kernel_Name"] or "Memset" in r["Kernel_Name"] or "fill" in r["Kernel_Name"].lower()]
t0 = None
for r in rows[-40:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if t0 is None: t0 = s
    name = r["Kernel_Name"].split("namespace)::")[-1].split("(")[0][:60]
    print(f"{(s - t0) / 1e3:10.1f} us  +{(e - s) / 1e3:7.1f} us  grid {r.get('Grid_Size', '?'):>8s}  {name}")
PY
grep fused_ms "$out/log.txt"
rm -rf "$out/p"
