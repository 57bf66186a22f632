#!/usr/bin/env bash
# Development aid (GPU box): graph_score_kernel against the number of regions (tiles): what part of its time does not scale
# with the work (start, tail, imbalance).   scripts/fused_size_sweep.sh <tag> [n_regions ...]
root="$GRAFT_REPO_ROOT"; tag="${1:-sweep}"; shift
out="$root/gpurun_out/$tag"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for n in ${@:-2500 5000 10000 20000 40000}; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/p_$n" -- python3 "$root/scripts/fused_prof.py" $n > "$out/log_$n.txt" 2>&1
  f=$(ls -t "$out"/p_$n/*/*kernel_stats.csv 2>/dev/null | head -1)
  python3 - "$f" "$n" <<'PY'
import csv, sys
rows = {r["Name"].split("namespace)::")[1].split("(")[0]: float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(sys.argv[1])) if "graph_" in r["Name"]}
print(f"regions {sys.argv[2]:>6s}: " + ", ".join(f"{k} {v:.1f}" for k, v in rows.items() if k.startswith(("graph_score_kernel", "graph_del_score", "graph_heavy"))))
PY
  rm -rf "$out/p_$n"
done 2>&1 | tee "$out/summary.txt"
