#!/usr/bin/env bash
# Development aid (GPU box): what graph_score_kernel spends its time on -- the same profile with parts switched off
# (GRAFIMO_FUSED_LAB bits: 1 no phase 2, 2 no base scores, 4 no window classification, 8 no booking; results are wrong in
# those runs, only the kernel times count; every run under `timeout`: a switched-off part must never leave a wavefront
# without the data its loops end on -- a bit that skipped the staging of the next tile did, and hung the box for 25 minutes).   scripts/fused_lab.sh <tag>
# Needs a LAB BUILD (the product library has neither the switches nor the timers): scripts/lab_build.sh fusedlab -DGFM_LAB
root="$GRAFT_REPO_ROOT"; out="$root/gpurun_out/${1:-r05lab}"; mkdir -p "$out"
export GRAFIMO_HIP_LIB="$root/lab/libgfm_fusedlab.so"
[ -f "$GRAFIMO_HIP_LIB" ] || { echo "build lab/libgfm_fusedlab.so first (scripts/lab_build.sh fusedlab -DGFM_LAB)"; exit 1; }
cd /tmp && export TMPDIR=/tmp
for lab in ${FUSED_LAB_MODES:-0 1 4 8 3 7 15}; do
  export GRAFIMO_FUSED_LAB=$lab
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/p$lab" -- python3 "$root/scripts/fused_prof.py" > "$out/log$lab.txt" 2>&1
  f=$(ls -t "$out"/p$lab/*/*kernel_stats.csv 2>/dev/null | head -1)
  python3 - "$f" "$lab" <<'PY'
import csv, sys
rows = {r["Name"].split("namespace)::")[1].split("(")[0]: float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(sys.argv[1])) if "graph_" in r["Name"]}
print(f"lab={sys.argv[2]:>2s}: " + ", ".join(f"{k} {v:.1f} us" for k, v in rows.items() if k.startswith(("graph_score_kernel<1, false", "graph_del_score_kernel", "graph_hist_reduce_kernel"))))
PY
  rm -rf "$out/p$lab"
done
