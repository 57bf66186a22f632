#!/usr/bin/env bash
# Development aid (GPU box): the batched launches of BASELINE config 5 one shape at a time -- the plan per width, the
# bare-stream floor of each byte mix (scripts/micro/stream_mm), and rocprofv3's per-shape kernel times of the bench.
#   scripts/config5_lab.sh <tag>   (outputs under gpurun_out/c5_<tag>/)
set -u
tag="${1:-a}"
root="$GRAFT_REPO_ROOT"
out="$root/gpurun_out/c5_$tag"
mkdir -p "$out"
cd "$root"
python scripts/config5_plan_probe.py > "$out/plan.txt" 2>&1
[ -x scripts/micro/stream_mm ] && timeout 120 scripts/micro/stream_mm > "$out/stream_mm.txt" 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof5" -- python3 "$root/bench.py" --config 5 --steps ${STEPS:-10} --no-cpu-baseline --no-e2e --no-extras > "$out/prof5.log" 2>&1
f=$(ls -t "$out"/prof5/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp "$f" "$out/kernel_stats_config5.csv"
rm -rf "$out/prof5"
python3 - "$out/kernel_stats_config5.csv" > "$out/shapes.txt" <<'PY'
import csv, re, sys
tot = 0.0
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"score_quad_kernel<(\d+), (\d+)>", r["Name"])
    if not m:
        continue
    W, MM = int(m.group(1)), int(m.group(2))
    avg = float(r["AverageNs"]) / 1e3
    b = 1e8 * (W + 4 * MM)
    tot += avg
    print(f"W={W:2d} MM={MM} calls={r['Calls']:>4s} {avg:8.1f} us  {b / avg / 1e6:5.2f} TB/s  frac {b / avg / 1e6 / 8:.3f}")
print(f"sum of the shapes' averages {tot / 1e3:.3f} ms")
PY
tail -1 "$out/prof5.log" | cut -c1-400
cat "$out/plan.txt" "$out/stream_mm.txt" "$out/shapes.txt" | grep -v amdgpu.ids
