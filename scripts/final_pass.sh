#!/usr/bin/env bash
# Round-end measurement pass on the GPU box: parity tests, the four bench lines, rocprofv3 summaries, PMC passes.
#   scripts/final_pass.sh <tag>      (outputs under gpurun_out/final_<tag>/)
set -u
tag="${1:-r02}"
root="$GRAFT_REPO_ROOT"
out="$root/gpurun_out/final_$tag"
mkdir -p "$out"
cd "$root"
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" > "$out/pytest.txt"
for c in 2 3 4 5; do
  python bench.py --config $c 2> "$out/bench_config$c.err" | tail -1 > "$out/bench_config$c.json"
done
cd /tmp && export TMPDIR=/tmp
for c in 2 3 5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof$c" -- python3 "$root/bench.py" --config $c --steps 50 --no-cpu-baseline --no-e2e > "$out/prof$c.log" 2>&1
  f=$(ls -t "$out"/prof$c/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp "$f" "$out/kernel_stats_config$c.csv"
  rm -rf "$out/prof$c"
done
bash "$root/scripts/lab_pmc.sh" default "$tag" > "$out/pmc.log" 2>&1
cp "$root/gpurun_out/pmc_$tag/summary.txt" "$out/pmc_summary.txt" 2>/dev/null
rm -rf "$root/gpurun_out/pmc_$tag"/*/
