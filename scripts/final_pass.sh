#!/usr/bin/env bash
# Round-end measurement pass on the GPU box: parity tests, the four bench lines, rocprofv3 summaries, PMC passes,
# the extraction kernels' summary, the ingest probe.
#   scripts/final_pass.sh <tag>      (outputs under gpurun_out/final_<tag>/)
set -u
tag="${1:-r03}"
root="$GRAFT_REPO_ROOT"
out="$root/gpurun_out/final_$tag"
mkdir -p "$out"
cd "$root"
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" > "$out/pytest.txt"
python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.txt" 2>&1
for c in 2 3 4 5; do
  python bench.py --config $c 2> "$out/bench_config$c.err" | tail -1 > "$out/bench_config$c.json"
done
cd /tmp && export TMPDIR=/tmp
for c in 2 3 5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof$c" -- python3 "$root/bench.py" --config $c --steps 50 --no-cpu-baseline --no-e2e --no-extras > "$out/prof$c.log" 2>&1
  f=$(ls -t "$out"/prof$c/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp "$f" "$out/kernel_stats_config$c.csv"
  rm -rf "$out/prof$c"
done
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/profx" -- python3 "$root/scripts/extract_bench.py" > "$out/extract.json" 2> "$out/profx.log"
f=$(ls -t "$out"/profx/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp "$f" "$out/extract_kernel_stats.csv"
rm -rf "$out/profx"
bash "$root/scripts/lab_pmc.sh" default "$tag" > "$out/pmc.log" 2>&1
cp "$root/gpurun_out/pmc_$tag/summary.txt" "$out/pmc_summary.txt" 2>/dev/null
rm -rf "$root/gpurun_out/pmc_$tag"/*/
cd "$root"
python scripts/ingest_probe.py > "$out/ingest_probe.txt" 2>&1
