#!/usr/bin/env bash
# Round-end measurement pass on the GPU box: parity tests, the four bench lines, rocprofv3 summaries (configs 2-5 and the
# fused extraction -> scoring kernels), PMC passes (score kernel; FETCH / WRITE of the fused kernels), the ingest probe.
#   scripts/final_pass.sh <tag>      (outputs under gpurun_out/final_<tag>/)
set -u
tag="${1:-r06}"
root="$GRAFT_REPO_ROOT"
out="$root/gpurun_out/final_$tag"
mkdir -p "$out"
cd "$root"
python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" > "$out/pytest.txt"
python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.txt" 2>&1
for c in 2 3 4 5; do
  python bench.py --config $c 2> "$out/bench_config$c.err" | tail -1 > "$out/bench_config$c.json"
done
cd /tmp && export TMPDIR=/tmp
for c in 2 3 4 5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof$c" -- python3 "$root/bench.py" --config $c --steps 50 --no-cpu-baseline --no-e2e --no-extras > "$out/prof$c.log" 2>&1
  f=$(ls -t "$out"/prof$c/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp "$f" "$out/kernel_stats_config$c.csv"
  rm -rf "$out/prof$c"
done
# the fused extraction -> scoring kernels: time summary, then FETCH_SIZE / WRITE_SIZE in passes of their own
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/profx" -- python3 "$root/scripts/fused_prof.py" > "$out/profx.log" 2>&1
f=$(ls -t "$out"/profx/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp "$f" "$out/extract_kernel_stats.csv"
rm -rf "$out/profx"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$out/pmcx_$c" -- python3 "$root/scripts/fused_prof.py" > "$out/pmcx_$c.log" 2>&1
done
python3 - "$out" <<'PY'
import sys, glob, csv, collections, os
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob(out + "/pmcx_*/")):
    for f in sorted(glob.glob(d + "*/*counter_collection.csv"), key=os.path.getmtime)[-1:]:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-40:]
            if k.startswith("graph_"):
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/pmc_extract_kernels.txt", "w") as fh:
    fh.write("FETCH_SIZE / WRITE_SIZE (KiB per launch) of the fused extraction -> scoring kernels, scripts/fused_prof.py "
             "(10 000 regions x 200 bp, 6.04e6 rows scored per gfm_graph_score call); separate --pmc passes\n")
    for k, d in sorted(agg.items()):
        for c, v in sorted(d.items()):
            fh.write(f"{k:28s} {c:12s} launches={len(v):3d} mean={sum(v)/len(v):12.1f} KiB\n")
PY
rm -rf "$out"/pmcx_*/
python3 "$root/scripts/extract_bench.py" 2>/dev/null | tail -1 > "$out/extract.json"
bash "$root/scripts/lab_pmc.sh" default "$tag" > "$out/pmc.log" 2>&1
cp "$root/gpurun_out/pmc_$tag/summary.txt" "$out/pmc_summary.txt" 2>/dev/null
rm -rf "$root/gpurun_out/pmc_$tag"/*/
cd "$root"
python scripts/ingest_probe.py > "$out/ingest_probe.txt" 2>&1
# SQ counters of the fused kernels (the bench's extract.roofline prices graph_score_kernel against them)
bash "$root/scripts/pmc_fused_sq.sh" > /dev/null 2>&1
cp "$root/gpurun_out/pmc_fused_sq/summary.txt" "$out/pmc_fused_sq.txt" 2>/dev/null
# per-phase timers exist in lab builds only (scripts/lab_build.sh fusedlab -DGFM_LAB)
# (GRAFIMO_FUSED_TIMERS=2: one store per tile, no atomics -- mode 1's per-phase sums go through atomics on a few words and
# stretch the kernel tenfold)
[ -f "$root/lab/libgfm_fusedlab.so" ] && GRAFIMO_HIP_LIB="$root/lab/libgfm_fusedlab.so" GRAFIMO_FUSED_TIMERS=2 python scripts/fused_prof.py 2>&1 | grep "\[fused\]" | tail -18 > "$out/fused_timers.txt"
# the five fuzz drivers on fresh seeds (bounded: FUZZ_S seconds each), the round's last code
{
  s0=$(( $(date +%s) % 100000 ))
  for f in score_fuzz scan_fuzz results_fuzz extract_fuzz tables_fuzz; do
    echo "== scripts/$f.py ${FUZZ_S:-100} $s0"
    timeout $(( ${FUZZ_S:-100} + 200 )) python scripts/$f.py ${FUZZ_S:-100} $s0 2>&1 | grep -v -E "amdgpu.ids|^NOTE|^WARNING" | tail -4
  done
} > "$out/fuzz_final.txt" 2>&1
