#!/usr/bin/env bash
# Development aid (GPU box): the step's gap over the score kernel and the tail's length under variants of the resident
# pipeline -- slots (host-paced reuse), CUs left to the tail stream -- several runs each, on one box.
#   scripts/step_gap_probe.sh <tag>
root="$GRAFT_REPO_ROOT"; tag="${1:-step_gap}"; out="$root/gpurun_out/$tag.txt"
cd "$root"
run() {   # label, env, args...
  label="$1"; envs="$2"; shift 2
  for rep in 1 2 3; do
    env $envs python3 bench.py --no-cpu-baseline --no-e2e --no-extras "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; t=d.get('tail_ms') or {}
print('%-34s value %.4g  step %.2f us  kernel %.2f us  gap %.2f us  tail avg %.1f max %.1f us  bursts %s' % ('$label', d['value'], 1e3*d['ms_per_step'], 1e3*r['kernel_ms_avg'], 1e3*(d['ms_per_step']-r['kernel_ms_avg']), 1e3*t.get('avg',0), 1e3*t.get('max',0), d['config']['burst_ms']))"
  done
}
{
run "slots 3, ring 3 (round 5)" "" --slots 3 --score-buffers 0
run "slots 4, ring 2" "" --slots 4 --score-buffers 2
run "slots 6, ring 2" "" --slots 6 --score-buffers 2
run "slots 8, ring 2" "" --slots 8 --score-buffers 2
run "slots 8, ring 3" "" --slots 8 --score-buffers 3
} | tee "$out"
# ... and inside the FULL bench line (CPU baseline, TSV directories, extras: what the driver runs)
for cfg in "--slots 4 --score-buffers 2" "--slots 8 --score-buffers 2"; do
  python3 bench.py --no-config45 $cfg 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; t=d.get('tail_ms') or {}
print('FULL bench %-28s value %.4g  step %.2f us  kernel %.2f us  gap %.2f us  tail avg %.1f max %.1f us' % ('$cfg', d['value'], 1e3*d['ms_per_step'], 1e3*r['kernel_ms_avg'], 1e3*(d['ms_per_step']-r['kernel_ms_avg']), 1e3*t.get('avg',0), 1e3*t.get('max',0)))" | tee -a "$out"
done
