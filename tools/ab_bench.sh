#!/bin/bash
# tools/ab_bench.sh "<ENV=..>" [rounds] [extra bench.py flags]: same-box A/B of the training step: default build against the same build
# with the given tuning variables (SEGCLIP_TUNING=1 is added), alternating runs; prints pairs/s and ms per step of each run.
envb="$1"; rounds="${2:-2}"; shift; shift
line() { python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1]); print(sys.argv[1], d['value'], d['ms_per_step'])" "$1"; }
for i in $(seq 1 $rounds); do
  python bench.py --no-roofline --no-cpu-baseline --no-parity-leg --no-traffic "$@" 2>/dev/null | line "default      "
  env SEGCLIP_TUNING=1 $envb python bench.py --no-roofline --no-cpu-baseline --no-parity-leg --no-traffic "$@" 2>/dev/null | line "$envb"
done
