#!/bin/bash
# probe the box's C2 time; only a box whose clock is comparable with round 5's evidence box runs the full profile
mkdir -p gpurun_out/$1
ms=$(python3 bench.py --config C2 --steps 20 --warmup 10 --no-cpu-baseline --no-secondary --no-pmc --full-out /dev/null 2>/dev/null | tail -n 1 | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
echo "probe C2 ms=$ms" | tee gpurun_out/$1/probe.txt
if python3 -c "import sys; sys.exit(0 if float('$ms') < 4.37 else 1)"; then
  bash tools/profile_round.sh $1 > /dev/null 2>&1
  python3 -c "
import json; c=json.load(open('gpurun_out/$1/bench_compact.json')); print(c['ms_per_step'], c['value_strict'], {k:v.get('ms_per_step', v.get('ms_per_iteration')) for k,v in c['configs'].items()}, c['device_state'])"
else
  echo "slow box: skipped"
fi
