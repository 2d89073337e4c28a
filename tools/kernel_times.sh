#!/bin/bash
# per-kernel average durations (rocprofv3 --kernel-trace --stats) of library variants:  bash tools/kernel_times.sh <preset> <variant> [variant ...]
REPO=$(pwd); PRESET=$1; shift
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  rm -rf /tmp/kt_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$v -o run -- python3 $REPO/tools/run_variant.py $v $PRESET > /tmp/kt_$v.out 2> /tmp/kt_$v.err
  tail -1 /tmp/kt_$v.out
  f=$(find /tmp/kt_$v -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$v" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if float(r["Percentage"]) > 1.0:
        print(f'  {sys.argv[2]:12s} {r["Name"][:78]:78s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e6:8.4f} ms  {float(r["Percentage"]):5.1f} %')
PY
done
