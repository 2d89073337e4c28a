#!/bin/bash
# A/B of backward-kernel builds on one box: the graphed training iteration (tools/bench_train.py --graph).  Variants = tools/_build/librnf_<name>.so
# built by tools/ab_variants.py --build; edit the list below.
cp rotationnormflow_amd/librnf_hip.so /tmp/keep.so
# an interrupted run must not leave a variant build in the product's place (build() would treat it as fresh)
trap 'cp /tmp/keep.so rotationnormflow_amd/librnf_hip.so' EXIT
for r in 1 2; do
for v in ool inl norare; do
  cp tools/_build/librnf_$v.so rotationnormflow_amd/librnf_hip.so
  echo "$v C4 b128: $(python3 tools/bench_train.py --graph --config C4 --batch 128 --steps 200 2>/dev/null | python3 -c 'import json,sys; print(json.loads(sys.stdin.read())["ms_per_iteration"])')"
  echo "$v C2 b1024: $(python3 tools/bench_train.py --graph --steps 200 2>/dev/null | python3 -c 'import json,sys; print(json.loads(sys.stdin.read())["ms_per_iteration"])')"
done
done
