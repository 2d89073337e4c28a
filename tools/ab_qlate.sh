#!/bin/bash
# round 6: same-box A/B -- q = sp (1 - |u|^2) of the inverse tile loop computed by the root finder's start loop, packed (flow_kernels.h RNF_INV_Q_LATE)
mkdir -p gpurun_out/r6
python3 tools/ab_variants.py --build ql0="-DRNF_INV_Q_LATE=0" > /dev/null 2>&1
: > gpurun_out/r6/ab_qlate.jsonl
for p in C5u C5 C2; do
  python3 tools/ab_variants.py --run ql0 cur --preset $p --direction inverse --rounds 7 2>/dev/null | tee -a gpurun_out/r6/ab_qlate.jsonl
done
python3 tools/time_trained_inverse.py ql0 cur 2>/dev/null | tee -a gpurun_out/r6/ab_qlate.jsonl
