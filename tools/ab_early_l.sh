#!/bin/bash
# round 6: same-box A/B -- the next layer's fc_last image requested IN FRONT of the inverse root finder (flow_kernels.h RNF_INV_EARLY_L)
mkdir -p gpurun_out/r6
python3 tools/ab_variants.py --build el0="-DRNF_INV_EARLY_L=0" el1="-DRNF_INV_EARLY_L=1" > /dev/null 2>&1
: > gpurun_out/r6/ab_early_l.jsonl
for p in C5u C5 C2; do
  python3 tools/ab_variants.py --run el0 el1 --preset $p --direction inverse --rounds 7 2>/dev/null | tee -a gpurun_out/r6/ab_early_l.jsonl
done
python3 tools/time_trained_inverse.py el0 el1 2>/dev/null | tee -a gpurun_out/r6/ab_early_l.jsonl
