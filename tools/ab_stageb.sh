#!/bin/bash
# round 6: same-box A/B -- stage B of the inverse tile loop (segment parameters) two segments per packed instruction (flow_kernels.h RNF_INV_STAGEB_PK)
mkdir -p gpurun_out/r6
python3 tools/ab_variants.py --build pk0="-DRNF_INV_STAGEB_PK=0" > /dev/null 2>&1
: > gpurun_out/r6/ab_stageb.jsonl
for p in C5u C5 C2; do
  python3 tools/ab_variants.py --run pk0 cur --preset $p --direction inverse --rounds 7 2>/dev/null | tee -a gpurun_out/r6/ab_stageb.jsonl
done
python3 tools/time_trained_inverse.py pk0 cur 2>/dev/null | tee -a gpurun_out/r6/ab_stageb.jsonl
