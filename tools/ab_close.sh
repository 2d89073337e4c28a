#!/bin/bash
# round 6: same-box A/B of the closing f' evaluation's denominator  2 (1 - a) + |u|^2 - 1  (flow_kernels.h RNF_RF_E1FOLD_CLOSE)
mkdir -p gpurun_out/r6
python3 tools/ab_variants.py --build cl0="-DRNF_RF_E1FOLD_CLOSE=0" > /dev/null 2>&1
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_trained.py tests/test_gpu_scale_properties.py -q -m gpu -x -k "inv or inverse or pose or sample or rootfinder" 2>&1 | grep -E "passed|failed|^FAILED|^ERROR" | tee gpurun_out/r6/ab_close_tests.txt
: > gpurun_out/r6/ab_close.jsonl
for p in C5u C5 C2; do
  python3 tools/ab_variants.py --run cl0 cur --preset $p --direction inverse --rounds 7 2>/dev/null | tee -a gpurun_out/r6/ab_close.jsonl
done
python3 tools/time_trained_inverse.py cl0 cur 2>/dev/null | tee -a gpurun_out/r6/ab_close.jsonl
