#!/bin/bash
# round 6: same-box A/B of the centre-evaluated confirming passes of the inverse root finder (flow_kernels.h RNF_RF_CENTRE)
mkdir -p gpurun_out/r6
python3 tools/ab_variants.py --build c0="-DRNF_RF_CENTRE=0" f3="-DRNF_RF_COND_FIRST4=0" > /dev/null 2>&1
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_trained.py tests/test_gpu_scale_properties.py -q -m gpu -x -k "inv or inverse or pose or sample" 2>&1 | grep -E "passed|failed|^FAILED|^ERROR" | tee gpurun_out/r6/ab_centre_tests.txt
: > gpurun_out/r6/ab_centre.jsonl
for p in C5u C5 C2; do
  python3 tools/ab_variants.py --run c0 cur f3 --preset $p --direction inverse --rounds 5 2>/dev/null | tee -a gpurun_out/r6/ab_centre.jsonl
done
python3 tools/time_trained_inverse.py c0 cur f3 2>/dev/null | tee -a gpurun_out/r6/ab_centre.jsonl
python3 bench.py --config C5q --steps 10 --warmup 5 --no-cpu-baseline --no-secondary --no-pmc --full-out /dev/null 2>/dev/null | tail -n 1 | python3 -c "import json,sys; c=json.loads(sys.stdin.read()); print(json.dumps({'variant':'cur','config':'C5q','ms_per_step':c['ms_per_step']}))" | tee -a gpurun_out/r6/ab_centre.jsonl
