#!/bin/bash
# Quick profile of one bench.py workload on the GPU box: rocprofv3 kernel stats + PMC passes (each counter group in its own run, no trace
# domains beside --kernel-trace, as MI355X_MICROARCH.md prescribes) + in-kernel phase stamps.
#   usage: bash tools/pmc_quick.sh <out dir under gpurun_out> [config=C2] [kernel substring=flow_stack_kernel]
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/${1:-pmc}
CFG=${2:-C2}
KERN=${3:-flow_stack_kernel}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 $REPO/bench.py --config $CFG --steps 10 --warmup 5 --no-cpu-baseline --no-secondary --no-pmc --full-out $OUT/bench_under_rocprofv3.json > /dev/null 2> $OUT/stats.err
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_COEXEC_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc$i -o run -- python3 $REPO/bench.py --config $CFG --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --no-pmc --full-out $OUT/pmc$i.json > /dev/null 2> $OUT/pmc$i.err
done
cd $REPO
python3 tools/pmc_summary.py $OUT/pmc_summary.csv $OUT/pmc_summary.json $KERN $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 $OUT/pmc4 $OUT/pmc5 $OUT/pmc6 > /dev/null
python3 tools/phase_stamps.py --preset $CFG > $OUT/stamps.txt 2>&1
cp $OUT/stats/*/run_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null || find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT -name "*.csv" -size +1M -delete
find $OUT -name "*.db" -delete
cat $OUT/pmc_summary.csv; cat $OUT/stamps.txt
