#!/bin/bash
# Evidence for profiles/: bench line, rocprofv3 kernel stats of the same command, PMC passes (each counter group in its own run, as
# MI355X_MICROARCH.md prescribes), phase stamps, all BASELINE configs.   usage (GPU box): bash tools/profile_round.sh <out dir under gpurun_out>
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/${1:-round}
mkdir -p $OUT
python3 bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 $REPO/bench.py --steps 10 --warmup 5 --no-cpu-baseline > $OUT/bench_under_rocprofv3.json 2> $OUT/stats.err
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc$i -o run -- python3 $REPO/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OUT/pmc$i.json 2> $OUT/pmc$i.err
done
cd $REPO
python3 tools/pmc_summary.py $OUT/pmc_summary.csv $OUT/pmc_summary.json flow_stack_kernel $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 $OUT/pmc4 $OUT/pmc5 > /dev/null
python3 tools/phase_stamps.py --preset C2 > $OUT/stamps_c2.txt 2>&1
python3 tools/phase_stamps.py --preset C4 > $OUT/stamps_c4.txt 2>&1
python3 tools/bench_configs.py > $OUT/configs.jsonl 2> $OUT/configs.err
python3 tools/parity_stats.py > $OUT/parity_stats.jsonl 2>&1
find $OUT -name "*.csv" -size +2M -delete
ls $OUT
