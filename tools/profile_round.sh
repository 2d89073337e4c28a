#!/bin/bash
# Evidence for profiles/<round>/: the driver's bench line, rocprofv3 kernel stats of the same command, PMC passes (each counter group in its
# own run with --kernel-trace only, as MI355X_MICROARCH.md prescribes), phase stamps, every BASELINE config, parity / inverse statistics.
#   usage (GPU box): bash tools/profile_round.sh <out dir under gpurun_out>      then copy what you want judged into profiles/<round>/
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/${1:-round}
mkdir -p $OUT
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_C2.json 2> $OUT/bench.err
for c in C1 C4 C5 C5u; do python3 bench.py --config $c --steps 10 --warmup 3 > $OUT/bench_$c.json 2>> $OUT/bench.err; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 $REPO/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-secondary > $OUT/bench_under_rocprofv3.json 2> $OUT/stats.err
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc$i -o run -- python3 $REPO/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/pmc$i.json 2> $OUT/pmc$i.err
done
# C4: HBM traffic of the feature projection and of the stack kernel
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc$i -o run -- python3 $REPO/bench.py --config C4 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/pmc$i.json 2> $OUT/pmc$i.err
done
cd $REPO
python3 tools/pmc_summary.py $OUT/pmc_C2_f16x2.csv $OUT/pmc_C2_f16x2.json flow_stack_kernel $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 $OUT/pmc4 $OUT/pmc5 $OUT/pmc6 > /dev/null
python3 tools/pmc_summary.py $OUT/pmc_C4_stack.csv $OUT/pmc_C4_stack.json flow_stack_kernel $OUT/pmc7 $OUT/pmc8 > /dev/null
python3 tools/pmc_summary.py $OUT/pmc_C4_featproj.csv $OUT/pmc_C4_featproj.json featproj_kernel $OUT/pmc7 $OUT/pmc8 > /dev/null
cp $OUT/stats/run_kernel_stats.csv $OUT/rocprofv3_kernel_stats_C2.csv 2>/dev/null || find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/rocprofv3_kernel_stats_C2.csv \;
python3 tools/phase_stamps.py --preset C2 > $OUT/stamps_C2.txt 2>&1
python3 tools/phase_stamps.py --preset C4 > $OUT/stamps_C4.txt 2>&1
python3 tools/parity_stats.py > $OUT/parity_stats.jsonl 2>/dev/null
python3 tools/inverse_stats.py > $OUT/inverse_stats.jsonl 2>/dev/null
python3 tools/host_overhead.py > $OUT/host_overhead.jsonl 2>/dev/null
python3 tools/bench_train.py > $OUT/train.json 2>/dev/null
python3 tools/bench_train.py --graph > $OUT/train_graph.json 2>/dev/null
python3 tools/bench_train.py --graph --config C4 --batch 128 > $OUT/train_graph_c4.json 2>/dev/null
find $OUT -name "*.csv" -size +1M -delete
find $OUT -name "*.db" -delete
rm -rf $OUT/stats
ls $OUT
