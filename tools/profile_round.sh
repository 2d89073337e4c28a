#!/bin/bash
# Evidence for profiles/<round>/: the driver's bench lines for every BASELINE config, rocprofv3 kernel stats of the same commands, PMC passes
# (each counter group in its own run with --kernel-trace only, as MI355X_MICROARCH.md prescribes; FETCH_SIZE and WRITE_SIZE in separate
# passes), phase stamps, parity / inverse statistics, the fused-projection A/B.
#   usage (GPU box): bash tools/profile_round.sh <out dir under gpurun_out>      then copy what you want judged into profiles/<round>/
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/${1:-round}
mkdir -p $OUT
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_C2.json 2> $OUT/bench.err
for c in C1 C4 C5 C5u; do python3 bench.py --config $c --steps 10 --warmup 3 > $OUT/bench_$c.json 2>> $OUT/bench.err; done
python3 bench.py --config C3 --steps 6 --warmup 2 > $OUT/bench_C3.json 2>> $OUT/bench.err
RNF_FUSED=1 python3 bench.py --config C4 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $OUT/bench_C4_fused.json 2>> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
for c in C2 C4 C5 C5u; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$c -o run -- python3 $REPO/bench.py --config $c --steps 10 --warmup 5 --no-cpu-baseline --no-secondary > $OUT/bench_${c}_under_rocprofv3.json 2> $OUT/stats_$c.err
  find $OUT/stats_$c -name "*kernel_stats.csv" -exec cp {} $OUT/rocprofv3_kernel_stats_$c.csv \;
done
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc$i -o run -- python3 $REPO/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/pmc$i.json 2> $OUT/pmc$i.err
done
# HBM traffic of the conditional / inverse configs (projection pre-pass and stack kernel), and of the fused-projection variant of C4
for c in C4 C5 C5u; do
  for grp in "FETCH_SIZE" "WRITE_SIZE"; do
    rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc_${c}_$grp -o run -- python3 $REPO/bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2> $OUT/pmc_${c}_$grp.err
  done
done
export RNF_FUSED=1
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc_C4fused_$grp -o run -- python3 $REPO/bench.py --config C4 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2> $OUT/pmc_C4fused_$grp.err
done
unset RNF_FUSED
cd $REPO
python3 tools/pmc_summary.py $OUT/pmc_C2_f16x2.csv $OUT/pmc_C2_f16x2.json flow_stack_kernel $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 $OUT/pmc4 $OUT/pmc5 $OUT/pmc6 > /dev/null
export RNF_PMC_ROTATIONS=262144     # the conditional / inverse configs run in chunks of 2^18 rotations per launch
python3 tools/pmc_summary.py $OUT/pmc_C4_stack.csv $OUT/pmc_C4_stack.json flow_stack_kernel $OUT/pmc_C4_FETCH_SIZE $OUT/pmc_C4_WRITE_SIZE > /dev/null
python3 tools/pmc_summary.py $OUT/pmc_C4_featproj.csv $OUT/pmc_C4_featproj.json featproj_kernel $OUT/pmc_C4_FETCH_SIZE $OUT/pmc_C4_WRITE_SIZE > /dev/null
python3 tools/pmc_summary.py $OUT/pmc_C4fused_stack.csv $OUT/pmc_C4fused_stack.json flow_stack_kernel $OUT/pmc_C4fused_FETCH_SIZE $OUT/pmc_C4fused_WRITE_SIZE > /dev/null
python3 tools/pmc_summary.py $OUT/pmc_C5_stack.csv $OUT/pmc_C5_stack.json flow_stack_kernel $OUT/pmc_C5_FETCH_SIZE $OUT/pmc_C5_WRITE_SIZE > /dev/null
python3 tools/pmc_summary.py $OUT/pmc_C5_featproj.csv $OUT/pmc_C5_featproj.json featproj_ksplit_kernel $OUT/pmc_C5_FETCH_SIZE $OUT/pmc_C5_WRITE_SIZE > /dev/null
RNF_PMC_ROTATIONS=1048576 python3 tools/pmc_summary.py $OUT/pmc_C5u_stack.csv $OUT/pmc_C5u_stack.json flow_stack_kernel $OUT/pmc_C5u_FETCH_SIZE $OUT/pmc_C5u_WRITE_SIZE > /dev/null
unset RNF_PMC_ROTATIONS
python3 tools/phase_stamps.py --preset C2 > $OUT/stamps_C2.txt 2>&1
python3 tools/phase_stamps.py --preset C4 > $OUT/stamps_C4.txt 2>&1
RNF_FUSED=1 python3 tools/phase_stamps.py --preset C4 > $OUT/stamps_C4_fused.txt 2>&1
python3 tools/parity_stats.py > $OUT/parity_stats.jsonl 2>/dev/null
python3 tools/inverse_stats.py > $OUT/inverse_stats.jsonl 2>/dev/null
python3 tools/bench_train.py > $OUT/train.json 2>/dev/null
python3 tools/bench_train.py --graph > $OUT/train_graph.json 2>/dev/null
# training: the two backward kernels (16- / 64-rotation workgroups) over the batch sizes, C4, phase stamps, kernel times of one eager iteration
: > $OUT/train_blocks.jsonl
for blk in 16 64; do
  for b in 256 1024 2048 4096 8192 65536; do
    RNF_TRAIN_BLOCK=$blk python3 tools/bench_train.py --graph --batch $b --steps 200 2>/dev/null | sed "s/^{/{\"block\": $blk, /" >> $OUT/train_blocks.jsonl
  done
  RNF_TRAIN_BLOCK=$blk python3 tools/bench_train.py --graph --config C4 --batch 128 --steps 200 2>/dev/null | sed "s/^{/{\"block\": $blk, /" >> $OUT/train_blocks.jsonl
done
python3 tools/phase_stamps_train.py --block 16 > $OUT/train_stamps_c2_b1024_block16.txt 2>&1
python3 tools/phase_stamps_train.py --block 64 > $OUT/train_stamps_c2_b1024_block64.txt 2>&1
python3 tools/phase_stamps_train.py --block 16 --preset C4 --batch 128 > $OUT/train_stamps_c4_b128_block16.txt 2>&1
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_train -o run -- python3 $REPO/tools/bench_train.py --batch 1024 --steps 30 --fused-adam > $OUT/train_under_rocprofv3.json 2> $OUT/stats_train.err)
find $OUT/stats_train -name "*kernel_stats.csv" -exec cp {} $OUT/rocprofv3_kernel_stats_train.csv \;
rm -rf $OUT/stats_train
find $OUT -name "*.csv" -size +1M -delete
find $OUT -name "*.db" -delete
rm -rf $OUT/stats_C2 $OUT/stats_C4 $OUT/stats_C5 $OUT/stats_C5u
rm -rf $OUT/pmc[0-9]* $OUT/pmc_C*_FETCH_SIZE $OUT/pmc_C*_WRITE_SIZE
ls $OUT
