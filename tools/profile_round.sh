#!/bin/bash
# Evidence for profiles/<round>/ (rounds 4-6).  The driver's bench line now carries every BASELINE config and collects its own counters
# (bench.py runs the FETCH_SIZE / WRITE_SIZE / SQ rocprofv3 passes as children of the run, each group in its own process, as
# MI355X_MICROARCH.md prescribes), so this script only adds what the line does not hold: the rocprofv3 --kernel-trace --stats summaries of the
# same per-config commands, the kernel resource table, training numbers, phase stamps.
#   usage (GPU box): bash tools/profile_round.sh <out dir under gpurun_out>      then copy what you want judged into profiles/<round>/
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/${1:-round}
mkdir -p $OUT
# 1. the default invocation, exactly as the driver runs it (+ the live PMC summary kept as pmc_live.json: bench.py replays it only if rocprofv3
#    is missing on a box AND the kernel sources are unchanged)
#    bench_default.json = the FULL record (the BENCH_FULL line / bench_full.json); bench_compact.json = the last stdout line the driver parses
python3 bench.py --save-pmc $OUT --full-out $OUT/bench_default.json > $OUT/bench_stdout.txt 2> $OUT/bench.err
tail -n 1 $OUT/bench_stdout.txt > $OUT/bench_compact.json
rm -f $OUT/bench_stdout.txt
# 2. kernel-trace statistics of the per-config commands (no PMC in these runs: timing and counters are never mixed)
cd /tmp && export TMPDIR=/tmp
for c in C1 C2 C3 C4 C5 C5u C4q C5q; do
  # (the bench lines of these profiled runs are not kept: their times include the profiler; the kernel statistics are what is read)
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$c -o run -- python3 $REPO/bench.py --config $c --steps 10 --warmup 5 --no-cpu-baseline --no-secondary --no-pmc --full-out /dev/null > /dev/null 2> $OUT/stats_$c.err
  find $OUT/stats_$c -name "*kernel_stats.csv" -exec cp {} $OUT/rocprofv3_kernel_stats_$c.csv \;
  rm -rf $OUT/stats_$c
done
# (round 6) the strict arithmetic's kernels: the same commands with RNF_PRECISION=bf16x3 (exported BEFORE rocprofv3: no exec hop behind it)
export RNF_PRECISION=bf16x3
for c in C2 C5u C4q; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_b3_$c -o run -- python3 $REPO/bench.py --config $c --steps 10 --warmup 5 --no-cpu-baseline --no-secondary --no-pmc --full-out /dev/null > /dev/null 2> $OUT/stats_b3_$c.err
  find $OUT/stats_b3_$c -name "*kernel_stats.csv" -exec cp {} $OUT/rocprofv3_kernel_stats_${c}_bf16x3.csv \;
  rm -rf $OUT/stats_b3_$c
done
unset RNF_PRECISION
cd $REPO
# 3. training: eager iteration with the reference's plain Adam on the flattened flow (what get_flow hands train_uncondition.py), the classic
#    per-tensor flow beside it, the HIP-graph replay, the batch-size table of the two backward kernels
: > $OUT/train.jsonl
python3 tools/bench_train.py >> $OUT/train.jsonl 2>/dev/null
python3 tools/bench_train.py --fused-adam >> $OUT/train.jsonl 2>/dev/null
python3 tools/bench_train.py --classic >> $OUT/train.jsonl 2>/dev/null
python3 tools/bench_train.py --classic --fused-adam >> $OUT/train.jsonl 2>/dev/null
python3 tools/bench_train.py --graph >> $OUT/train.jsonl 2>/dev/null
python3 tools/bench_train.py --config C4 --batch 128 >> $OUT/train.jsonl 2>/dev/null
for b in 256 4096 8192 65536; do python3 tools/bench_train.py --graph --batch $b --steps 100 >> $OUT/train.jsonl 2>/dev/null; done
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_train -o run -- python3 $REPO/tools/bench_train.py --batch 1024 --steps 30 > $OUT/train_under_rocprofv3.json 2> $OUT/stats_train.err)
find $OUT/stats_train -name "*kernel_stats.csv" -exec cp {} $OUT/rocprofv3_kernel_stats_train.csv \;
rm -rf $OUT/stats_train
# 4. phase stamps of the forward stack kernels, parity / inverse statistics
python3 tools/phase_stamps.py --preset C2 > $OUT/stamps_C2.txt 2>&1
python3 tools/phase_stamps.py --preset C4 > $OUT/stamps_C4.txt 2>&1
python3 tools/parity_stats.py --all-forward > $OUT/parity_stats.jsonl 2>/dev/null
python3 tools/inverse_stats.py > $OUT/inverse_stats.jsonl 2>/dev/null
# (the kernel resource table needs no GPU: `python tools/kernel_resources.py > profiles/<round>/kernel_resources.txt` in the build container)
find $OUT -name "*.csv" -size +1M -delete
find $OUT -name "*.db" -delete
ls $OUT
