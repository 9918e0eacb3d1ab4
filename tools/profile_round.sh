#!/bin/bash
# Round profile of the default bench (run on the GPU box from the repo root): kernel trace + the two HBM-traffic PMC passes +
# one MFMA / LDS counter pass. Writes under gpurun_out/prof_$1; summaries are then copied into profiles/ by hand.
#   bash tools/profile_round.sh r03 [extra bench.py flags]
set -u
TAG=${1:-r03}; shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fast-tiers --no-f1024 --no-latency --detail-file $OUT/bench_detail.json $*"
OMGSR_KERNEL_TABLE=$OUT/ktable.md rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python3 $BENCH > $OUT/bench_under_trace.json 2> $OUT/trace.log
DB=$(find $OUT/trace -name "*.db" | head -1)
python tools/rocpd_summary.py $DB $OUT/kernel_stats.md > /dev/null
# three pipeline passes: the first also runs the one-time constant folding; tools/traffic_summary.py keeps the last two (steady state)
ONE="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-fast-tiers --no-f1024 --no-latency --detail-file $OUT/bench_detail_pmc.json $*"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$C -o p -- python3 $ONE > /dev/null 2> $OUT/pmc_$C.log
done
F=$(find $OUT/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1)
W=$(find $OUT/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python tools/traffic_summary.py $F $W $OUT/traffic.json > /dev/null
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -o p -- python3 $ONE > /dev/null 2> $OUT/pmc_mfma.log
M=$(find $OUT/pmc_mfma -name "*counter_collection.csv" | head -1)
python tools/pmc_table.py $M > $OUT/pmc_mfma.md 2>> $OUT/pmc_mfma.log
rm -rf $OUT/trace $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_mfma     # raw traces are large; the summaries stay
ls -la $OUT
