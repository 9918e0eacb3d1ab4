#!/bin/bash
# usage: mfma_pass.sh <outdir> <bench flags...>
OUT=$1; shift
export TMPDIR=/tmp
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -o p -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-fast-tiers --no-f1024 "$@" > /dev/null 2> $OUT/pmc_mfma.log
M=$(find $OUT/pmc_mfma -name "*counter_collection.csv" | head -1)
python tools/pmc_table.py $M > $OUT/pmc_mfma.md 2>> $OUT/pmc_mfma.log
rm -rf $OUT/pmc_mfma
