#!/bin/bash
# Matrix-pipe busy cycles of the Gram SYRK on the benchmark's own first-layer factor (rocprofv3 --pmc; the N(0,1) pass is part of scripts/r05_measure.sh)
set -o pipefail
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r05b; mkdir -p $O
cd /tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-include-regex "gemm256_bx" --kernel-trace -d $O/pmc_syrk_mfma_bench -o p --output-format csv -- python3 $R/scripts/pmc_syrk_full.py bench > $O/pmc_syrk_mfma_bench.log 2>&1 || { tail -5 $O/pmc_syrk_mfma_bench.log; exit 1; }
cd $R
find $O -name "*.csv" | head; tail -2 $O/pmc_syrk_mfma_bench.log
