#!/bin/bash
# Round-6 measurement pass on the GPU box (final code): the bench line with the driver's flags, kernel-trace stats of the same
# command without the extras, PMC passes (separate runs: FETCH_SIZE / WRITE_SIZE / matrix-pipe busy + clock) for the Gram SYRK
# on N(0,1) data and on the bench's own factor, and for the sliding-window Q2 kernel.  Outputs under gpurun_out/r06m; what is to
# be kept is copied into profiles/ by hand.    scripts/r06_measure.sh [nobench]
set -o pipefail
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r06m
mkdir -p $O
if [ "$1" != "nobench" ]; then
  python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
  tail -c 300 $O/bench.json; echo
fi
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/ktrace -o kt --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-verify --no-secondary --no-configs > $O/ktrace.log 2>&1 || { tail -5 $O/ktrace.log; exit 1; }
echo "kernel trace done"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-include-regex "gemm256_bx|bx_split" --kernel-trace -d $O/pmc_syrk_$c -o p --output-format csv -- python3 $R/scripts/pmc_syrk_full.py > $O/pmc_syrk_$c.log 2>&1 || { tail -5 $O/pmc_syrk_$c.log; exit 1; }
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-include-regex "gemm256_bx" --kernel-trace -d $O/pmc_syrk_mfma -o p --output-format csv -- python3 $R/scripts/pmc_syrk_full.py > $O/pmc_syrk_mfma.log 2>&1 || { tail -5 $O/pmc_syrk_mfma.log; exit 1; }
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-include-regex "gemm256_bx" --kernel-trace -d $O/pmc_syrk_mfma_bench -o p --output-format csv -- python3 $R/scripts/pmc_syrk_full.py bench > $O/pmc_syrk_mfma_bench.log 2>&1 || { tail -5 $O/pmc_syrk_mfma_bench.log; exit 1; }
echo "syrk pmc done"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-include-regex "qs_apply|qs_prepare" --kernel-trace -d $O/pmc_q2_$c -o p --output-format csv -- python3 $R/scripts/probe/q2_time1.py 40960 1 > $O/pmc_q2_$c.log 2>&1 || { tail -5 $O/pmc_q2_$c.log; exit 1; }
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-include-regex "qs_apply" --kernel-trace -d $O/pmc_q2_mfma -o p --output-format csv -- python3 $R/scripts/probe/q2_time1.py 40960 1 > $O/pmc_q2_mfma.log 2>&1 || { tail -5 $O/pmc_q2_mfma.log; exit 1; }
echo "q2 pmc done"
cd $R
find $O -name "*_kernel_trace.csv" -size +2M -delete   # gpurun copies at most 64 MiB back
python3 scripts/r06_pmc_summary.py $O | tee $O/pmc_summary.txt
du -sh $O
