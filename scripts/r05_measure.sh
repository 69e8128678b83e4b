#!/bin/bash
# Round-5 measurement pass on the GPU box (final code): the bench line, kernel-trace stats of the same command without the
# extras, PMC passes (separate runs: FETCH_SIZE / WRITE_SIZE / matrix-pipe busy) for the Gram SYRK, the
# sliding-window Q2 kernel and the band reduction's streaming panel product.  Outputs under gpurun_out/r05; copy what is to be kept into profiles/.
set -o pipefail
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r05
mkdir -p $O
if [ "$1" != "nobench" ]; then
  python bench.py --steps 3 --warmup 1 > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
  tail -c 400 $O/bench.json; echo
fi
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/ktrace -o kt --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-verify --no-secondary --no-configs > $O/ktrace.log 2>&1 || { tail -5 $O/ktrace.log; exit 1; }
echo "kernel trace done"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-include-regex "gemm256_bx|bx_split" --kernel-trace -d $O/pmc_syrk_$c -o p --output-format csv -- python3 $R/scripts/pmc_syrk_full.py > $O/pmc_syrk_$c.log 2>&1 || { tail -5 $O/pmc_syrk_$c.log; exit 1; }
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-include-regex "gemm256_bx" --kernel-trace -d $O/pmc_syrk_mfma -o p --output-format csv -- python3 $R/scripts/pmc_syrk_full.py > $O/pmc_syrk_mfma.log 2>&1 || { tail -5 $O/pmc_syrk_mfma.log; exit 1; }
echo "syrk pmc done"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-include-regex "qs_apply|qs_prepare" --kernel-trace -d $O/pmc_q2_$c -o p --output-format csv -- python3 $R/scripts/probe/q2_time1.py 40960 1 > $O/pmc_q2_$c.log 2>&1 || { tail -5 $O/pmc_q2_$c.log; exit 1; }
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-include-regex "qs_apply" --kernel-trace -d $O/pmc_q2_mfma -o p --output-format csv -- python3 $R/scripts/probe/q2_time1.py 40960 1 > $O/pmc_q2_mfma.log 2>&1 || { tail -5 $O/pmc_q2_mfma.log; exit 1; }
echo "q2 pmc done"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-include-regex "gemm64_bx|g64_split" --kernel-trace -d $O/pmc_g64_$c -o p --output-format csv -- python3 $R/scripts/probe/panel_product1.py 40960 > $O/pmc_g64_$c.log 2>&1 || { tail -5 $O/pmc_g64_$c.log; exit 1; }
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-include-regex "gemm64_bx" --kernel-trace -d $O/pmc_g64_mfma -o p --output-format csv -- python3 $R/scripts/probe/panel_product1.py 40960 > $O/pmc_g64_mfma.log 2>&1 || { tail -5 $O/pmc_g64_mfma.log; exit 1; }
echo "panel product pmc done"
cd $R
find $O -name "*_kernel_trace.csv" -size +2M -delete   # gpurun copies at most 64 MiB back
find $O -name "*.csv" | head -40
du -sh $O
