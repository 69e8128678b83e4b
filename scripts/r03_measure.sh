#!/bin/bash
# Round-3 measurement pass on the GPU box: in-kernel clock, the bench line, kernel-trace stats, FETCH/WRITE counters.
set -o pipefail
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03
mkdir -p $O
python scripts/probe/bx_clock.py real  2>&1 | grep -v amdgpu.ids | tee $O/bx_clock_real.txt
python scripts/probe/bx_clock.py randn 2>&1 | grep -v amdgpu.ids | tee $O/bx_clock_randn.txt
python bench.py --steps 3 --warmup 1 > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
tail -c 600 $O/bench.json; echo
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/ktrace -o kt --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-verify --no-secondary --no-configs > $O/ktrace.log 2>&1 || { tail -5 $O/ktrace.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "gemm256_bx|bx_split" --kernel-trace -d $O/pmc_fetch -o f --output-format csv -- python3 $R/scripts/pmc_syrk_full.py > $O/pmc_fetch.log 2>&1 || { tail -5 $O/pmc_fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "gemm256_bx|bx_split" --kernel-trace -d $O/pmc_write -o w --output-format csv -- python3 $R/scripts/pmc_syrk_full.py > $O/pmc_write.log 2>&1 || { tail -5 $O/pmc_write.log; exit 1; }
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-include-regex "gemm256_bx" --kernel-trace -d $O/pmc_mfma -o m --output-format csv -- python3 $R/scripts/pmc_syrk_full.py > $O/pmc_mfma.log 2>&1 || { tail -5 $O/pmc_mfma.log; exit 1; }
cd $R
find $O -name "*_kernel_trace.csv" -size +2M -delete   # gpurun copies at most 64 MiB back
find $O -name "*.csv" | head -30
du -sh $O
