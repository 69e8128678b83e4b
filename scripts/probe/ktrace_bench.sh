#!/bin/bash
# rocprofv3 --kernel-trace --stats of the bench command without the extras (the kernel-trace part of scripts/r05_measure.sh alone)
set -o pipefail
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r05k; mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/ktrace -o kt --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-verify --no-secondary --no-configs > $O/ktrace.log 2>&1 || { tail -5 $O/ktrace.log; exit 1; }
cd $R
find $O -name "*_kernel_trace.csv" -size +2M -delete
head -8 $O/ktrace/kt_kernel_stats.csv | cut -c1-150; tail -1 $O/ktrace.log | cut -c1-300
