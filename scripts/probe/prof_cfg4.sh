#!/bin/bash
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/cfg4
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/cfg4 -o t --output-format csv -- python3 $R/bench_configs.py 4 > $R/gpurun_out/cfg4/log.txt 2>&1
cd $R
head -25 gpurun_out/cfg4/t_kernel_stats.csv | cut -d, -f1-4 | cut -c1-170
find gpurun_out/cfg4 -name "*_kernel_trace.csv" -size +5M -delete
