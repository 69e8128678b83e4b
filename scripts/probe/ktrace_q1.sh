#!/bin/bash
# Kernel trace of one full-size symeig (scripts/probe/stage_times.py <n>) and the per-kernel breakdown of its Q1 back-transformation.
#   scripts/probe/ktrace_q1.sh [n]      (run on the GPU box; outputs under gpurun_out/r06/ktrace_q1)
export TMPDIR=/tmp
R=$PWD
N=${1:-40960}
O=$R/gpurun_out/r06/ktrace_q1
mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace -d $O -o kt --output-format csv -- python3 $R/scripts/probe/stage_times.py $N > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
python3 $R/scripts/probe/q1_breakdown.py $O/kt_kernel_trace.csv | tee $O/q1_breakdown.txt
rm -f $O/kt_kernel_trace.csv
