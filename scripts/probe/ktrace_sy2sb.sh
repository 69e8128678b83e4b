#!/bin/bash
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r06/ktrace_sy2sb
mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --stats -d $O -o kt --output-format csv -- python3 $R/scripts/probe/stage_times.py 20480 > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
grep -E "w_coef|larft|skinny64|gemm_tsk_kernel|qr_persist|sb_panel|coef_gather|copyBufferRect|fillBuffer" $O/kt_kernel_stats.csv | cut -d, -f1-4 | cut -c1-140
