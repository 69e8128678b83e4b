#!/bin/bash
for i in 1 2; do
  TAG=r03 python scripts/probe/ab_gemm.py 2>&1 | grep "ms |"
  TAG=r02 VIVIT_LIB=scripts/probe/libr02.so python scripts/probe/ab_gemm.py 2>&1 | grep "ms |"
done
python -m pytest tests/test_kernels_gpu.py tests/test_two_stage_gpu.py tests/test_gram_precision_gpu.py -q -x 2>&1 | tail -3
TAG=r03 python scripts/probe/stage_times.py 40960 2>&1 | grep total
TAG=r02 VIVIT_LIB=scripts/probe/libr02.so python scripts/probe/stage_times.py 40960 2>&1 | grep total
python scripts/probe/syrk_flush_headline.py 2>&1 | grep flush=
VIVIT_LIB=scripts/probe/libr02.so python scripts/probe/syrk_flush_headline.py 2>&1 | grep flush=
