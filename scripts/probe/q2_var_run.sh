#!/bin/bash
# Time the product Q2 kernel and its timing-only variants (scripts/probe/q2_variants.sh) at one size on the same box.
cd "$(dirname "$0")/../.."
n=${1:-40960}
python scripts/probe/q2_time1.py $n 2>&1 | grep -v amdgpu.ids
for v in 1 3 4 5 6 7 8; do
  [ -f scripts/probe/libq2v$v.so ] && VIVIT_HIP_ALLOW_STALE=1 VIVIT_HIP_LIB=scripts/probe/libq2v$v.so python scripts/probe/q2_time1.py $n 2>&1 | grep -v amdgpu.ids
done
python scripts/probe/q2_time1.py $n 2>&1 | grep -v amdgpu.ids
