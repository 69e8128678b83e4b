#!/bin/bash
# Timing-only builds of the sliding-window Q2 kernel (csrc/q2slide.hip, -DQS_VAR=1..4: wrong results) as separate
# libraries scripts/probe/libq2v<N>.so; select one with VIVIT_HIP_LIB.  Run on the CPU box (hipcc cross-compiles).
set -e
cd "$(dirname "$0")/../.."
for v in "$@"; do
  case $v in [0-9]*) def="-DQS_VAR=$v";; *) def="-D${v//,/ -D}";; esac   # NAME=1,OTHER=2 -> several defines
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-unused-function $def -c vivit_amd/csrc/q2slide.hip -o /tmp/q2slide_v$v.o
  objs=$(ls vivit_amd/csrc/obj/*.o | grep -v q2slide.o)
  # the product's vivit_hip_source_hash() lives outside csrc/obj/ on purpose: a variant names itself (needs VIVIT_HIP_ALLOW_STALE=1 to load)
  echo "const char *vivit_hip_source_hash(void) { return \"variant-q2v$v\"; }" > /tmp/q2v${v}_info.c && gcc -c -fPIC /tmp/q2v${v}_info.c -o /tmp/q2v${v}_info.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o scripts/probe/libq2v$v.so $objs /tmp/q2slide_v$v.o /tmp/q2v${v}_info.o
  echo built scripts/probe/libq2v$v.so
done
