#!/bin/bash
# Timing-only builds of one translation unit with extra defines, as separate libraries scripts/probe/lib<tag>.so (select
# one with VIVIT_HIP_LIB; results of such builds are WRONG by construction).  Run on the CPU box (hipcc cross-compiles).
#   scripts/probe/variants.sh gemm_f32 g64v1 -DG64_VAR=1
set -e
cd "$(dirname "$0")/../.."
unit=$1; tag=$2; shift 2
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-unused-function -Wno-uninitialized "$@" -c vivit_amd/csrc/$unit.hip -o /tmp/${unit}_$tag.o
objs=$(ls vivit_amd/csrc/obj/*.o | grep -v "/$unit.o")
# the product's vivit_hip_source_hash() lives outside csrc/obj/ on purpose: a variant names itself (needs VIVIT_HIP_ALLOW_STALE=1 to load)
echo "const char *vivit_hip_source_hash(void) { return \"variant-$tag\"; }" > /tmp/${tag}_info.c && gcc -c -fPIC /tmp/${tag}_info.c -o /tmp/${tag}_info.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o scripts/probe/lib$tag.so $objs /tmp/${unit}_$tag.o /tmp/${tag}_info.o
echo built scripts/probe/lib$tag.so
