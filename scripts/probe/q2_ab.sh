#!/bin/bash
# Same-box A/B of the Q2 sliding-window kernel: scripts/probe/libq2old.so (q2slide.hip of the previous commit, the rest of
# the current tree) against the current library, interleaved twice.  usage: scripts/probe/q2_ab.sh [n ...]
cd "$(dirname "$0")/../.."
for rep in 1 2; do
  echo "== old (libq2old.so) rep $rep"; VIVIT_HIP_ALLOW_STALE=1 VIVIT_HIP_LIB=scripts/probe/libq2old.so python scripts/probe/q2_time.py "$@" 2>&1 | grep -v amdgpu.ids
  echo "== new rep $rep"; python scripts/probe/q2_time.py "$@" 2>&1 | grep -v amdgpu.ids
done
