#!/bin/bash
# timing of the band reduction at n = 40960 for SGRP in {2, 4, 8} x VIVIT_GEMM256_KMIN in {1024, 512} (bf16 pipe on)
cd "$(dirname "$0")/../.."
for lib in vivit_amd/libvivit_hip.so scripts/probe/libvivit_sgrp4.so scripts/probe/libvivit_sgrp8.so; do
  for kmin in 1024 512; do
    echo -n "KMIN=$kmin "
    VIVIT_GEMM256_KMIN=$kmin python scripts/probe/run_variant_sy2sb.py $lib 2>&1 | grep -v Warn | tail -1
  done
done
