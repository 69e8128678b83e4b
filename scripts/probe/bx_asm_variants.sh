#!/bin/bash
# Libraries scripts/probe/libbx<NAME>.so whose gemm256_bx_kernel<6, true> runs the block BX_KLOOP_ASM_<NAME> of csrc/bx_kloop_asm.inc
# (scripts/gen_bx_kloop.py: E* = experiments with the product's arithmetic, T* = timing-only attribution builds), asm loop on by
# default, in-kernel stamps for scripts/probe/bx_timeline.py (-DBX_STAMP=2).  "P" = the product's block, "P32" = the 32x32x16 form
# (-DBX_SHAPE16=0; its E* / T* variants need SHAPE=-DBX_SHAPE16=0 as well).
#   scripts/probe/bx_asm_variants.sh P E1 E2 T1 ...        (NOSTAMP=1: without the stamps, for whole-product timings)
cd "$(dirname "$0")/../.."
python scripts/gen_bx_kloop.py --variants > /dev/null   # the experiment blocks (csrc/bx_kloop_asm_variants.inc, git-ignored)
stamp="-DBX_STAMP=2"; [ -n "$NOSTAMP" ] && stamp=""
for v in "$@"; do
  if [ "$v" = P ]; then sel=""; elif [ "$v" = P32 ]; then sel="-DBX_SHAPE16=0"; else sel="-DBX_KLOOP_TEXT_OVERRIDE=BX_KLOOP_ASM_${v}_TEXT -DBX_KLOOP_CLOB_OVERRIDE=BX_KLOOP_ASM_${v}_CLOBBERS -DBX_KLOOP_UNROLL_OVERRIDE=BX_KLOOP_ASM_${v}_UNROLL"; fi
  scripts/probe/variants.sh gemm_f32 bx$v $stamp -DBX_ASM_DEFAULT=1 $SHAPE $sel > /tmp/bxvar_$v.log 2>&1 &
  while [ "$(jobs -r | wc -l)" -ge 5 ]; do sleep 2; done
done
wait
for v in "$@"; do tail -n 1 /tmp/bxvar_$v.log; done
