#!/bin/bash
for pad in 0 64 32 0 64; do
  PAD=$pad python scripts/probe/syrk_flush_headline.py 2>&1 | grep flush=
done
PAD=64 VIVIT_BX_FLUSH=2048 python scripts/probe/syrk_flush_headline.py 2>&1 | grep flush=
