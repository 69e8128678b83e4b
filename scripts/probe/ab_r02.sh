#!/bin/bash
for i in 1 2; do
  python scripts/probe/syrk_flush_headline.py 2>&1 | grep flush=
  VIVIT_BX_SYNC=1 python scripts/probe/syrk_flush_headline.py 2>&1 | grep flush=
done
