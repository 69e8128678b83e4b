#!/bin/bash
for i in 1 2; do
  TAG=sgrp4 python scripts/probe/stage_times.py 40960 2>&1 | grep total
  TAG=sgrp8 VIVIT_LIB=scripts/probe/lib_sgrp8.so python scripts/probe/stage_times.py 40960 2>&1 | grep total
  TAG=sgrp2 VIVIT_LIB=scripts/probe/lib_sgrp2.so python scripts/probe/stage_times.py 40960 2>&1 | grep total
done
