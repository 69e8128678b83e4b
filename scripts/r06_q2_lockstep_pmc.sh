#!/bin/bash
# Round 6: the per-XCD lock-step of Q2's image loaders (VIVIT_Q2_LOCKSTEP = window in blocks; csrc/q2slide.hip): time and
# FETCH_SIZE of qs_apply_kernel per window, one solve's worth of Q2 at n = 40 960 (scripts/probe/q2_time1.py 40960 1).
set -o pipefail
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r06/q2_lockstep_pmc
mkdir -p $O
cd /tmp
for w in 0 2 4 8 16; do
  VIVIT_Q2_LOCKSTEP=$w python3 $R/scripts/probe/q2_time1.py 40960 3 2>&1 | grep -v amdgpu.ids | sed "s/^/window $w: /"
  VIVIT_Q2_LOCKSTEP=$w rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "qs_apply" --kernel-trace -d $O/w$w -o p --output-format csv -- python3 $R/scripts/probe/q2_time1.py 40960 1 > $O/w$w.log 2>&1 || { tail -5 $O/w$w.log; exit 1; }
  python3 - $O/w$w <<'PY'
import csv, glob, sys
tot, n = 0.0, 0
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if row.get("Counter_Name") == "FETCH_SIZE" and "qs_apply" in row.get("Kernel_Name", ""):
            tot += float(row["Counter_Value"]); n += 1
print(f"  FETCH_SIZE {tot:.4e} KiB over {n} dispatches = {tot * 1024 / 2 * 2 / 1e12:.3f} TB raw x 1 (two solves: warm-up + 1) -> per solve {tot * 1024 / 2 / 1e12:.3f} TB (x 2 for the gfx950 wide-read correction: {tot * 1024 / 1e12:.3f} TB)")
PY
done
find $O -name "*_kernel_trace.csv" -size +2M -delete
