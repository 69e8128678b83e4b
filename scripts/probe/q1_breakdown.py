"""Reads a rocprofv3 --kernel-trace CSV of scripts/probe/stage_times.py and prints, for the LAST symeig in it, the kernels
between the last q2_apply16 launch and the end (= the Q1 back-transformation + output): time per kernel name and the idle
gaps between consecutive kernels."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last_q2 = max(i for i, r in enumerate(rows) if "q2_apply" in r["Kernel_Name"] or "qs_apply" in r["Kernel_Name"])
tail = rows[last_q2 + 1:]
by = collections.defaultdict(lambda: [0, 0])
gap = 0
prev_end = int(rows[last_q2]["End_Timestamp"])
t0 = prev_end
for r in tail:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][-60:]
    key = name
    if "gemm" in name:
        key = name + " grid=" + r.get("Grid_Size_X", r.get("Grid_Size", "?")) + "x" + r.get("Grid_Size_Y", "")
    by[key][0] += e - s
    by[key][1] += 1
    gap += max(0, s - prev_end)
    prev_end = max(prev_end, e)
print(f"window {1e-6 * (prev_end - t0):.1f} ms, idle gaps {1e-6 * gap:.1f} ms")
agg = collections.defaultdict(lambda: [0, 0])
for k, (t, c) in by.items():
    agg[k.split(" grid=")[0]][0] += t
    agg[k.split(" grid=")[0]][1] += c
for k, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"{1e-6 * t:9.2f} ms {c:6d} x  {k}")
print("largest gemm launches by grid:")
for k, (t, c) in sorted(((k, v) for k, v in by.items() if "grid=" in k), key=lambda kv: -kv[1][0])[:12]:
    print(f"{1e-6 * t:9.2f} ms {c:6d} x  {k}")
