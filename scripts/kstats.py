#!/usr/bin/env python3
"""Register / scratch / occupancy figures of the kernels of one translation unit (hipcc remarks; CPU box, no GPU needed).
    python scripts/kstats.py sy2sb [name-filter] [extra hipcc flags ...]"""
import os
import re
import subprocess
import sys

unit = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "."
extra = sys.argv[3:]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wno-unused-function",
       "-Rpass-analysis=kernel-resource-usage"] + extra + ["-c", os.path.join(root, "vivit_amd", "csrc", unit + ".hip"), "-o", "/tmp/kstats.o"]
out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark:\s+([A-Za-z][^:]*): +(\S+)", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2)
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], stdout=subprocess.PIPE, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name).replace("vivit::", "").replace("void ", "")
    if not re.search(pat, name):
        continue
    g = lambda k: r.get(k, "?")
    print(f"{name[:64]:64s} VGPR {g('VGPRs'):>4} AGPR {g('AGPRs'):>4} SGPR {g('SGPRs'):>4} scratch {g('ScratchSize [bytes/lane]'):>5} "
          f"occ {g('Occupancy [waves/SIMD]'):>2} LDS {g('LDS Size [bytes/block]')}")
if not rows:
    print(out[-3000:])
