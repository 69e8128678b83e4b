#!/usr/bin/env python3
"""Static check of the asm loads of scripts/probe/q2slide_r05_experiment.hip.txt (round-5 experiment, not the product) (`load_unit_asm`): the four global_load_dwordx4 of a unit write VGPRs that
hipcc believes are defined at once; the data arrives later and is only waited for by the hand-placed `s_waitcnt vmcnt(N)` of
`wait_unit`.  Between the loads and that wait NO instruction may read or write the destination registers (a copy or a spill
there would move garbage).  This script compiles the file to ISA, walks every path from each group of asm loads to the
next counted wait inside qs_apply_kernel<8|12> and fails if a destination register is named on the way.
Also reports scratch (spill) instructions inside the innermost loop.   usage: python scripts/check_q2_asm_loads.py"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vivit_amd import _build  # noqa: E402

# (the experimental kernel of round 5, kept beside this script; it was compiled with -fno-slp-vectorize)
SRC = os.path.join(ROOT, "scripts", "probe", "q2slide_r05_experiment.hip.txt")
cmd = ["/opt/rocm/bin/hipcc"] + [f for f in _build.FLAGS if f != "-Wall"] + ["-fno-slp-vectorize", "-I", os.path.join(ROOT, "vivit_amd", "csrc"),
       "-S", "--cuda-device-only", "-x", "hip", SRC, "-o", "/tmp/q2slide_check.s"]
subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
lines = open("/tmp/q2slide_check.s").read().split("\n")


def regs(tok):
    """VGPR indices named by an operand such as v12 or v[12:15]."""
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", tok):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", tok):
        out.add(int(m.group(1)))
    return out


bad = 0
for kern in ("ILi8E", "ILi12E"):
    start = [i for i, l in enumerate(lines) if l.startswith("_ZN5vivit15qs_apply_kernel" + kern)][0]
    end = [i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end")][0]
    body = [(i, l.strip()) for i, l in enumerate(lines[start:end], start) if l.strip() and not l.strip().startswith(";")]
    # instruction list with labels; build a linear scan: from each asm load group walk forward (following the layout and
    # unconditional structure conservatively: every instruction until the next `s_waitcnt vmcnt(4|0)` inside an ASM block)
    idx = 0
    groups = 0
    depth2_scratch = 0
    in_depth2 = False
    while idx < len(body):
        i, l = body[idx]
        if re.match(r"^\.LBB\d+_\d+:", l):
            in_depth2 = "Depth=2" in lines[i]
        if l.startswith("scratch_") and in_depth2:
            depth2_scratch += 1
            print(f"  note: {kern}: scratch instruction in the innermost loop: line {i}: {l[:70]}")
        if l.startswith("global_load_dwordx4") and "off" not in l.split(",")[-1] and re.search(r", s\[\d+:\d+\]", l):
            # first of four asm loads (saddr form)
            dst = set()
            j = idx
            while j < len(body) and body[j][1].startswith("global_load_dwordx4") and re.search(r", s\[\d+:\d+\]", body[j][1]):
                dst |= regs(body[j][1].split(",")[0])
                j += 1
            groups += 1
            # Paths from the loads to a hand-placed wait (`s_waitcnt vmcnt(N) ; wait_unit`): (a) layout order --
            # falls out of the loop into the wait that follows it; (b) every backward branch met on the way: from its
            # target label (the loop header) forward to the wait inside the loop.
            labels = {m.group(1): n for n, (_, t) in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", t)] if m}

            def scan(k0, what):
                nonlocal_bad = 0
                k, steps, back = k0, 0, []
                while k < len(body) and steps < 8000:
                    lk = body[k][1]
                    if lk.startswith("s_waitcnt vmcnt(") and "wait_unit" in lk:
                        return nonlocal_bad, back, True
                    mb = re.match(r"^s_c?branch\S*\s+(\.LBB\d+_\d+)", lk)
                    if mb and mb.group(1) in labels and labels[mb.group(1)] < k:
                        back.append(labels[mb.group(1)])
                    if not (lk.startswith(".") or lk.startswith("s_") or lk.startswith(";")):
                        ops = lk.split(None, 1)[1] if " " in lk else ""
                        hit = regs(ops) & dst
                        if hit:
                            nonlocal_bad += 1
                            print(f"FAIL {kern}: line {body[k][0]}: `{lk}` names v{sorted(hit)} on the way ({what}) from the asm "
                                  f"loads (line {i}) to their wait")
                    k += 1
                    steps += 1
                return nonlocal_bad, back, False

            b1, back, ok1 = scan(j, "layout order")
            bad += b1
            if not ok1:
                bad += 1
                print(f"FAIL {kern}: loads at line {i}: no hand-placed wait found in layout order")
            for tgt in sorted(set(back)):
                b2, _, ok2 = scan(tgt, f"back-edge to {body[tgt][1]}")
                bad += b2
                if not ok2:
                    bad += 1
                    print(f"FAIL {kern}: loads at line {i}: no hand-placed wait after the back-edge target {body[tgt][1]}")
            idx = j
            continue
        idx += 1
    print(f"{kern}: {groups} asm load groups checked, {depth2_scratch} scratch instructions in the innermost loop")
print("OK" if bad == 0 else f"{bad} violations")
sys.exit(1 if bad else 0)
