"""Sums of the PMC passes of scripts/r06_measure.sh per kernel family: counter totals, dispatch counts and kernel time (from the
kernel trace of the same pass), and the derived figures bench.py quotes (traffic with the gfx950 wide-read correction 2 x FETCH_SIZE,
matrix-pipe busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs over GRBM_GUI_ACTIVE / 8 XCDs, clock = GRBM_GUI_ACTIVE / 8 / time).
usage: python scripts/r06_pmc_summary.py gpurun_out/r06m"""
import csv
import glob
import os
import sys

root = sys.argv[1]


def totals(d, pattern):
    out, disp = {}, {}
    for f in glob.glob(os.path.join(root, d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if pattern in row.get("Kernel_Name", ""):
                out[row["Counter_Name"]] = out.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                disp[row["Counter_Name"]] = disp.get(row["Counter_Name"], 0) + 1
    t = 0.0
    for f in glob.glob(os.path.join(root, d, "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if pattern in row.get("Kernel_Name", ""):
                t += (float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) * 1e-9
    return out, disp, t


for tag, pat in (("syrk", "gemm256_bx"), ("syrk", "bx_split"), ("q2", "qs_apply"), ("q2", "qs_prepare")):
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        tot, disp, t = totals(f"pmc_{tag}_{c}", pat)
        if tot:
            print(f"{tag:5s} {pat:12s} {c}: {tot[c]:.4e} KiB over {disp[c]} dispatches" + (f", kernel time {t:.3f} s" if t else ""))
for d, pat in (("pmc_syrk_mfma", "gemm256_bx"), ("pmc_syrk_mfma_bench", "gemm256_bx"), ("pmc_q2_mfma", "qs_apply")):
    tot, disp, t = totals(d, pat)
    if tot and t:
        busy = tot["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (tot["GRBM_GUI_ACTIVE"] / 8.0)
        ghz = tot["GRBM_GUI_ACTIVE"] / 8.0 / t / 1e9
        print(f"{d:22s} {pat}: MFMA busy cycles {tot['SQ_VALU_MFMA_BUSY_CYCLES']:.4e}, GRBM_GUI_ACTIVE {tot['GRBM_GUI_ACTIVE']:.4e}, "
              f"{disp['GRBM_GUI_ACTIVE']} dispatches, {t:.3f} s -> pipe busy {busy:.3f}, clock {ghz:.3f} GHz, busy x clock / 2.4 = {busy * ghz / 2.4:.3f}")
