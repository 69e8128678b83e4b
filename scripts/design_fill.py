"""DESIGN.md = scripts/design_template.md with the numbers of one bench.py JSON line filled in (phase table, measured block,
config table).   usage: python scripts/design_fill.py profiles/r06_bench_n40960_s16.json"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
bench = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
t = open(os.path.join(ROOT, 'scripts', 'design_template.md')).read()
kern = {"Gram build": "`gemm256_bx_kernel<6, asm>` + `bx_split_kernel` (+ `syrk_diag_kernel`)",
        "prepare": "`trd_scan` / `symmetrize`", "panel QR": "`qr_persist_kernel` + ~15 small launches per panel",
        "streaming panel": "`gemm64_bx_kernel`", "trailing updates": "`gemm256_bx_kernel` (SYRK, K = 1024, mirrored)",
        "sb2st": "`sb2st_persist_kernel`", "tridiagonal eigenproblem": "`dc_*` kernels (`stedc.hip`)", "Q2": "`qs_prepare_kernel` + `qs_apply_kernel<12>`",
        "Q1": "`gemm256_bx_kernel` x 3 per super-block + `bt_*`", "sort": "`dc_transpose_out_kernel`"}
rows = []
for r in bench["roofline_phases"]:
    k = next((v for key, v in kern.items() if key in r["stage"]), "")
    if r.get("bound") == "mfma":
        work = f"{r['flops']:.3g} flop on {r.get('pipe', 'bf16 MFMA, 6 partial products per fp32 product')}"; ach = f"{r['achieved']:.1f} / {r['peak']:.1f} TF"
    elif r.get("bound") == "hbm":
        work = f"{r['bytes']:.3g} B"; ach = f"{r['achieved']:.0f} / {r['peak']:.0f} GB/s"
    elif r.get("bound") == "latency":
        work = f"latency model {r['model_seconds']:.3f} s"; ach = "model / measured"
    else:
        work, ach = "", ""
    rows.append(f"| {r['stage'].replace('|', '/')} | {k} | {r.get('bound')} | {work} | {r['seconds']:.3f} | {r.get('frac', float('nan')):.3f} ({ach}) |")
t = t.replace("{{PHASE_TABLE}}", "\n".join(rows))
ph = bench["phases"]
t = t.replace("{{STEP_S}}", f"{bench['ms_per_step'] / 1e3:.3f}").replace("{{GRAM_S}}", f"{ph['gram_s']:.2f}").replace("{{EIG_S}}", f"{ph['symeig_s']:.2f}")
t = t.replace("{{BENCH_FILE}}", os.path.basename(sys.argv[1])).replace("{{VALUE}}", f"{bench['value']:.0f}").replace("{{BOX}}", "one MI355X, 20 steps after 5 warm-up steps")
rf, cb, v = bench["roofline"], bench["cpu_baseline"], bench["verified"]
sp = bench["ms_per_step_spread"]
blk = (f"Last run with the driver's flags (`python bench.py --steps 20 --warmup 5`, `profiles/{os.path.basename(sys.argv[1])}`): **{bench['value']:.0f} eigenpairs/s** "
       f"({bench['ms_per_step']:.1f} ms per step; per-step min / median / max {sp['min']:.1f} / {sp['median']:.1f} / {sp['max']:.1f} ms); `roofline`: {rf['achieved']:.1f} TF of "
       f"fp32 work = **{rf['frac']:.3f}** of the bf16 / 6 ceiling (= {rf['vs_fp32_mfma_peak']:.2f} x the fp32 MFMA peak), traffic {rf['traffic'] / 1e12:.2f} TB per SYRK against "
       f"0.069 TB algorithmic (tile re-reads through L2 / MALL: not the limiter), clock under load {{CLOCK_TXT}}; `cpu_baseline` (oracle, {cb['cores']} threads of an "
       f"{cb['cpu_model']}): materialised {cb['materialised_eigenpairs_per_s']:.1f}, factorised {cb['factorised_eigenpairs_per_s']:.1f} eigenpairs/s ⇒ "
       f"**{bench['value'] / cb['materialised_eigenpairs_per_s']:.0f}x** (like for like; fit residuals ≤ {max(cb['fit_worst_log_residual'].values()):.2f}); `verified`: Gram entries "
       f"{v['gram']['entry_err']:.2e} of √(G_ii G_jj), 2-norm residual over all eigenpairs {v['symeig']['residual_2norm_fp64']:.1e} λmax (target 1e-5), orthonormality "
       f"{v['symeig']['orth_err']:.1e}.")
t = t.replace("{{MEASURED_BLOCK}}", blk)
crow = []
for c in bench.get("configs", []):
    if "error" in c:
        crow.append(f"| {c['config']} | error | | |")
    else:
        crow.append(f"| {c['config'].replace('|', '/')} | {c['backward_s'] * 1e3:.1f} ms | {c['factors_s'] * 1e3:.1f} ms | {c['total_s'] * 1e3:.1f} ms |")
t = t.replace("{{CONFIG_TABLE}}", "\n".join(crow))
t = t.replace("{CLOCK_TXT}", "2.08 GHz at 82.2 % matrix-pipe busy on the bench's own factor, 1.78 GHz at 82.9 % on N(0,1) data "
              "(`profiles/r06_pmc_s16_summary.txt`; the 32 x 32 x 16 form: 1.96 GHz at 82.3 %; round 5: 2.00 GHz at 77.1 %)")
open(os.path.join(ROOT, 'DESIGN.md'), 'w').write(t)
print(len(t.encode()), "bytes")
