"""The one generated source of the library -- csrc/bx_kloop_asm.inc, the hand-scheduled K loops of gemm256_bx_kernel<6, true> -- must be
exactly what scripts/gen_bx_kloop.py writes (nobody edits the blocks by hand; a generator change without regeneration is caught here),
and the blocks must keep the properties their arithmetic contract rests on.  Runs without a GPU."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))

import gen_bx_kloop as gen  # noqa: E402


def test_committed_file_is_the_generators_output():
    have = open(os.path.join(ROOT, "vivit_amd", "csrc", "bx_kloop_asm.inc")).read()
    assert have == gen.product_text(), "vivit_amd/csrc/bx_kloop_asm.inc is stale: run `python scripts/gen_bx_kloop.py`"


def _tiles(lines, first_line_prefix, unroll):
    body = lines[lines.index("BXK_G0_%=:") + 1:]
    tiles, cur = [], []
    for l in body:           # split the unrolled trip into tiles at the per-tile address setup
        if l.startswith(first_line_prefix) and cur:
            tiles.append(cur)
            cur = []
        cur.append(l)
    tiles.append(cur)
    assert len(tiles) == unroll
    return body, tiles


def _common_tile_checks(tile, reads_per_gap):
    """One barrier and one vmcnt(0) per tile, the twelve requests behind them, at most `reads_per_gap` fragment reads between two
    MFMAs (except behind the tile's last MFMA)."""
    assert sum(l == "s_barrier" for l in tile) == 1 and sum(l == "s_waitcnt vmcnt(0)" for l in tile) == 1
    bar = tile.index("s_barrier")
    assert all(i > bar for i, l in enumerate(tile) if l.startswith("global_load_lds_dwordx4"))
    assert sum(l.startswith("global_load_lds_dwordx4") for l in tile) == 12
    run, last_mfma = 0, max(i for i, l in enumerate(tile) if l.startswith("v_mfma"))
    for l in tile[:last_mfma]:
        if l.startswith("ds_read_b128"):
            run += 1
            assert run <= reads_per_gap, "too many fragment reads in one MFMA gap"
        elif l.startswith("v_mfma"):
            run = 0


def test_product_block_keeps_the_arithmetic_contract():
    """16x16x32 form: per K tile each of the wave's 64 accumulators (8 x 8 tiles of 16 x 16, pinned AGPR quadruples) receives exactly
    three instructions in the order of the C++ loop's mfma_row -- A combination m with B combination (0, 1, 1)[m] -- 192 per tile;
    every accumulator is at least 8 instructions away from its previous use; M0 is written at least one
    instruction before the request that reads it and not overwritten in between."""
    _, lines, G = gen.block_text("BX_KLOOP_ASM")
    body, tiles = _tiles(lines, f"v_add_u32 {G.v_ac(0)},", 3)
    a_combo = {G.va(s_, c): c for s_ in (0, 1) for c in range(3)}
    b_combo = {G.vb(ct, d): (ct, d) for ct in range(8) for d in range(2)}
    accs = {G.acc(rt, ct): (rt, ct) for rt in range(8) for ct in range(8)}
    assert len(accs) == 64 and sorted(int(a[2:].split(":")[0]) for a in accs) == list(range(0, 256, 4))
    for tile in tiles:
        mfma = [l for l in tile if l.startswith("v_mfma")]
        assert len(mfma) == 192 and all(l.startswith("v_mfma_f32_16x16x32_bf16 ") for l in mfma)
        per_acc, last_use = {}, {}
        for k, l in enumerate(mfma):
            acc, a, b, c = [x.strip() for x in l.split(" ", 1)[1].split(", ")]
            assert acc == c
            rt, ct = accs[acc]
            ctb, d = b_combo[b]
            assert ctb == ct
            per_acc.setdefault(acc, []).append((a_combo[a], d))
            assert k - last_use.get(acc, -100) >= 8
            last_use[acc] = k
        assert len(per_acc) == 64 and all(v == [(0, 0), (1, 1), (2, 1)] for v in per_acc.values())
        _common_tile_checks(tile, 1)
    m0_at = None
    for i, l in enumerate(body):
        if l.startswith("s_add_u32 m0,"):
            m0_at = i
        if l.startswith("global_load_lds_dwordx4"):
            assert m0_at is not None and i - m0_at >= 2, "LDS-DMA needs one wait state behind the write of M0"
            assert re.fullmatch(r"global_load_lds_dwordx4 v\d+, s\[\d+:\d+\]", l)
            m0_at = None
    # the operand list pins acc[i][j][q] to the registers the text names
    ops = gen.acc_operands16().split(", ")
    assert len(ops) == 64 and ops[0] == '"+{a[0:3]}"(ACC[0][0][0])' and ops[-1] == '"+{a[252:255]}"(ACC[3][3][3])'
    for i in range(4):
        for j in range(4):
            for q in range(4):
                assert G.acc(2 * i + (q >> 1), 2 * j + (q & 1)) == f"a[{16 * (4 * i + j) + 4 * q}:{16 * (4 * i + j) + 4 * q + 3}]"


def test_32x32x16_block_keeps_the_arithmetic_contract():
    """-DBX_SHAPE16=0 form: per K tile every accumulator tile receives exactly the six partial products of mfma_row<6>, in that order
    (lo hi, hi lo, mid mid, mid hi, hi mid, hi hi); 96 MFMAs per tile; never more than two fragment reads between two MFMAs."""
    _, lines, G = gen.block_text("BX_KLOOP_ASM32")
    body, tiles = _tiles(lines, f"v_add_u32 {G.v_aa},", 3)
    a_piece = {G.va(s_, pc): pc for s_ in (0, 1) for pc in range(3)}
    b_piece = {G.vb(j, pc): (j, pc) for j in range(G.nj) for pc in range(3)}
    for tile in tiles:
        mfma = [l for l in tile if l.startswith("v_mfma_f32_32x32x16_bf16")]
        assert len(mfma) == 96
        per_acc = {}
        for l in mfma:
            acc, a, b, c = [x.strip() for x in l.split(" ", 1)[1].split(", ")]
            assert acc == c
            j, pb = b_piece[b]
            per_acc.setdefault(acc, []).append((a_piece[a], pb))
        assert len(per_acc) == 16 and all(v == gen.PRODUCTS for v in per_acc.values())
        _common_tile_checks(tile, 2)
    for i, l in enumerate(body):
        if l.startswith("global_load_lds_dwordx4"):
            assert body[i - 1] == "s_nop 0" and body[i - 2].startswith("s_add_u32 m0,")
            assert re.fullmatch(r"global_load_lds_dwordx4 v\d+, s\[\d+:\d+\]", l)
