"""The one generated source of the library -- csrc/bx_kloop_asm.inc, the hand-scheduled K loop of gemm256_bx_kernel<6, true> -- must be
exactly what scripts/gen_bx_kloop.py writes (nobody edits the block by hand; a generator change without regeneration is caught here),
and the block must keep the properties its arithmetic contract rests on.  Runs without a GPU."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))

import gen_bx_kloop as gen  # noqa: E402


def _product_block():
    kw = gen.variants()["BX_KLOOP_ASM"]
    lines, G = gen.gen(**kw)
    return kw, lines, G


def test_committed_block_is_the_generators_output():
    kw, lines, G = _product_block()
    want = gen.HEADER + gen.render(lines, G, "BX_KLOOP_ASM", saddr=kw.get("addr") == "saddr", unroll=kw.get("unroll", 1))
    have = open(os.path.join(ROOT, "vivit_amd", "csrc", "bx_kloop_asm.inc")).read()
    assert have == want, "vivit_amd/csrc/bx_kloop_asm.inc is stale: run `python scripts/gen_bx_kloop.py`"


def test_block_keeps_the_arithmetic_contract():
    """Per K tile every accumulator tile receives exactly the six partial products of mfma_row<6>, in that order (lo hi, hi lo,
    mid mid, mid hi, hi mid, hi hi); 96 MFMAs per tile; every request sits behind the mid-tile barrier; never more than two
    fragment reads between two MFMAs (except behind the tile's last MFMA); one barrier and one vmcnt(0) per tile."""
    kw, lines, G = _product_block()
    body = lines[lines.index("BXK_G0_%=:") + 1:]
    tiles, cur = [], []
    for l in body:           # split the unrolled trip into tiles at the per-tile address setup
        if l.startswith(f"v_add_u32 {G.v_aa},") and cur:
            tiles.append(cur)
            cur = []
        cur.append(l)
    tiles.append(cur)
    assert len(tiles) == kw["unroll"]
    a_piece = {}             # register -> (set, piece)
    for s_ in (0, 1):
        for pc in range(3):
            a_piece[G.va(s_, pc)] = pc
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
        assert sum(l == "s_barrier" for l in tile) == 1 and sum(l == "s_waitcnt vmcnt(0)" for l in tile) == 1
        bar = tile.index("s_barrier")
        assert all(i > bar for i, l in enumerate(tile) if l.startswith("global_load_lds_dwordx4"))
        assert sum(l.startswith("global_load_lds_dwordx4") for l in tile) == 12
        run, last_mfma = 0, max(i for i, l in enumerate(tile) if l.startswith("v_mfma"))
        for i, l in enumerate(tile[:last_mfma]):
            if l.startswith("ds_read_b128"):
                run += 1
                assert run <= 2, "more than two fragment reads in one MFMA gap"
            elif l.startswith("v_mfma"):
                run = 0
    # the M0 hazard pad in front of every request, scalar request bases
    for i, l in enumerate(body):
        if l.startswith("global_load_lds_dwordx4"):
            assert body[i - 1] == "s_nop 0" and body[i - 2].startswith("s_add_u32 m0,")
            assert re.fullmatch(r"global_load_lds_dwordx4 v\d+, s\[\d+:\d+\]", l)
