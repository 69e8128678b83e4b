#!/usr/bin/env python3
"""Generator of the hand-scheduled K loop of gemm256_bx_kernel<6> (vivit_amd/csrc/gemm_f32.hip) as ONE inline-asm block.

    python scripts/gen_bx_kloop.py            # writes vivit_amd/csrc/bx_kloop_asm.inc (committed: the build never runs this)

VERDICT r03-r05 asked for the loop body in assembly: fixed register map, the twelve global -> LDS requests of a K tile
placed by hand between the MFMAs.  The block runs `ntiles` K tiles (16 k each) of the pipeline the C++ loop runs
(gemm_f32.hip, "Pipeline (tile t lives in stage t % 3)") and is entered / left in that loop's invariant:

    entry: tile t has landed in stage st and is published (every wave is past the barrier that followed its wait), the
           requests of tile t + 1 are in flight, nothing of tile t is in registers;
    exit : the same for t + ntiles; the 16 accumulator tiles are updated (st and the request pointers are inputs: the caller
           advances its copies by ntiles).

Arithmetic contract (bit-identical to the C++ loop): every accumulator tile (i, j) sees, per K tile, the six partial
products in the order (lo hi), (hi lo), (mid mid), (mid hi), (hi mid), (hi hi) of (A piece, B piece) -- mfma_row<6> --
and the K tiles in ascending order.  Only the order BETWEEN accumulators and the placement of memory instructions differ.

One K tile per trip (the C++ loop needs two: its fragment sets ping-pong by name).  Per wave and K tile:
    96 v_mfma_f32_32x32x16_bf16, 27 ds_read_b128 (12 B fragments, 12 A fragments, 3 of the next tile's first row),
    12 global_load_lds_dwordx4 + 12 v_lshl_add_u64 (request pointers), one s_waitcnt vmcnt(0) + s_barrier in the middle.
Rows 0-2 walk the four column tiles round-robin per product; row 3 walks column by column so that the B fragments of column
j are dead behind its six MFMAs and the NEXT tile's B fragments of column j can take their registers.  The A fragments of
the rows alternate between two register sets (row 0, 2: X; row 1, 3: Y; the next tile's row 0: X).

Register map (VGPRs named as clobbers, so hipcc keeps its own values out of them):
    B fragments  v[128:175]   B(j, pc) = v[128 + 4 (3 j + pc) ..+3]
    A set X      v[176:187]   A(pc)    = v[176 + 4 pc ..+3]        A set Y  v[188:199]
    request pointers (64 bit, one per (operand, block u, piece pc)): v[200:223]
    v224 / v225 / v226: LDS addresses of this lane's A fragments (tile t), B fragments (tile t + 1), A fragments (tile t + 1)
    s[84:95]: stage offsets, request base, loop counter
"""
import os
import sys

BX_PIECE = 8 * 1024
BX_OPER = 3 * BX_PIECE
BX_STAGE = 2 * BX_OPER

PRODUCTS = [(2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)]   # (A piece, B piece): mfma_row<6>, smallest partial products first

# operand numbers of the asm statement (see the wrapper this script writes)
ACC = lambda i, j: f"%{4 * i + j}"
FOFS_A, FOFS_B = "%16", "%17"
SRC = {("A", 0): "%18", ("A", 1): "%19", ("B", 0): "%20", ("B", 1): "%21"}
STRIDE = {"A": "%22", "B": "%23"}
STEP = {"A": "%24", "B": "%25"}
LDS0, ST, NT = "%26", "%27", "%28"

VB = lambda j, pc: f"v[{128 + 4 * (3 * j + pc)}:{128 + 4 * (3 * j + pc) + 3}]"
VA = lambda s, pc: f"v[{(176 if s == 0 else 188) + 4 * pc}:{(176 if s == 0 else 188) + 4 * pc + 3}]"


def pair(op, u, pc):
    k = 200 + 2 * ((0 if op == "A" else 6) + 3 * u + pc)
    return f"v[{k}:{k + 1}]"


V_AA, V_BN, V_AN = "v224", "v225", "v226"
S_CUR, S_NXT, S_NN, S_DMA, S_CNT, S_TMP = "s84", "s85", "s86", "s87", "s88", "s89"
CLOBBER_V = list(range(128, 227))
CLOBBER_S = list(range(84, 90))


class Emitter:
    def __init__(self):
        self.lines = []
        self.lgkm = []      # names of outstanding ds_reads, oldest first (the LDS returns data in order)

    def raw(self, s):
        self.lines.append(s)

    def ds_read(self, name, dst, addr, offset):
        self.raw(f"ds_read_b128 {dst}, {addr} offset:{offset}")
        self.lgkm.append(name)

    def need(self, names):
        """s_waitcnt lgkmcnt(N) so that every fragment in `names` has arrived (N = reads issued behind the youngest needed)."""
        idx = [self.lgkm.index(n) for n in names if n in self.lgkm]
        if not idx:
            return
        keep = min(len(self.lgkm) - 1 - max(idx), 15)   # (4-bit counter: waiting for more than needed is always safe)
        self.raw(f"s_waitcnt lgkmcnt({keep})")
        del self.lgkm[: max(idx) + 1]


def requests():
    """The 12 requests of one K tile in the C++ order: part q = (block u = q / 2, operand q % 2), three pieces each."""
    out = []
    for q in range(4):
        u, op = q >> 1, "AB"[q & 1]
        for pc in range(3):
            out.append((op, u, pc))
    return out


def gen(req_gaps=None, setprio=False, nop_m0=True):
    """`req_gaps`: for each of the 12 requests the MFMA slot (0..96) BEFORE which it is issued (>= 48: behind the barrier)."""
    if req_gaps is None:
        # the C++ placement: three per column tile of row 3 (behind the fragment reads of the column) -- slots 78, 84, 90, 96
        req_gaps = [78] * 3 + [84] * 3 + [90] * 3 + [96] * 3
    reqs = requests()
    e = Emitter()
    e.raw("; ---- entry: stage offsets of tiles t, t + 1, t + 2 from st; request pointers of the 12 (operand, block, piece) streams")
    e.raw(f"s_mul_i32 {S_CUR}, {ST}, {BX_STAGE}")
    e.raw(f"s_add_u32 {S_NXT}, {S_CUR}, {BX_STAGE}")
    e.raw(f"s_cmp_eq_u32 {ST}, 2")
    e.raw(f"s_cselect_b32 {S_NXT}, 0, {S_NXT}")
    e.raw(f"s_add_u32 {S_NN}, {S_NXT}, {BX_STAGE}")
    e.raw(f"s_cmp_eq_u32 {S_NXT}, {2 * BX_STAGE}")
    e.raw(f"s_cselect_b32 {S_NN}, 0, {S_NN}")
    e.raw(f"s_mov_b32 {S_CNT}, {NT}")
    for op in "AB":
        for u in range(2):
            e.raw(f"v_lshl_add_u64 {pair(op, u, 0)}, {SRC[(op, u)]}, 0, 0")
            e.raw(f"v_lshl_add_u64 {pair(op, u, 1)}, {STRIDE[op]}, 0, {SRC[(op, u)]}")
            e.raw(f"v_lshl_add_u64 {pair(op, u, 2)}, {STRIDE[op]}, 1, {SRC[(op, u)]}")
    # fragments of tile t: all of B, row 0 of A
    e.raw(f"v_add_u32 {V_AA}, {S_CUR}, {FOFS_A}")
    e.raw(f"v_add_u32 {V_BN}, {S_CUR}, {FOFS_B}")
    # (row 0 of A first, then B column by column: the order in which a trip requests the NEXT tile's fragments, so that one
    # model of the LDS return queue is valid at the loop head for the first trip and for every later one)
    for pc in range(3):
        e.ds_read(f"A0{pc}", VA(0, pc), V_AA, pc * BX_PIECE)
    for j in range(4):
        for pc in range(3):
            e.ds_read(f"B{j}{pc}", VB(j, pc), V_BN, pc * BX_PIECE + j * 1024)
    e.raw("BXK_LOOP_%=:")
    e.raw(f"v_add_u32 {V_AA}, {S_CUR}, {FOFS_A}")
    e.raw(f"v_add_u32 {V_BN}, {S_NXT}, {FOFS_B}")
    e.raw(f"v_add_u32 {V_AN}, {S_NXT}, {FOFS_A}")
    e.raw(f"s_add_u32 {S_DMA}, {LDS0}, {S_NN}")

    # ---- the MFMA stream: (slot, acc, A fragment name / register, B fragment name / register)
    stream = []
    for i in range(3):
        for pa, pb in PRODUCTS:
            for j in range(4):
                stream.append((i, j, pa, pb))
    for j in range(4):
        for pa, pb in PRODUCTS:
            stream.append((3, j, pa, pb))
    assert len(stream) == 96

    fillers = {g: [] for g in range(97)}
    # A fragments of rows 1..3 (current stage) and of the next tile's row 0 (next stage: behind the barrier)
    fillers[0] += [("ds", f"A1{pc}", VA(1, pc), V_AA, pc * BX_PIECE + 1 * 1024) for pc in range(3)]
    fillers[24] += [("ds", f"A2{pc}", VA(0, pc), V_AA, pc * BX_PIECE + 2 * 1024) for pc in range(3)]
    fillers[48] += [("ds", f"A3{pc}", VA(1, pc), V_AA, pc * BX_PIECE + 3 * 1024) for pc in range(3)]
    fillers[48] += [("barrier",)]
    fillers[72] += [("ds", f"N0{pc}", VA(0, pc), V_AN, pc * BX_PIECE) for pc in range(3)]
    for j in range(4):   # next tile's B fragments of column j behind the six MFMAs of column j of row 3
        fillers[72 + 6 * (j + 1)] += [("ds", f"NB{j}{pc}", VB(j, pc), V_BN, pc * BX_PIECE + j * 1024) for pc in range(3)]
    for r, g in enumerate(req_gaps):
        assert 48 <= g <= 96, "requests overwrite the stage of tile t - 1: only behind the mid-tile barrier"
        fillers[g].append(("req", r))

    def emit_fillers(g):
        for f in fillers[g]:
            if f[0] == "ds":
                e.ds_read(f[1], f[2], f[3], f[4])
            elif f[0] == "barrier":
                e.raw("s_waitcnt vmcnt(0)")      # this wave's share of tile t + 1 has landed
                e.raw("s_barrier")               # publishes tile t + 1; every wave is past tile t - 1
            elif f[0] == "req":
                op, u, pc = reqs[f[1]]
                ofs = (BX_OPER if op == "B" else 0) + pc * BX_PIECE + 4 * u * 1024
                e.raw(f"s_add_u32 m0, {S_DMA}, {ofs}")
                if nop_m0:
                    e.raw("s_nop 0")             # SALU write of M0 -> LDS-DMA read of M0: one wait state (gfx9 hazard table)
                e.raw(f"global_load_lds_dwordx4 {pair(op, u, pc)}, off")
                e.raw(f"v_lshl_add_u64 {pair(op, u, pc)}, {pair(op, u, pc)}, 0, {STEP[op]}")

    names_a = {0: "A0", 1: "A1", 2: "A2", 3: "A3"}
    # at the loop head the fragments of this tile were requested in the previous trip (or at entry): model them as pending
    e.lgkm = [f"A0{pc}" for pc in range(3)] + [f"B{j}{pc}" for j in range(4) for pc in range(3)]
    for g, (i, j, pa, pb) in enumerate(stream):
        emit_fillers(g)
        e.need([f"{names_a[i]}{pa}", f"B{j}{pb}"])
        if setprio and g in (0, 48):
            e.raw("s_setprio 1")
        seta = 0 if i in (0, 2) else 1
        e.raw(f"v_mfma_f32_32x32x16_bf16 {ACC(i, j)}, {VA(seta, pa)}, {VB(j, pb)}, {ACC(i, j)}")
    emit_fillers(96)
    # the next trip finds: B fragments (12) then A row 0 (3) pending, in THAT order for its lgkm model -> re-order check
    pend = list(e.lgkm)
    want = [f"N0{pc}" for pc in range(3)] + [f"NB{j}{pc}" for j in range(4) for pc in range(3)]
    assert [p for p in pend if p.startswith("N")] == want, pend
    # rotate the stages: (cur, nxt, nn) <- (nxt, nn, cur)
    e.raw(f"s_mov_b32 {S_TMP}, {S_CUR}")
    e.raw(f"s_mov_b32 {S_CUR}, {S_NXT}")
    e.raw(f"s_mov_b32 {S_NXT}, {S_NN}")
    e.raw(f"s_mov_b32 {S_NN}, {S_TMP}")
    e.raw(f"s_sub_u32 {S_CNT}, {S_CNT}, 1")
    e.raw(f"s_cmp_lg_u32 {S_CNT}, 0")
    e.raw("s_cbranch_scc1 BXK_LOOP_%=")
    e.raw("; ---- exit: nothing of the next tile is needed in registers (the C++ side re-reads its fragments)")
    e.raw("s_waitcnt lgkmcnt(0)")
    # (st and the request pointers are plain inputs: the C++ side advances its own copies by ntiles)
    return e.lines


def render(lines, name="BX_KLOOP_ASM"):
    body = "\n".join(f'  "{l}\\n\\t"' for l in lines)
    clob = ", ".join([f'"v{k}"' for k in CLOBBER_V] + [f'"s{k}"' for k in CLOBBER_S] + [ '"scc"', '"memory"'])
    return f"#define {name}_TEXT \\\n" + " \\\n".join(f'  "{l}\\n\\t"' for l in lines) + f"\n#define {name}_CLOBBERS {clob}\n"


HEADER = """// GENERATED by scripts/gen_bx_kloop.py -- do not edit; see that script for the register map, the pipeline invariant at
// entry / exit and the arithmetic contract (bit-identical to the C++ K loop of gemm256_bx_kernel<6>).
"""


def main():
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "vivit_amd", "csrc", "bx_kloop_asm.inc")
    text = HEADER + render(gen())
    # experiment variants (selected with -DBX_ASM_VARIANT=n; timing A/Bs, same arithmetic)
    variants = {
        1: dict(req_gaps=[54, 57, 60, 63, 66, 69, 75, 78, 81, 84, 87, 90]),          # one request per three MFMAs over rows 2 + 3
        2: dict(req_gaps=[76, 78, 80, 82, 84, 86, 88, 90, 92, 94, 96, 96]),          # one per two MFMAs in row 3
        3: dict(req_gaps=[78] * 3 + [84] * 3 + [90] * 3 + [96] * 3, setprio=True),
        4: dict(req_gaps=[78] * 3 + [84] * 3 + [90] * 3 + [96] * 3, nop_m0=False),
    }
    for k, kw in variants.items():
        text += render(gen(**kw), name=f"BX_KLOOP_ASM_V{k}")
    with open(out, "w") as f:
        f.write(text)
    print("wrote", os.path.normpath(out), len(gen()), "instructions per block")


if __name__ == "__main__":
    main()
