#!/usr/bin/env python3
"""Generator of the hand-scheduled K loops of gemm256_bx_kernel<6, true> (vivit_amd/csrc/gemm_f32.hip) as inline-asm blocks.

    python scripts/gen_bx_kloop.py            # writes vivit_amd/csrc/bx_kloop_asm.inc (committed: the build never runs this)
    python scripts/gen_bx_kloop.py --variants # + csrc/bx_kloop_asm_variants.inc: the experiment / attribution blocks (git-ignored)

Two forms: BX_KLOOP_ASM on v_mfma_f32_16x16x32_bf16 (the product; Geometry16 / gen16 below, with its own contract) and BX_KLOOP_ASM32
on v_mfma_f32_32x32x16_bf16 (-DBX_SHAPE16=0; Geometry / gen, described first).

VERDICT r03-r05 asked for the loop body in assembly: fixed register map, the global -> LDS requests of a K tile placed by
hand between the MFMAs.  A block runs `ntiles` K tiles (16 k each) of the pipeline the C++ loop runs (gemm_f32.hip,
"Pipeline (tile t lives in stage t % 3)") and is entered / left in that loop's invariant:

    entry: tile t has landed in stage st and is published (every wave is past the barrier that followed its wait), the
           requests of tile t + 1 are in flight, nothing of tile t is in registers;
    exit : the same for t + ntiles; the accumulator tiles are updated (st and the request pointers are inputs: the caller
           advances its copies by ntiles).

Arithmetic contract (bit-identical to the C++ loop): every accumulator tile (i, j) sees, per K tile, the six partial
products in the order (lo hi), (hi lo), (mid mid), (mid hi), (hi mid), (hi hi) of (A piece, B piece) -- mfma_row<6> --
and the K tiles in ascending order.  Only the order BETWEEN accumulators and the placement of memory instructions differ.

Two geometries of the 256 x 256 x 16 workgroup tile:
    NW = 4  four waves of 128 x 128 (4 x 4 accumulator tiles of 32 x 32, one wave per SIMD, 512 registers): per wave and K tile
            96 MFMAs, 27 ds_read_b128, 12 global_load_lds_dwordx4;
    NW = 8  eight waves of 128 x 64 (4 x 2 accumulator tiles, TWO waves per SIMD, 256 registers): per wave and K tile 48 MFMAs,
            21 ds_read_b128, 6 requests.  The point of the second wave: a global -> LDS request holds its wave's issue for
            ~60 cycles (MI355X_MICROARCH.md, LDS-DMA piece issue cost) -- with one wave per SIMD the matrix pipe idles for
            what exceeds an MFMA's 24 free issue cycles (12 x 36 cycles per K tile = 12 % of it); with two the partner's MFMAs
            run underneath.  The two wave groups (waves 0-3 / 4-7: one of each per SIMD) issue their requests in DIFFERENT rows.
One K tile per trip.  Rows 0-2 walk the column tiles round-robin per product; row 3 walks column by column so that the B
fragments of column j are dead behind its six MFMAs and the NEXT tile's B fragments of column j take their registers.  The A
fragments of the rows alternate between two register sets (rows 0, 2: X; rows 1, 3: Y; the next tile's row 0: X).

Register map (VGPRs / SGPRs named as clobbers, so hipcc keeps its own values out of them); base V0 = 128 (NW = 4) / 64 (NW = 8):
    B fragments  v[V0 ..]           B(j, pc): 4 registers each, 3 NJ fragments
    A sets X, Y  12 registers each
    request pointers (64 bit, one per (operand, block u, piece pc))
    three LDS addresses: this lane's A fragments (tile t), B fragments (tile t + 1), A fragments (tile t + 1)
    s[84:89]: stage offsets (tiles t, t + 1, t + 2), request base, loop counter, scratch
"""
import os

BX_PIECE = 8 * 1024
BX_OPER = 3 * BX_PIECE
BX_STAGE = 2 * BX_OPER

PRODUCTS = [(2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)]   # (A piece, B piece): mfma_row<6>, smallest partial products first
S_CUR, S_NXT, S_NN, S_DMA, S_CNT, S_TMP = "s84", "s85", "s86", "s87", "s88", "s89"
CLOBBER_S = list(range(84, 90))
# addr = "saddr": the twelve request streams as SGPR base pairs s[40:63] + ONE per-lane 32-bit offset (global_load_lds v, s[..]);
# steps in s[64:67]
SADDR_BASE, SADDR_STEP = 40, 64
CLOBBER_S_SADDR = list(range(40, 68))


class Geometry:
    def __init__(self, nw):
        assert nw in (4, 8)
        self.nw = nw
        self.nj = 4 if nw == 4 else 2          # column tiles per wave
        self.nu = 8 // nw                      # 1 KB blocks per wave, piece and operand
        self.nacc = 4 * self.nj
        self.nmfma = 6 * self.nacc
        self.row = 6 * self.nj                 # MFMAs per row of accumulator tiles
        v0 = 128 if nw == 4 else 64
        self.vb0 = v0
        self.vax = v0 + 12 * self.nj
        self.vay = self.vax + 12
        self.vp0 = self.vay + 12
        self.npair = 2 * self.nu * 3
        self.vaddr = self.vp0 + 2 * self.npair
        self.clobber_v = list(range(v0, self.vaddr + 4))
        # operand numbers of the asm statement (the C++ side passes them in this order)
        k = self.nacc
        self.fofs_a, self.fofs_b = f"%{k}", f"%{k + 1}"
        k += 2
        self.src = {}
        for op in "AB":
            for u in range(self.nu):
                self.src[(op, u)] = f"%{k}"
                k += 1
        self.stride = {"A": f"%{k}", "B": f"%{k + 1}"}
        self.step = {"A": f"%{k + 2}", "B": f"%{k + 3}"}
        self.lds0, self.st, self.nt, self.grp = f"%{k + 4}", f"%{k + 5}", f"%{k + 6}", f"%{k + 7}"
        self.noperands = k + 8

    def acc(self, i, j):
        return f"%{self.nj * i + j}"

    def vb(self, j, pc):
        k = self.vb0 + 4 * (3 * j + pc)
        return f"v[{k}:{k + 3}]"

    def va(self, s, pc):
        k = (self.vax if s == 0 else self.vay) + 4 * pc
        return f"v[{k}:{k + 3}]"

    def pair(self, op, u, pc):
        k = self.vp0 + 2 * ((0 if op == "A" else 3 * self.nu) + 3 * u + pc)
        return f"v[{k}:{k + 1}]"

    @property
    def v_aa(self):
        return f"v{self.vaddr}"

    @property
    def v_bn(self):
        return f"v{self.vaddr + 1}"

    @property
    def v_an(self):
        return f"v{self.vaddr + 2}"

    @property
    def v_off(self):
        return f"v{self.vaddr + 3}"

    def sbase(self, op, u, pc):
        k = SADDR_BASE + 2 * ((0 if op == "A" else 3 * self.nu) + 3 * u + pc)
        return k

    def requests(self):
        """The requests of one K tile in the C++ order: part q = (block u = q / 2, operand q % 2), three pieces each."""
        out = []
        for q in range(2 * self.nu):
            u, op = q >> 1, "AB"[q & 1]
            for pc in range(3):
                out.append((op, u, pc))
        return out


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


def loop_body(e, G, req_gaps, label, setprio=False, nop_m0=True, ds_per_gap=3, row0_colmajor=False, barrier="full", dma=True, addr="vaddr",
              unroll=1, mfma16=False):
    """`unroll` tiles per trip: the stage registers take their roles by renaming (no rotation moves), ntiles % unroll == 0."""
    e.raw(f"{label}:")
    regs = [S_CUR, S_NXT, S_NN]
    for k in range(unroll):
        one_tile(e, G, req_gaps, regs[k % 3], regs[(k + 1) % 3], regs[(k + 2) % 3], setprio, nop_m0, ds_per_gap, row0_colmajor, barrier, dma, addr, mfma16)
    if unroll % 3 != 0:   # rotate the stages: (cur, nxt, nn) <- (nxt, nn, cur), `unroll` times
        e.raw(f"s_mov_b32 {S_TMP}, {regs[0]}")
        if unroll % 3 == 1:
            e.raw(f"s_mov_b32 {regs[0]}, {regs[1]}")
            e.raw(f"s_mov_b32 {regs[1]}, {regs[2]}")
            e.raw(f"s_mov_b32 {regs[2]}, {S_TMP}")
        else:
            e.raw(f"s_mov_b32 {regs[0]}, {regs[2]}")
            e.raw(f"s_mov_b32 {regs[2]}, {regs[1]}")
            e.raw(f"s_mov_b32 {regs[1]}, {S_TMP}")
    e.raw(f"s_sub_u32 {S_CNT}, {S_CNT}, {unroll}")
    e.raw(f"s_cmp_lg_u32 {S_CNT}, 0")
    e.raw(f"s_cbranch_scc1 {label}")


def one_tile(e, G, req_gaps, S_CUR, S_NXT, S_NN, setprio, nop_m0, ds_per_gap, row0_colmajor, barrier, dma, addr, mfma16=False):
    """One K tile: the MFMA stream with its fillers; `req_gaps[r]` = the MFMA slot BEFORE which request r is issued.
    `ds_per_gap`: at most this many ds_read_b128 per MFMA gap (MI355X_MICROARCH.md, LDS: a third read per gap by every wave
    saturates the LDS array and stretches the gap to 48 cycles); `row0_colmajor`: row 0 column by column like row 3, so that the
    B fragments of the last column, read at the very end of the previous tile, are first needed 18 MFMAs later instead of 3.
    `barrier` = "none" / `dma` = False: TIMING-ONLY builds (wrong results) that take one ingredient out of the loop."""
    R = G.row
    reqs = G.requests()
    assert len(req_gaps) == len(reqs)
    e.raw(f"v_add_u32 {G.v_aa}, {S_CUR}, {G.fofs_a}")
    e.raw(f"v_add_u32 {G.v_bn}, {S_NXT}, {G.fofs_b}")
    e.raw(f"v_add_u32 {G.v_an}, {S_NXT}, {G.fofs_a}")
    e.raw(f"s_add_u32 {S_DMA}, {G.lds0}, {S_NN}")
    stream = []
    for i in range(3):
        if i == 0 and row0_colmajor:
            for j in range(G.nj):
                for pa, pb in PRODUCTS:
                    stream.append((i, j, pa, pb))
            continue
        for pa, pb in PRODUCTS:
            for j in range(G.nj):
                stream.append((i, j, pa, pb))
    for j in range(G.nj):
        for pa, pb in PRODUCTS:
            stream.append((3, j, pa, pb))
    assert len(stream) == G.nmfma
    fillers = {g: [] for g in range(G.nmfma + 1)}
    # A fragments of rows 1..3 (current stage) and of the next tile's row 0 (next stage: behind the barrier)
    fillers[0] += [("ds", f"A1{pc}", G.va(1, pc), G.v_aa, pc * BX_PIECE + 1 * 1024) for pc in range(3)]
    fillers[R] += [("ds", f"A2{pc}", G.va(0, pc), G.v_aa, pc * BX_PIECE + 2 * 1024) for pc in range(3)]
    fillers[2 * R] += [("ds", f"A3{pc}", G.va(1, pc), G.v_aa, pc * BX_PIECE + 3 * 1024) for pc in range(3)]
    fillers[2 * R] += [("barrier",)]
    fillers[3 * R] += [("ds", f"N0{pc}", G.va(0, pc), G.v_an, pc * BX_PIECE) for pc in range(3)]
    for j in range(G.nj):   # next tile's B fragments of column j behind the six MFMAs of column j of row 3
        fillers[3 * R + 6 * (j + 1)] += [("ds", f"NB{j}{pc}", G.vb(j, pc), G.v_bn, pc * BX_PIECE + j * 1024) for pc in range(3)]
    for r, g in enumerate(req_gaps):
        assert 2 * R <= g <= G.nmfma, "requests overwrite the stage of tile t - 1: only behind the mid-tile barrier"
        if dma:
            fillers[g].append(("req", r))
    # at most ds_per_gap fragment reads per gap: the overflow moves to the following gaps (never across the end of the trip)
    for g in range(G.nmfma + 1):
        reads = [f for f in fillers[g] if f[0] == "ds"]
        if len(reads) > ds_per_gap and g < G.nmfma:
            move = reads[ds_per_gap:]
            fillers[g] = [f for f in fillers[g] if f not in move]
            fillers[g + 1] = move + fillers[g + 1]

    def emit_fillers(g):
        for f in fillers[g]:
            if f[0] == "ds":
                e.ds_read(f[1], f[2], f[3], f[4])
            elif f[0] == "barrier":
                if dma:
                    e.raw("s_waitcnt vmcnt(0)")      # this wave's share of tile t + 1 has landed
                if barrier == "full":
                    e.raw("s_barrier")               # publishes tile t + 1; every wave is past tile t - 1
            elif f[0] == "req":
                op, u, pc = reqs[f[1]]
                ofs = (BX_OPER if op == "B" else 0) + pc * BX_PIECE + G.nw * u * 1024
                e.raw(f"s_add_u32 m0, {S_DMA}, {ofs}")
                if nop_m0:
                    e.raw("s_nop 0")             # SALU write of M0 -> LDS-DMA read of M0: one wait state (gfx9 hazard table)
                if addr == "saddr":
                    b = G.sbase(op, u, pc)
                    st_ = SADDR_STEP + (0 if op == "A" else 2)
                    e.raw(f"global_load_lds_dwordx4 {G.v_off}, s[{b}:{b + 1}]")
                    e.raw(f"s_add_u32 s{b}, s{b}, s{st_}")
                    e.raw(f"s_addc_u32 s{b + 1}, s{b + 1}, s{st_ + 1}")
                else:
                    e.raw(f"global_load_lds_dwordx4 {G.pair(op, u, pc)}, off")
                    e.raw(f"v_lshl_add_u64 {G.pair(op, u, pc)}, {G.pair(op, u, pc)}, 0, {G.step[op]}")

    names_a = {0: "A0", 1: "A1", 2: "A2", 3: "A3"}
    # at the loop head the fragments of this tile were requested by the previous trip (or at entry), in this order
    e.lgkm = [f"A0{pc}" for pc in range(3)] + [f"B{j}{pc}" for j in range(G.nj) for pc in range(3)]
    for g, (i, j, pa, pb) in enumerate(stream):
        emit_fillers(g)
        e.need([f"{names_a[i]}{pa}", f"B{j}{pb}"])
        if setprio and g in (0, 2 * R):
            e.raw("s_setprio 1")
        if mfma16:
            # TIMING ONLY (wrong results): the same flops as two v_mfma_f32_16x16x32_bf16 on the same operand registers, accumulators
            # taken as fixed 4-register groups of the 256 AGPRs -- what the other MFMA shape does to cycles and clock
            # (MI355X_MICROARCH.md, DVFS give-back item 7: 1.12-1.14 x the FLOP/s of the 32x32x16 shape in power-limited loops)
            for half in range(2):
                a4 = 4 * ((2 * g + half) % 64)
                e.raw(f"v_mfma_f32_16x16x32_bf16 a[{a4}:{a4 + 3}], {G.va(0 if i in (0, 2) else 1, pa)}, {G.vb(j, pb)}, a[{a4}:{a4 + 3}]")
            continue
        e.raw(f"v_mfma_f32_32x32x16_bf16 {G.acc(i, j)}, {G.va(0 if i in (0, 2) else 1, pa)}, {G.vb(j, pb)}, {G.acc(i, j)}")
    emit_fillers(G.nmfma)
    want = [f"N0{pc}" for pc in range(3)] + [f"NB{j}{pc}" for j in range(G.nj) for pc in range(3)]
    assert [p for p in e.lgkm if p.startswith("N")] == want, e.lgkm


def gen(nw=4, req_gaps=None, req_gaps_g1=None, setprio=False, nop_m0=True, **body_kw):
    """`req_gaps`: placement of the requests (wave group 0); `req_gaps_g1`: of wave group 1 (NW = 8: waves 4-7), None = the same."""
    G = Geometry(nw)
    R = G.row
    nreq = len(G.requests())
    if req_gaps is None:
        # the C++ placement: three per column tile of row 3, behind the fragment reads of the column
        req_gaps = [3 * R + 6 * (q + 1) for q in range(nreq // 3) for _ in range(3)]
    e = Emitter()
    e.raw("; ---- entry: stage offsets of tiles t, t + 1, t + 2 from st; request pointers of the (operand, block, piece) streams")
    e.raw(f"s_mul_i32 {S_CUR}, {G.st}, {BX_STAGE}")
    e.raw(f"s_add_u32 {S_NXT}, {S_CUR}, {BX_STAGE}")
    e.raw(f"s_cmp_eq_u32 {G.st}, 2")
    e.raw(f"s_cselect_b32 {S_NXT}, 0, {S_NXT}")
    e.raw(f"s_add_u32 {S_NN}, {S_NXT}, {BX_STAGE}")
    e.raw(f"s_cmp_eq_u32 {S_NXT}, {2 * BX_STAGE}")
    e.raw(f"s_cselect_b32 {S_NN}, 0, {S_NN}")
    e.raw(f"s_mov_b32 {S_CNT}, {G.nt}")
    for op in "AB":
        for u in range(G.nu):
            e.raw(f"v_lshl_add_u64 {G.pair(op, u, 0)}, {G.src[(op, u)]}, 0, 0")
            e.raw(f"v_lshl_add_u64 {G.pair(op, u, 1)}, {G.stride[op]}, 0, {G.src[(op, u)]}")
            e.raw(f"v_lshl_add_u64 {G.pair(op, u, 2)}, {G.stride[op]}, 1, {G.src[(op, u)]}")
    if body_kw.get("addr") == "saddr":
        # wave-uniform bases of the streams (lane 0 of each pointer: its lane offset is 0) + the one per-lane offset (16 lane)
        first = G.pair("A", 0, 0)
        for op in "AB":
            for u in range(G.nu):
                for pc in range(3):
                    b, vp = G.sbase(op, u, pc), G.pair(op, u, pc)
                    lo = int(vp[2:].split(":")[0])
                    e.raw(f"v_readfirstlane_b32 s{b}, v{lo}")
                    e.raw(f"v_readfirstlane_b32 s{b + 1}, v{lo + 1}")
        lo0 = int(first[2:].split(":")[0])
        e.raw(f"v_readfirstlane_b32 {S_TMP}, v{lo0}")
        e.raw("s_nop 4")                                   # VALU-written SGPR -> read by VALU / VMEM
        e.raw(f"v_sub_u32 {G.v_off}, v{lo0}, {S_TMP}")
        e.raw(f"s_mov_b64 s[{SADDR_STEP}:{SADDR_STEP + 1}], {G.step['A']}")
        e.raw(f"s_mov_b64 s[{SADDR_STEP + 2}:{SADDR_STEP + 3}], {G.step['B']}")
    # fragments of tile t: row 0 of A first, then B column by column -- the order in which a trip requests the NEXT tile's
    # fragments, so that one model of the LDS return queue is valid at the loop head for the first trip and every later one
    e.raw(f"v_add_u32 {G.v_aa}, {S_CUR}, {G.fofs_a}")
    e.raw(f"v_add_u32 {G.v_bn}, {S_CUR}, {G.fofs_b}")
    for pc in range(3):
        e.ds_read(f"A0{pc}", G.va(0, pc), G.v_aa, pc * BX_PIECE)
    for j in range(G.nj):
        for pc in range(3):
            e.ds_read(f"B{j}{pc}", G.vb(j, pc), G.v_bn, pc * BX_PIECE + j * 1024)
    if req_gaps_g1 is not None:
        e.raw(f"s_cmp_lg_u32 {G.grp}, 0")
        e.raw("s_cbranch_scc1 BXK_G1_%=")
    loop_body(e, G, req_gaps, "BXK_G0_%=", setprio, nop_m0, **body_kw)
    if req_gaps_g1 is not None:
        e.raw("s_branch BXK_END_%=")
        loop_body(e, G, req_gaps_g1, "BXK_G1_%=", setprio, nop_m0, **body_kw)
        e.raw("BXK_END_%=:")
    e.raw("; ---- exit: nothing of the next tile is needed in registers (the C++ side re-reads its fragments)")
    e.raw("s_waitcnt lgkmcnt(0)")
    return e.lines, G


# ------------------------------------------------------------------------------------------------------------------------------
# The v_mfma_f32_16x16x32_bf16 form (the product since round 6; gemm_f32.hip "S16").  One instruction adds TWO partial products:
# its K = 32 is the tile's 16 k twice, with a different pair of pieces each time (the lanes of k blocks 0, 1 read one piece, those
# of blocks 2, 3 the other -- the per-lane piece offset is folded into the fragment address the C++ side passes in).  Per 16 x 16
# accumulator (rt, ct) and K tile three instructions in this order (smallest terms first; mfma_row of the C++ loop):
#     m = 0: A[a2 | a0] x B[b0 | b2]      m = 1: A[a1 | a1] x B[b1 | b0]      m = 2: A[a0 | a0] x B[b1 | b0]
# The wave's 128 x 128 are 8 x 8 such accumulators: a[16 (4 (rt / 2) + ct / 2) + 4 (2 (rt % 2) + ct % 2) ..+3] (acc[i][j][q] of the
# C++ side, pinned by "+{a[..]}" constraints).  Per wave and K tile: 192 MFMAs (16 cycles each), 40 ds_read_b128 (3 A fragments per
# row tile, 2 B fragments per column tile; the 16 B fragments stay in registers for the tile), 12 requests.
# Register map (base 128): B fragments 64, A sets X / Y 12 each, 8 fragment addresses (A now x 3, A next x 3, B next x 2), the lane
# offset of the requests, 24 scratch registers for the request bases at entry.
class Geometry16:
    nw, nj, nu = 4, 4, 2
    nmfma = 192
    nacc_operands = 64

    def __init__(self):
        v0 = 128
        self.vb0 = v0
        self.vax = v0 + 64
        self.vay = self.vax + 12
        self.vaddr = self.vay + 12       # 8 address registers + lane offset
        self.vp0 = self.vaddr + 10      # (64-bit pairs: even register numbers)
        self.npair = 12
        self.clobber_v = list(range(v0, self.vp0 + 2 * self.npair))
        assert self.clobber_v[-1] < 256
        k = self.nacc_operands
        self.cofs_a = [f"%{k + c}" for c in range(3)]
        self.cofs_b = [f"%{k + 3 + d}" for d in range(2)]
        k += 5
        self.src = {}
        for op in "AB":
            for u in range(self.nu):
                self.src[(op, u)] = f"%{k}"
                k += 1
        self.stride = {"A": f"%{k}", "B": f"%{k + 1}"}
        self.step = {"A": f"%{k + 2}", "B": f"%{k + 3}"}
        self.lds0, self.st, self.nt, self.grp = f"%{k + 4}", f"%{k + 5}", f"%{k + 6}", f"%{k + 7}"
        self.noperands = k + 8

    @staticmethod
    def acc(rt, ct):
        k = 16 * (4 * (rt >> 1) + (ct >> 1)) + 4 * (2 * (rt & 1) + (ct & 1))
        return f"a[{k}:{k + 3}]"

    def vb(self, ct, d):
        k = self.vb0 + 4 * (2 * ct + d)
        return f"v[{k}:{k + 3}]"

    def va(self, s, c):
        k = (self.vax if s == 0 else self.vay) + 4 * c
        return f"v[{k}:{k + 3}]"

    def v_ac(self, c):      # address of this lane's A fragments of combination c, current tile
        return f"v{self.vaddr + c}"

    def v_an(self, c):      # ... next tile
        return f"v{self.vaddr + 3 + c}"

    def v_bn(self, d):      # B fragments of combination d, next tile
        return f"v{self.vaddr + 6 + d}"

    @property
    def v_off(self):
        return f"v{self.vaddr + 8}"

    def pair(self, op, u, pc):
        k = self.vp0 + 2 * ((0 if op == "A" else 3 * self.nu) + 3 * u + pc)
        return f"v[{k}:{k + 1}]"

    sbase = Geometry.sbase
    requests = Geometry.requests

    @staticmethod
    def tile_ofs(t):        # byte offset of 16-row tile t of the wave inside a piece
        return (t >> 1) * 1024 + (t & 1) * 256


def stream16(pairs=False):
    """(rt, ct, m) of the 192 MFMAs of a K tile: row tile by row tile, product by product over the eight column tiles, so that an
    accumulator comes back every 8th instruction (walking the columns of row tiles 0 and 7 in pairs -- `pairs`, an accumulator every
    2nd instruction -- costs ~130 cycles per K tile: the dependent instruction waits for the result)."""
    out = []
    for rt in range(8):
        if pairs and rt in (0, 7):
            for p in range(4):
                for m in range(3):
                    for ct in (2 * p, 2 * p + 1):
                        out.append((rt, ct, m))
        else:
            for m in range(3):
                for ct in range(8):
                    out.append((rt, ct, m))
    return out


def next_b_order(pairs):
    """Order in which a tile reads the NEXT tile's B fragments (= the order of the reads at entry)."""
    if pairs:
        return [(ct, d) for ct in range(8) for d in range(2)]
    return [(ct, 0) for ct in range(8)] + [(ct, 1) for ct in range(8)]


def one_tile16(e, G, req_gaps, S_CUR, S_NXT, S_NN, nop_m0=True, barrier="full", dma=True, barrier_gap=96, split_req=True, pairs=False, m0_early=False, merge_waits=0, transposed=False):
    reqs = G.requests()
    assert len(req_gaps) == len(reqs)
    for c in range(3):
        e.raw(f"v_add_u32 {G.v_ac(c)}, {S_CUR}, {G.cofs_a[c]}")
    e.raw(f"s_add_u32 {S_DMA}, {G.lds0}, {S_NN}")
    stream = stream16(pairs)
    N = G.nmfma
    assert len(stream) == N
    fillers = {g: [] for g in range(N + 2)}
    for rt in range(1, 8):      # A fragments of row tile rt behind the first MFMAs of row tile rt - 1
        fillers[24 * (rt - 1)] += [("ds", f"A{rt}{c}", G.va(rt & 1, c), G.v_ac(c), G.tile_ofs(rt)) for c in range(3)]
    fillers[barrier_gap] += [("barrier",)]
    # addresses of the next tile's fragments: computed behind the barrier, used from row tile 7 on
    fillers[barrier_gap + 1] += [("valu", f"v_add_u32 {G.v_an(c)}, {S_NXT}, {G.cofs_a[c]}") for c in range(3)]
    fillers[barrier_gap + 2] += [("valu", f"v_add_u32 {G.v_bn(d)}, {S_NXT}, {G.cofs_b[d]}") for d in range(2)]
    fillers[24 * 7] += [("ds", f"N0{c}", G.va(0, c), G.v_an(c), 0) for c in range(3)]
    if pairs:
        for p in range(4):      # next tile's B fragments of column pair p behind the pair's six MFMAs of row tile 7
            fillers[24 * 7 + 6 * (p + 1)] += [("ds", f"NB{ct}{d}", G.vb(ct, d), G.v_bn(d), G.tile_ofs(ct)) for ct in (2 * p, 2 * p + 1) for d in range(2)]
    else:
        # row tile 7: the [b0 | b2] fragments are dead behind its first eight MFMAs, the [b1 | b0] fragment of column ct behind the
        # last MFMA of that column -- the next tile's take their registers, one read per gap
        fillers[24 * 7 + 8] += [("ds", f"NB{ct}0", G.vb(ct, 0), G.v_bn(0), G.tile_ofs(ct)) for ct in range(8)]
        for ct in range(8):
            fillers[24 * 7 + 16 + ct + 1] += [("ds", f"NB{ct}1", G.vb(ct, 1), G.v_bn(1), G.tile_ofs(ct))]
    for r, g in enumerate(req_gaps):
        assert barrier_gap <= g <= N, "requests overwrite the stage of tile t - 1: only behind the mid-tile barrier"
        if dma:
            if m0_early:        # M0 one gap ahead: the MFMA in between is the wait state the LDS-DMA needs behind a write of M0
                assert r == 0 or g - 1 > req_gaps[r - 1], "M0 of a request is written before the previous request has been issued"
                fillers[g - 1].append(("m0", r))
            fillers[g].append(("req", r))
            if split_req:
                fillers[g + 1].append(("reqstep", r))
    # one fragment read per gap (16 cycles): the overflow moves to the following gaps (never across the end of the tile)
    for g in range(N):
        reads = [f for f in fillers[g] if f[0] == "ds"]
        if len(reads) > 1:
            move = reads[1:]
            fillers[g] = [f for f in fillers[g] if f not in move]
            fillers[g + 1] = move + fillers[g + 1]

    def step_of(r):
        op, u, pc = reqs[r]
        b = G.sbase(op, u, pc)
        st_ = SADDR_STEP + (0 if op == "A" else 2)
        return [f"s_add_u32 s{b}, s{b}, s{st_}", f"s_addc_u32 s{b + 1}, s{b + 1}, s{st_ + 1}"]

    def emit_fillers(g):
        for f in fillers[g]:
            if f[0] == "ds":
                e.ds_read(f[1], f[2], f[3], f[4])
            elif f[0] == "valu":
                e.raw(f[1])
            elif f[0] == "barrier":
                if dma:
                    e.raw("s_waitcnt vmcnt(0)")      # this wave's share of tile t + 1 has landed
                if barrier == "full":
                    e.raw("s_barrier")               # publishes tile t + 1; every wave is past tile t - 1
            elif f[0] == "req":
                op, u, pc = reqs[f[1]]
                ofs = (BX_OPER if op == "B" else 0) + pc * BX_PIECE + G.nw * u * 1024
                b = G.sbase(op, u, pc)
                if not m0_early:
                    e.raw(f"s_add_u32 m0, {S_DMA}, {ofs}")
                    if nop_m0:
                        e.raw("s_nop 0")             # SALU write of M0 -> LDS-DMA read of M0: one wait state
                e.raw(f"global_load_lds_dwordx4 {G.v_off}, s[{b}:{b + 1}]")
                if not split_req:
                    for l in step_of(f[1]):
                        e.raw(l)
            elif f[0] == "reqstep":
                for l in step_of(f[1]):
                    e.raw(l)
            elif f[0] == "m0":
                op, u, pc = reqs[f[1]]
                e.raw(f"s_add_u32 m0, {S_DMA}, {(BX_OPER if op == 'B' else 0) + pc * BX_PIECE + G.nw * u * 1024}")

    e.lgkm = [f"A0{c}" for c in range(3)] + [f"B{ct}{d}" for ct, d in next_b_order(pairs)]
    for g, (rt, ct, m) in enumerate(stream):
        emit_fillers(g)
        d = 0 if m == 0 else 1
        want_now = [f"A{rt}{m}", f"B{ct}{d}"]
        if merge_waits and m == 0 and ct == 0:
            want_now += [f"A{rt}1", f"A{rt}2"]       # one wait for the row tile's three A fragments (read 24 MFMAs ago)
        if merge_waits >= 2 and rt == 0 and ct % merge_waits == 0:
            want_now += [f"B{c2}{d}" for c2 in range(ct, min(ct + merge_waits, 8))]
        e.need(want_now)
        # (B fragment first: the instruction computes the transposed tile -- a lane's four values are consecutive in a row of C)
        first, second = (G.vb(ct, d), G.va(rt & 1, m)) if transposed else (G.va(rt & 1, m), G.vb(ct, d))
        e.raw(f"v_mfma_f32_16x16x32_bf16 {G.acc(rt, ct)}, {first}, {second}, {G.acc(rt, ct)}")
    emit_fillers(N)
    emit_fillers(N + 1)
    want = [f"N0{c}" for c in range(3)] + [f"NB{ct}{d}" for ct, d in next_b_order(pairs)]
    assert e.lgkm == want, e.lgkm


def gen16(req_gaps=None, unroll=3, **kw):
    G = Geometry16()
    if req_gaps is None:
        req_gaps = [100, 104, 108, 112, 116, 124, 128, 132, 136, 140, 148, 152]
    e = Emitter()
    e.raw("; ---- entry: stage offsets of tiles t, t + 1, t + 2 from st; scalar bases of the twelve request streams")
    e.raw(f"s_mul_i32 {S_CUR}, {G.st}, {BX_STAGE}")
    e.raw(f"s_add_u32 {S_NXT}, {S_CUR}, {BX_STAGE}")
    e.raw(f"s_cmp_eq_u32 {G.st}, 2")
    e.raw(f"s_cselect_b32 {S_NXT}, 0, {S_NXT}")
    e.raw(f"s_add_u32 {S_NN}, {S_NXT}, {BX_STAGE}")
    e.raw(f"s_cmp_eq_u32 {S_NXT}, {2 * BX_STAGE}")
    e.raw(f"s_cselect_b32 {S_NN}, 0, {S_NN}")
    e.raw(f"s_mov_b32 {S_CNT}, {G.nt}")
    for op in "AB":
        for u in range(G.nu):
            e.raw(f"v_lshl_add_u64 {G.pair(op, u, 0)}, {G.src[(op, u)]}, 0, 0")
            e.raw(f"v_lshl_add_u64 {G.pair(op, u, 1)}, {G.stride[op]}, 0, {G.src[(op, u)]}")
            e.raw(f"v_lshl_add_u64 {G.pair(op, u, 2)}, {G.stride[op]}, 1, {G.src[(op, u)]}")
    first = G.pair("A", 0, 0)
    for op in "AB":
        for u in range(G.nu):
            for pc in range(3):
                b, vp = G.sbase(op, u, pc), G.pair(op, u, pc)
                lo = int(vp[2:].split(":")[0])
                e.raw(f"v_readfirstlane_b32 s{b}, v{lo}")
                e.raw(f"v_readfirstlane_b32 s{b + 1}, v{lo + 1}")
    lo0 = int(first[2:].split(":")[0])
    e.raw(f"v_readfirstlane_b32 {S_TMP}, v{lo0}")
    e.raw("s_nop 4")                                   # VALU-written SGPR -> read by VALU / VMEM
    e.raw(f"v_sub_u32 {G.v_off}, v{lo0}, {S_TMP}")
    e.raw(f"s_mov_b64 s[{SADDR_STEP}:{SADDR_STEP + 1}], {G.step['A']}")
    e.raw(f"s_mov_b64 s[{SADDR_STEP + 2}:{SADDR_STEP + 3}], {G.step['B']}")
    # fragments of tile t in the order in which a tile requests the NEXT tile's (one model of the LDS return queue for every tile)
    for c in range(3):
        e.raw(f"v_add_u32 {G.v_an(c)}, {S_CUR}, {G.cofs_a[c]}")
    for d in range(2):
        e.raw(f"v_add_u32 {G.v_bn(d)}, {S_CUR}, {G.cofs_b[d]}")
    for c in range(3):
        e.ds_read(f"A0{c}", G.va(0, c), G.v_an(c), 0)
    for ct, d in next_b_order(kw.get("pairs", False)):
        e.ds_read(f"B{ct}{d}", G.vb(ct, d), G.v_bn(d), G.tile_ofs(ct))
    label = "BXK_G0_%="
    e.raw(f"{label}:")
    regs = [S_CUR, S_NXT, S_NN]
    assert unroll % 3 == 0
    for k in range(unroll):
        one_tile16(e, G, req_gaps, regs[k % 3], regs[(k + 1) % 3], regs[(k + 2) % 3], **kw)
    e.raw(f"s_sub_u32 {S_CNT}, {S_CNT}, {unroll}")
    e.raw(f"s_cmp_lg_u32 {S_CNT}, 0")
    e.raw(f"s_cbranch_scc1 {label}")
    e.raw("; ---- exit: nothing of the next tile is needed in registers (the C++ side re-reads its fragments)")
    e.raw("s_waitcnt lgkmcnt(0)")
    return e.lines, G


def acc_operands16():
    """The 64 accumulators of the C++ side (acc[i][j][q]) pinned to the AGPRs the block names."""
    ops = []
    for i in range(4):
        for j in range(4):
            for q in range(4):
                k = 16 * (4 * i + j) + 4 * q
                ops.append(f'"+{{a[{k}:{k + 3}]}}"(ACC[{i}][{j}][{q}])')
    return ", ".join(ops)


def render(lines, G, name, saddr=False, unroll=1, acc_ops=None):
    clob = ", ".join([f'"v{k}"' for k in G.clobber_v] + [f'"s{k}"' for k in CLOBBER_S + (CLOBBER_S_SADDR if saddr else [])] + ['"scc"', '"memory"'])
    # (m0 is written too: clang reserves it and refuses it as a clobber; hipcc keeps nothing live in m0 across an asm statement)
    out = (f"#define {name}_TEXT \\\n" + " \\\n".join(f'  "{l}\\n\\t"' for l in lines) + f"\n#define {name}_CLOBBERS {clob}\n#define {name}_UNROLL {unroll}\n")
    if acc_ops is not None:
        out += f"#define {name}_ACC(ACC) {acc_ops}\n"
    return out


HEADER = """// GENERATED by scripts/gen_bx_kloop.py -- do not edit; see that script for the register map, the pipeline invariant at
// entry / exit and the arithmetic contract (bit-identical to the C++ K loop of gemm256_bx_kernel<6>).
"""


def variants():
    """name -> (generator, keywords).  BX_KLOOP_ASM is the product's block (16x16x32 MFMAs), BX_KLOOP_ASM32 the 32x32x16 form of the
    first half of round 6 (built with -DBX_SHAPE16=0 for same-box comparisons); both go to the committed file."""
    spread = [51, 55, 59, 63, 67, 71, 75, 77, 81, 83, 87, 89]      # one request per gap in rows 2 + 3, never beside fragment reads
    row3 = [74, 75, 76, 77, 80, 81, 82, 83, 86, 87, 88, 89]        # all in row 3, one per gap
    best = dict(nw=4, ds_per_gap=2, row0_colmajor=True, addr="saddr")
    early = [99, 101, 103, 105, 107, 109, 111, 113, 115, 117, 119, 123]
    return {
        # ---- the product's blocks
        "BX_KLOOP_ASM": (gen16, dict(m0_early=True)),
        "BX_KLOOP_ASM32": (gen, dict(best, req_gaps=spread, unroll=3)),
        # ---- experiments on the 16x16x32 form (same arithmetic: results stay bit-identical)
        "BX_KLOOP_ASM_F1": (gen16, dict(req_gaps=early)),                       # requests right behind the barrier
        "BX_KLOOP_ASM_F2": (gen16, dict(split_req=False)),
        "BX_KLOOP_ASM_F3": (gen16, dict(barrier_gap=72, req_gaps=[76, 80, 84, 88, 92, 100, 104, 108, 112, 116, 124, 128])),
        "BX_KLOOP_ASM_F4": (gen16, dict(pairs=True)),
        "BX_KLOOP_ASM_F6": (gen16, dict(m0_early=True, merge_waits=1)),         # one lgkm wait per row tile for its A fragments
        "BX_KLOOP_ASM_F7": (gen16, dict(m0_early=True, merge_waits=2)),         # + the B fragments of row tile 0 two at a time
        "BX_KLOOP_ASM_F8": (gen16, dict(m0_early=True, merge_waits=4)),         # + four at a time
        "BX_KLOOP_ASM_F9": (gen16, dict(m0_early=True, transposed=True)),       # B fragment first (-DBX_S16_TRANSPOSED=1): the transposed tile                           # row tiles 0 and 7 column pair by column pair
        "BX_KLOOP_ASM_F5": (gen16, dict(m0_early=False)),                       # M0 written right in front of its request (+ s_nop)
        # ---- TIMING ONLY (wrong results)
        "BX_KLOOP_ASM_U1": (gen16, dict(barrier="none")),
        "BX_KLOOP_ASM_U2": (gen16, dict(dma=False)),
        "BX_KLOOP_ASM_U3": (gen16, dict(dma=False, barrier="none")),
        # ---- experiments on the 32x32x16 form (same arithmetic as BX_KLOOP_ASM32)
        "BX_KLOOP_ASM_E1": (gen, dict(best, req_gaps=spread, unroll=1)),
        "BX_KLOOP_ASM_E2": (gen, dict(best, req_gaps=row3, unroll=3)),
        "BX_KLOOP_ASM_E3": (gen, dict(best, req_gaps=spread, unroll=3, setprio=True)),
        "BX_KLOOP_ASM_E4": (gen, dict(best, req_gaps=spread, unroll=3, nop_m0=False)),
        "BX_KLOOP_ASM_E5": (gen, dict(nw=4, ds_per_gap=2, row0_colmajor=True, req_gaps=spread, unroll=3)),          # 64-bit per-lane addresses
        "BX_KLOOP_ASM_E6": (gen, dict(nw=4)),                                                                       # the C++ loop's schedule
        # ---- attribution builds (TIMING ONLY, wrong results): what one ingredient of the loop costs (profiles/r06_bx_attribution*.log)
        "BX_KLOOP_ASM_T1": (gen, dict(best, req_gaps=spread, unroll=3, barrier="none")),
        "BX_KLOOP_ASM_T2": (gen, dict(best, req_gaps=spread, unroll=3, dma=False)),
        "BX_KLOOP_ASM_T3": (gen, dict(best, req_gaps=spread, unroll=3, dma=False, barrier="none")),
        "BX_KLOOP_ASM_T4": (gen, dict(best, req_gaps=spread, unroll=3, mfma16=True)),
        "BX_KLOOP_ASM_T5": (gen, dict(best, req_gaps=spread, unroll=3, mfma16=True, dma=False, barrier="none")),
    }


PRODUCT_BLOCKS = ("BX_KLOOP_ASM", "BX_KLOOP_ASM32")


def block_text(name):
    fn, kw = variants()[name]
    lines, G = fn(**kw)
    s16 = fn is gen16
    return render(lines, G, name, saddr=s16 or kw.get("addr") == "saddr", unroll=kw.get("unroll", 3 if s16 else 1),
                  acc_ops=acc_operands16() if name == "BX_KLOOP_ASM" else None), lines, G


def product_text():
    """The committed file csrc/bx_kloop_asm.inc."""
    return HEADER + "".join(block_text(name)[0] for name in PRODUCT_BLOCKS)


def main():
    """The product's blocks -> csrc/bx_kloop_asm.inc (committed); with --variants also the experiment / attribution blocks ->
    csrc/bx_kloop_asm_variants.inc (git-ignored; included by gemm_f32.hip only under -DBX_KLOOP_TEXT_OVERRIDE=...,
    scripts/probe/bx_asm_variants.sh)."""
    import sys

    base = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "vivit_amd", "csrc")
    texts = {"bx_kloop_asm.inc": product_text()}
    if "--variants" in sys.argv:
        texts["bx_kloop_asm_variants.inc"] = HEADER
        for name in variants():
            if name not in PRODUCT_BLOCKS:
                text, lines, G = block_text(name)
                texts["bx_kloop_asm_variants.inc"] += text
                print(name, len(lines), "lines,", G.noperands, "operands")
    for fname, text in texts.items():
        with open(os.path.join(base, fname), "w") as f:
            f.write(text)
        print("wrote", os.path.normpath(os.path.join(base, fname)))


if __name__ == "__main__":
    main()
