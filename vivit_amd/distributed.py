"""Multi-GPU Gram path: one process per GPU, RCCL (backend "nccl" on ROCm) over xGMI.

The reference is single-process (no collectives).  The layout here is data parallel: rank ``g`` ran
forward/backward on its batch shard ``N_g`` and holds the sqrt-GGN factors of *its* samples only.  The Gram matrix
``G[(c,n),(d,m)] = sum_p V[c,n,p] V[d,m,p]`` couples every pair of samples, so shards are not partial sums
(SURVEY.md 8e); two exchange schemes make them so:

* **materialised factors** ``V_g [C, N_g, *param]`` (generic layers, BackPACK ``SqrtGGN*``): one
  ``all_to_all`` per parameter turns the batch shards into *parameter* shards -- rank ``r`` receives columns
  ``[lo_r, hi_r)`` of every rank's rows, i.e. ``V[:, :, lo_r:hi_r]`` for ALL samples (class-major rows restored on
  receipt).  The contraction over parameters IS a sum (the reference's own ``gram += gram_p``,
  vivit/utils/gram.py:104-116), so each rank runs the full-size MFMA SYRK on ``1/R`` of the contraction length and the
  partial Gram matrices are summed with one all-reduce.  On the point-to-point xGMI fabric the all-to-all uses all
  seven links of every GPU at once and moves ``(R-1)/R`` of ``V`` exactly once (a ring pass of row blocks would move
  ``V`` ``R/2`` times).
* **factorised Linear weights** ``(s_g [C, N_g, out], z_g [N_g, in])`` (vivit/extensions/secondorder/vivit/linear.py:41-42;
  the only form that exists for BASELINE config 5): all-gather the small factors, every rank computes its **block row**
  ``G[N_g, :] = (z_g z^T) o (s_g s^T)`` (``1/R`` of the flops), block rows are all-gathered and stored class-major.

The eigensolver: the full -> band reduction is sharded by block rows of the trailing matrix with a replicated panel
factorisation (:func:`sy2sb_sharded_`: two collectives per panel); bulge chasing and the tridiagonal solve run replicated
(deterministic, no atomics: bit-identical on all ranks, nothing is broadcast); the back-transformations act on every
eigenvector independently, so rank ``r`` back-transforms the ``r``-th slice (``vivit_symeig_banded_rows_f32`` /
``vivit_symeig_rows_f32``) and one all-gather delivers the eigenvectors (:func:`symeig`).  Back-projections ``V v`` are true sums over samples: each rank applies its own rows of ``V`` and one
all-reduce of ``P`` floats finishes the Newton step (vivit/optim/directional_damped_newton.py:370-373;
:func:`all_reduce_sum_`).

For functional tests several ranks may share one GPU with the ``gloo`` backend; collectives on device tensors are then
staged through host memory (``_staged``) -- test plumbing, never the measured path.
"""
import os as _os
from typing import Iterable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from vivit_amd import kernels

SMALL_PARAM_COLUMNS = 64  # parameters with fewer columns per rank than this are owned whole by one rank
EXCHANGE_CHUNK_COLUMNS = 8192  # columns of a rank's parameter shard exchanged (and multiplied) at a time: the all-to-all
                               # of chunk j + 1 runs on RCCL's stream while the SYRK of chunk j runs (at n = 40 960, R = 8:
                               # 168 MB per peer and chunk, ~3 ms over one xGMI link, against a ~60 ms SYRK)


# ------------------------------------------------------------------------------------------------------------------
# collectives (thin wrappers: RCCL directly; gloo + device tensors are staged through the host for 1-GPU functional tests)
def _forced() -> bool:
    """``VIVIT_DIST_FORCE_COLLECTIVES=1``: a world of ONE rank issues every collective anyway (an all-to-all with itself,
    an all-reduce over one rank, ...) instead of taking the single-process shortcuts.  That is how a one-GPU box executes the
    RCCL code path of this module -- ``init_process_group("nccl")``, ``all_to_all_single(async_op=True)`` + ``work.wait()``
    against the SYRK stream, the packed all-reduce -- at least once (tests/test_distributed_gpu.py); results are unchanged."""
    return _os.environ.get("VIVIT_DIST_FORCE_COLLECTIVES") == "1" and dist.is_available() and dist.is_initialized()


def _active(group=None) -> bool:
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or _forced())


def _alone(R: int) -> bool:
    """True when the single-process shortcut may be taken (one rank and collectives not forced)."""
    return R == 1 and not _forced()


def world_size(group=None) -> int:
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def rank_of(group=None) -> int:
    return dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0


def _staged(t: torch.Tensor, group) -> bool:
    return t.is_cuda and dist.get_backend(group) == "gloo"


def all_reduce_sum_(t: torch.Tensor, group=None) -> torch.Tensor:
    """In-place sum over the ranks of ``group`` (the Gram partials, the ``P``-float Newton step, ...)."""
    if not _active(group):
        return t
    if _staged(t, group):
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
    else:
        if not t.is_contiguous():
            raise ValueError("all_reduce_sum_ needs a contiguous tensor")
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def all_reduce_sym_(G: torch.Tensor, group=None) -> torch.Tensor:
    """In-place sum over the ranks of a SYMMETRIC matrix (every rank's summand symmetric): only the packed lower
    triangle travels (n (n + 1) / 2 floats instead of n^2 -- the all-reduce is bound by the bytes per xGMI link), the
    upper triangle is mirrored locally afterwards (``vivit_pack_lower_f32`` / ``vivit_unpack_lower_f32``)."""
    if not _active(group):
        return G
    packed = kernels.pack_lower(G)
    all_reduce_sum_(packed, group)
    return kernels.unpack_lower_(packed, G)


def all_gather_cat(t: torch.Tensor, group=None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``[R * t.shape[0], *t.shape[1:]]``: the ranks' equally shaped ``t`` stacked along dim 0 in rank order."""
    R = world_size(group)
    t = t.contiguous()
    if out is None:
        out = torch.empty((R * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    if _alone(R):
        out.copy_(t)
        return out
    if _staged(t, group):
        h = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(h, t.cpu(), group=group)
        out.copy_(h)
    else:
        dist.all_gather_into_tensor(out, t, group=group)
    return out


def broadcast_(t: torch.Tensor, src: int, group=None) -> torch.Tensor:
    """In place: rank ``src``'s contiguous ``t`` to every rank of ``group`` (``src``: rank within the group)."""
    if not _active(group):
        return t
    gsrc = dist.get_global_rank(group, src) if group is not None else src
    if _staged(t, group):
        h = t.cpu()
        dist.broadcast(h, gsrc, group=group)
        t.copy_(h)
    else:
        if not t.is_contiguous():
            raise ValueError("broadcast_ needs a contiguous tensor")
        dist.broadcast(t, gsrc, group=group)
    return t


class _Done:
    def wait(self):
        return True


def all_to_all_flat(send: torch.Tensor, in_splits: Sequence[int], out_splits: Sequence[int], group=None,
                    out: Optional[torch.Tensor] = None, async_op: bool = False):
    """1-D all-to-all with per-peer element counts; returns the received flat buffer (peer order), ``out`` if given
    (contiguous, ``sum(out_splits)`` elements).  ``async_op``: returns ``(recv, work)`` instead; ``work.wait()`` makes
    the current stream wait for the exchange (RCCL runs it on its own stream behind the work already queued here)."""
    recv = torch.empty(int(sum(out_splits)), dtype=send.dtype, device=send.device) if out is None else out
    work = _Done()
    if _staged(send, group):
        h = torch.empty(recv.shape, dtype=recv.dtype)
        dist.all_to_all_single(h, send.cpu(), list(out_splits), list(in_splits), group=group)
        recv.copy_(h)
    else:
        w = dist.all_to_all_single(recv, send, list(out_splits), list(in_splits), group=group, async_op=async_op)
        if async_op:
            work = w
    return (recv, work) if async_op else recv


# ------------------------------------------------------------------------------------------------------------------
def column_slices(num_columns: int, world: int):
    """Balanced contiguous column ranges ``[(lo, hi)] * world`` of a ``[n, num_columns]`` factor."""
    return [((num_columns * r) // world, (num_columns * (r + 1)) // world) for r in range(world)]


def row_slices(n: int, world: int):
    """Equal-length eigenvector ranges (the last ones may be shorter / empty): ``[(lo, hi)] * world``."""
    per = -(-n // world)
    return [(min(r * per, n), min((r + 1) * per, n)) for r in range(world)]


def parameter_slices(num_columns: int, world: int, index: int = 0):
    """Column ranges of one parameter's factor owned by each rank after the all-to-all.  Large parameters are cut
    into balanced contiguous ranges; a parameter too small to give every rank ``SMALL_PARAM_COLUMNS`` columns goes
    whole to rank ``index % world`` (a rank-``k`` SYRK with tiny ``k`` costs an ``n^2`` read-modify-write per rank)."""
    if num_columns < SMALL_PARAM_COLUMNS * world:
        owner = index % world
        return [(0, num_columns) if r == owner else (0, 0) for r in range(world)]
    return column_slices(num_columns, world)


def partial_gram(local_factors: Iterable[torch.Tensor], start_dim: int = 2, out: Optional[torch.Tensor] = None):
    """Gram of this rank's factor slices: sum over slices of ``A A^T`` (``A`` = slice viewed
    ``[prod(leading dims), -1]``), accumulated in-kernel (beta = 1)."""
    G = out
    first = True
    for f in local_factors:
        lead = 1
        for s in f.shape[:start_dim]:
            lead *= int(s)
        A = f.detach().reshape(lead, -1)
        if G is None:
            G = kernels.gram_syrk(A)
        else:
            kernels.gram_syrk(A, out=G, alpha=1.0, beta=0.0 if first else 1.0)
        first = False
    return G


def sharded_gram(local_factors: Iterable[torch.Tensor], start_dim: int = 2, group=None,
                 out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Full Gram matrix on every rank from PARAMETER-sharded factors (partial Gram + all-reduce): the layout of a
    tensor-parallel layer, and the second half of the batch-sharded scheme below."""
    G = partial_gram(local_factors, start_dim=start_dim, out=out)
    return all_reduce_sum_(G, group)


def to_parameter_shard(local: torch.Tensor, lead_dims: int, group=None, index: int = 0) -> torch.Tensor:
    """Batch shard -> parameter shard of one factor by all-to-all.

    ``local``: this rank's ``[C, N_g, *param]`` (``lead_dims = 2``; sqrt-GGN factor) or ``[N_g, *param]``
    (``lead_dims = 1``; per-sample gradients).  Returns ``[C * N, w]`` (resp. ``[N, w]``): ALL samples (rank-major
    order ``n = g N_g + n_local`` inside each class, i.e. the reference's class-major rows ``c N + n``), columns
    ``[lo, hi)`` of the flattened parameter owned by this rank (:func:`parameter_slices`), ``w = hi - lo`` (may be 0).
    """
    R, me = world_size(group), rank_of(group)
    if lead_dims == 1:
        local = local.unsqueeze(0)
    C, Ng = int(local.shape[0]), int(local.shape[1])
    F = local.detach().reshape(C, Ng, -1)
    P = F.shape[2]
    if _alone(R):
        return F.reshape(C * Ng, P)
    slices = parameter_slices(P, R, index)
    widths = [hi - lo for lo, hi in slices]
    w_mine = widths[me]
    nch = -(-max(widths) // EXCHANGE_CHUNK_COLUMNS)
    if nch <= 1:
        return _exchange_columns(F, [(lo, hi) for lo, hi in slices], w_mine, group)
    # column chunks into the final buffer: the peak beside the result is one chunk's send, receive and permuted copy
    # (a one-shot exchange holds three more copies of the whole shard)
    out = torch.empty(C * Ng * R, w_mine, dtype=F.dtype, device=F.device)
    for j in range(nch):
        cols = [(lo + min(j * EXCHANGE_CHUNK_COLUMNS, w), lo + min((j + 1) * EXCHANGE_CHUNK_COLUMNS, w))
                for (lo, _), w in zip(slices, widths)]
        j0, j1 = cols[me][0] - slices[me][0], cols[me][1] - slices[me][0]
        part = _exchange_columns(F, cols, j1 - j0, group)
        if j1 > j0:
            out[:, j0:j1].copy_(part)
    return out


def _pack_columns(F: torch.Tensor, cols: Sequence[Tuple[int, int]]) -> Tuple[torch.Tensor, List[int]]:
    """Send buffer of one exchange: for every peer ``r`` the columns ``cols[r]`` of ``F [C, N_g, P]`` (all classes, all
    local samples), peer-major.  Equal-width slices at a constant pitch are ONE strided copy; otherwise one copy per peer."""
    C, Ng, P = F.shape
    R = len(cols)
    widths = [hi - lo for lo, hi in cols]
    splits = [C * Ng * w for w in widths]
    w0 = widths[0]
    pitch = cols[1][0] - cols[0][0] if R > 1 else 0
    if R > 1 and w0 > 0 and all(w == w0 for w in widths) and all(cols[r][0] == cols[0][0] + r * pitch for r in range(R)) and pitch > 0:
        view = F.as_strided((R, C, Ng, w0), (pitch * F.stride(2), F.stride(0), F.stride(1), F.stride(2)),
                            F.storage_offset() + cols[0][0] * F.stride(2))
        return view.contiguous().reshape(-1), splits
    send = torch.empty(sum(splits), dtype=F.dtype, device=F.device)
    off = 0
    for (lo, hi), cnt in zip(cols, splits):
        if cnt > 0:
            send[off:off + cnt].view(C, Ng, hi - lo).copy_(F[:, :, lo:hi])
        off += cnt
    return send, splits


def _class_major(recv: torch.Tensor, R: int, C: int, Ng: int, w: int) -> torch.Tensor:
    """Received ``[R, C, N_g, w]`` (peer-major) -> ``[C R N_g, w]``: the reference's class-major rows ``c N + n``."""
    if C == 1 or R == 1:
        return recv.view(C * R * Ng, w)
    return recv.view(R, C, Ng, w).permute(1, 0, 2, 3).reshape(C * R * Ng, w)


def _exchange_columns(F: torch.Tensor, cols, w_mine: int, group) -> torch.Tensor:
    C, Ng, _ = F.shape
    R = len(cols)
    send, in_splits = _pack_columns(F, cols)
    recv = all_to_all_flat(send, in_splits, [C * Ng * w_mine] * R, group)
    return _class_major(recv, R, C, Ng, w_mine)


def check_equal_shards(n_local: int, group=None, *more) -> int:
    """Global batch size; raises if the ranks' shards differ in size (the block layout needs equal shards).

    ``more``: further per-rank sizes (``None`` allowed) checked in the SAME collective -- every rank makes exactly one
    call whatever its arguments are, so ranks that disagree on an optional size raise instead of hanging."""
    R = world_size(group)
    if _alone(R):
        return n_local
    mine = (int(n_local),) + tuple(-1 if m is None else int(m) for m in more)
    sizes = [None] * R
    dist.all_gather_object(sizes, mine, group=group)
    if any(s != sizes[0] for s in sizes):
        raise ValueError(f"batch shards must have equal size on every rank, got {sizes}")
    return n_local * R


class BatchShardedGram:
    """Accumulate ``V^T V`` (and optionally ``V^T g``) of one parameter group from BATCH-sharded factors.

    ``C``: classes / MC samples, ``N_local``: this rank's samples.  After the ``add_*`` calls of all parameters,
    :meth:`finalize` returns the full ``[C, N, C, N]`` Gram matrix (``N = R N_local``, class-major like the reference:
    vivit/utils/gram.py:58-69) on every rank; :meth:`finalize_vtg` the ``[C, N, N_grad]`` dot products.
    """

    def __init__(self, C: int, N_local: int, group=None, N_grad_local: Optional[int] = None):
        self.group = group
        self.R, self.me = world_size(group), rank_of(group)
        self.C, self.Ng = int(C), int(N_local)
        # unequal shards would give mismatched collective sizes (hang / corruption); one collective for both sizes
        check_equal_shards(self.Ng, group, N_grad_local)
        self.N = self.Ng * self.R
        self.n = self.C * self.N
        self.Mg = None if N_grad_local is None else int(N_grad_local)
        self.M = None if self.Mg is None else self.Mg * self.R
        self.partial = None    # [n, n]   parameter-sharded partial sum (all-to-all path)
        self.rows = None       # [C N_g, n] block row of this rank (factorised / all-gather path)
        self.partial_g = None  # [n, M]
        self.rows_g = None     # [C N_g, M]
        self._count = 0

    # -- materialised factors: all-to-all to parameter shards, full-size SYRK on 1/R of the contraction ------------
    def add_factor(self, V_local: torch.Tensor, g_local: Optional[torch.Tensor] = None):
        """``V_local: [C, N_g, *param]``; ``g_local: [M_g, *param]`` adds ``V^T g`` (K2) of the same parameter.

        The rank's column range of the parameter is exchanged and multiplied in chunks of ``EXCHANGE_CHUNK_COLUMNS``:
        the all-to-all of chunk ``j + 1`` is issued (asynchronously, on RCCL's stream) before the SYRK of chunk ``j``
        is launched, so the exchange hides behind the matrix work; only the first chunk's exchange is exposed."""
        idx = self._count
        self._count += 1
        C, Ng, R, me = self.C, self.Ng, self.R, self.me
        F = V_local.detach().reshape(C, Ng, -1)
        P = F.shape[2]
        Fg = None if g_local is None else g_local.detach().reshape(1, self.Mg, -1)
        slices = parameter_slices(P, R, idx)
        widths = [hi - lo for lo, hi in slices]
        if self.partial is None:
            self.partial = torch.zeros((self.n, self.n), dtype=torch.float32, device=V_local.device)
        if g_local is not None and self.partial_g is None:
            self.partial_g = torch.zeros((self.n, self.M), dtype=torch.float32, device=V_local.device)
        nch = -(-max(widths) // EXCHANGE_CHUNK_COLUMNS)

        def issue(j):
            cols = [(lo + min(j * EXCHANGE_CHUNK_COLUMNS, w), lo + min((j + 1) * EXCHANGE_CHUNK_COLUMNS, w))
                    for (lo, _), w in zip(slices, widths)]
            wc = cols[me][1] - cols[me][0]
            send, in_splits = _pack_columns(F, cols)
            recv, work = all_to_all_flat(send, in_splits, [C * Ng * wc] * R, self.group, async_op=True)
            out = [(send, recv, work, wc)]
            if Fg is not None:
                sg, isg = _pack_columns(Fg, cols)
                rg, wg = all_to_all_flat(sg, isg, [self.Mg * wc] * R, self.group, async_op=True)
                out.append((sg, rg, wg, wc))
            return out

        pending = issue(0) if nch > 0 else None
        for j in range(nch):
            nxt = issue(j + 1) if j + 1 < nch else None
            (_, recv, work, wc) = pending[0]
            work.wait()
            if wc > 0:
                A = _class_major(recv, R, C, Ng, wc)
                kernels.gram_syrk(A, out=self.partial, alpha=1.0, beta=1.0)
                if Fg is not None:
                    (_, rg, wg, _) = pending[1]
                    wg.wait()
                    kernels.gemm_nt(A, rg.view(self.M, wc), out=self.partial_g, alpha=1.0, beta=1.0)
            elif Fg is not None:
                pending[1][2].wait()
            pending = nxt

    # -- small materialised factors (biases, ...): all-gather the factor, block row by one NT GEMM -----------------
    def add_factor_rows(self, V_local: torch.Tensor, g_local: Optional[torch.Tensor] = None):
        self._count += 1
        C, Ng = self.C, self.Ng
        A = V_local.detach().reshape(C * Ng, -1)
        full = self._gather_class_major(V_local.detach().reshape(C, Ng, -1))        # [C, N, P]
        self.rows = kernels.gemm_nt(A, full.reshape(self.n, -1), out=self.rows, alpha=1.0,
                                    beta=0.0 if self.rows is None else 1.0)
        if g_local is not None:
            gfull = all_gather_cat(g_local.detach().reshape(self.Mg, -1), self.group)  # [M, P]
            self.rows_g = kernels.gemm_nt(A, gfull, out=self.rows_g, alpha=1.0, beta=0.0 if self.rows_g is None else 1.0)

    # -- factorised Linear weight: all-gather (s, z), block row (z_g z^T) o (s_g s^T) -------------------------------
    def add_linear(self, s_local: torch.Tensor, z_local: torch.Tensor, delta_local: Optional[torch.Tensor] = None,
                   zg_local: Optional[torch.Tensor] = None):
        """``s_local: [C, N_g, out]``, ``z_local: [N_g, in]`` (``V_t[c,n,o,i] = s[c,n,o] z[n,i]``);
        ``delta_local: [M_g, out]`` with ``zg_local: [M_g, in]`` (default ``z_local``): the per-sample gradient
        ``g[m,o,i] = delta[m,o] zg[m,i]``, adds ``V^T g``."""
        self._count += 1
        C, Ng, N = self.C, self.Ng, self.N
        s_local, z_local = s_local.detach().contiguous(), z_local.detach().contiguous()
        z = all_gather_cat(z_local, self.group)                                     # [N, in]
        s = self._gather_class_major(s_local)                                       # [C, N, out]
        Gz = kernels.gemm_nt(z_local, z)                                            # [N_g, N]
        Gs = kernels.gemm_nt(s_local.reshape(C * Ng, -1), s.reshape(C * N, -1))     # [C N_g, C N]
        self.rows = kernels.gram_hadamard_block(Gz, Gs, C, Ng, C, N, out=self.rows, alpha=1.0,
                                                beta=0.0 if self.rows is None else 1.0)
        if delta_local is not None:
            zg_local = z_local if zg_local is None else zg_local.detach().contiguous()
            zg = all_gather_cat(zg_local, self.group)                               # [M, in]
            delta = all_gather_cat(delta_local.detach().contiguous(), self.group)   # [M, out]
            Gzg = kernels.gemm_nt(z_local, zg)                                      # [N_g, M]
            Gsd = kernels.gemm_nt(s_local.reshape(C * Ng, -1), delta)               # [C N_g, M]
            self.rows_g = kernels.gram_hadamard_block(Gzg, Gsd, C, Ng, 1, self.M, out=self.rows_g, alpha=1.0,
                                                      beta=0.0 if self.rows_g is None else 1.0)

    def _gather_class_major(self, t_local: torch.Tensor) -> torch.Tensor:
        """``[C, N_g, X]`` per rank -> ``[C, N, X]`` (samples in rank order inside each class)."""
        C, Ng = self.C, self.Ng
        g = all_gather_cat(t_local.reshape(1, C, Ng, -1), self.group)               # [R, C, N_g, X]
        if C == 1 or self.R == 1:
            return g.reshape(C, self.N, -1)
        return g.permute(1, 0, 2, 3).reshape(C, self.N, -1)

    def _assemble(self, rows, partial, width, symmetric=False):
        """Block rows ``[C N_g, width]`` of all ranks -> class-major ``[n, width]``, plus the summed partials
        (``symmetric``: every rank's partial is a sum of SYRKs, so only its packed lower triangle is all-reduced)."""
        C, Ng, R, n = self.C, self.Ng, self.R, self.n
        out = None
        if partial is not None:
            out = all_reduce_sym_(partial, self.group) if symmetric else all_reduce_sum_(partial, self.group)
        if rows is not None:
            g = all_gather_cat(rows.view(1, C * Ng, width), self.group)            # [R, C N_g, width]
            if C > 1 and R > 1:  # class-major store: row (c, g N_g + n_local)
                g = g.view(R, C, Ng, width).permute(1, 0, 2, 3)
            g = g.reshape(n, width)
            out = g if out is None else out.add_(g)
        return out

    def finalize(self) -> torch.Tensor:
        """The group's Gram matrix ``[C, N, C, N]``, identical on every rank."""
        if self.rows is None and self.partial is None:
            raise ValueError("no factor was added")
        G = self._assemble(self.rows, self.partial, self.n, symmetric=True)
        self.rows = self.partial = None
        return G.view(self.C, self.N, self.C, self.N)

    def finalize_vtg(self) -> torch.Tensor:
        """``V^T g`` as ``[C, N, N_grad]`` (vivit/optim/directional_damped_newton.py:255), identical on every rank."""
        if self.rows_g is None and self.partial_g is None:
            raise ValueError("no gradient factor was added")
        A = self._assemble(self.rows_g, self.partial_g, self.M)
        self.rows_g = self.partial_g = None
        return A.view(self.C, self.N, self.M)

    def local_samples(self, t: torch.Tensor, dim: int) -> torch.Tensor:
        """Slice of a global-sample axis (length ``N``) that belongs to this rank."""
        return t.narrow(dim, self.me * self.Ng, self.Ng)


def batch_sharded_gram(local_factors: Sequence[torch.Tensor], group=None) -> torch.Tensor:
    """``[n, n]`` Gram matrix on every rank from batch-sharded materialised factors ``[C, N_g, *param]``."""
    C, Ng = int(local_factors[0].shape[0]), int(local_factors[0].shape[1])
    acc = BatchShardedGram(C, Ng, group)
    for V in local_factors:
        acc.add_factor(V)
    return acc.finalize().view(acc.n, acc.n)


def backproject_sum(coef: torch.Tensor, V_local: torch.Tensor, acc: BatchShardedGram) -> torch.Tensor:
    """``sum_{c,n} coef[k, c, n] V[c, n, ...]`` over ALL samples: every rank applies its own rows of ``V``, one
    all-reduce of ``K P`` floats sums the shards (the Newton step, directional_damped_newton.py:370-373, K = 1;
    the eigenvector back-projection, vivit/utils/ggn.py:94-115).  ``coef: [K, C, N]`` (global), ``V_local:
    [C, N_g, *param]``."""
    K = coef.shape[0]
    mine = acc.local_samples(coef.reshape(K, acc.C, acc.N), 2).reshape(K, acc.C * acc.Ng).contiguous()
    out = kernels.gemm_nn(mine, V_local.detach().reshape(acc.C * acc.Ng, -1))
    all_reduce_sum_(out, acc.group)
    return out.view(K, *V_local.shape[2:])


# The band reduction is sharded by default when at least SHARDED_BAND_MIN_RANKS ranks hold a matrix of at least
# SHARDED_BAND_MIN_N rows (VIVIT_SHARDED_BAND_MIN_N / VIVIT_SHARDED_BAND_MIN_RANKS; ``symeig(..., sharded_reduction=...)`` decides
# per call).  Below that the replicated reduction wins: every panel costs one collective on the critical path and ~25 small
# launches from Python, while the sharded passes over the trailing matrix only save (R - 1) / R of 0.55 s at n = 40 960.
# NOT MEASURED on more than one GPU (no node): the thresholds follow the projection in DESIGN.md section 6.
SHARDED_BAND_MIN_N = int(_os.environ.get("VIVIT_SHARDED_BAND_MIN_N", "8192"))
SHARDED_BAND_MIN_RANKS = int(_os.environ.get("VIVIT_SHARDED_BAND_MIN_RANKS", "4"))
# collectives issued by the last sy2sb_sharded_ call of this process: {"all_gather": on the critical path, one per panel;
# "broadcast": the next panel's stale block row, asynchronous, off the critical path; "total"} -- bench.py reports it
LAST_SHARDED_COLLECTIVES = {}


def _bcast_async(t: torch.Tensor, src: int, group):
    """Broadcast ``t`` from group rank ``src``; returns a handle whose ``wait()`` orders the current stream behind it (RCCL runs
    it on its own stream: the caller keeps queueing work that does not need ``t``).  gloo + device tensors: staged, synchronous."""
    if not _active(group):
        return _Done()
    if _staged(t, group) or dist.get_backend(group) == "gloo":
        broadcast_(t, src, group)
        return _Done()
    gsrc = dist.get_global_rank(group, src) if group is not None else src
    return dist.broadcast(t, gsrc, group=group, async_op=True)


def sy2sb_sharded_(A: torch.Tensor, group=None) -> torch.Tensor:
    """Band reduction of the replicated symmetric ``A`` (BOTH triangles valid: ``kernels.symeig_prepare_``) with the
    trailing matrix SHARDED over the ranks -- in place: on return every rank's ``A`` holds the band
    (``A[i][j]``, ``0 <= i - j <= 64``) and the first-stage reflectors exactly where the single-GPU
    ``vivit_sy2sb_f32`` leaves them; returns ``tau1 [n]``.

    Rows are dealt to the ranks in blocks of 64 (block ``b`` -> rank ``b % R``: the work stays balanced while the
    trailing matrix shrinks) and, the matrix being symmetric and stored in full, a rank's rows ARE its block columns.
    Panel ``p`` (the reference has no counterpart: ``Tensor.symeig`` is one LAPACK call, vivit/linalg/eigh.py:248-250):

      0. (off the critical path) the owner of row block ``p + 1`` broadcasts it AS IT IS NOW -- stale with respect to panel
         ``p`` -- asynchronously; it lands while the steps below run;
      1. every rank factors panel ``p`` itself from block row ``p`` (``vivit_sy2sb_panel_qr_f32``: same kernels, same data
         -- bit-identical ``V``, ``T`` everywhere) and files band entries and reflectors;
      2. ``P = A22 V``: every rank multiplies its own rows (``1/R`` of the streamed panel product); ONE all-gather of
         ``mp / R x 64`` floats puts ``P`` together -- the panel's only collective on the critical path;
      3. ``W = X - V (T^T (V^T X)) / 2`` with ``X = P T`` -- four 64-wide products, replicated;
      4. ``A22 -= V W^T + W V^T`` on the own rows only (``1/R`` of the rank-128 update), no communication;
      5. every rank applies the same update to the stale block row of step 0 (a 64 x mp x 128 product, replicated): that IS
         block row ``p + 1`` as panel ``p + 1`` needs it -- round 5 broadcast it after step 4, a second collective per panel
         on the critical path.

    Row sets are whole 64-row blocks in rank-cyclic order, so "rank r's rows of the trailing matrix" is a strided view of a
    block-padded buffer: no index tensors, ~25 launches per panel.  Replicated per rank: the panel QR and the 64-wide
    products; sharded: the two passes over the trailing matrix (0.30 s + 0.26 s of 0.94 s at n = 40 960 on one GPU)."""
    NB = kernels.BAND_NB
    n = A.shape[0]
    R, q = world_size(group), rank_of(group)
    nblk = -(-n // NB)
    dev = A.device
    f32 = torch.float32
    count = {"all_gather": 0, "broadcast": 0}
    # own rows, whole blocks: local block k holds global block q + k R (the last global block may be short: zero rows below it)
    nloc = len(range(q, nblk, R))
    Aloc = torch.zeros((nloc * NB, n), dtype=f32, device=dev)
    for k, b in enumerate(range(q, nblk, R)):
        rows = min((b + 1) * NB, n) - b * NB
        Aloc[k * NB:k * NB + rows] = A[b * NB:b * NB + rows]
    tau1 = torch.zeros(n, dtype=f32, device=dev)

    def block_row(b, j0):
        """Block row b from column j0 on, from its owner's rows (zero elsewhere): [rows of block b, n - j0]."""
        rows = min((b + 1) * NB, n) - b * NB
        t = torch.zeros((rows, n - j0), dtype=f32, device=dev)
        if q == b % R:
            t.copy_(Aloc[(b // R) * NB:(b // R) * NB + rows, j0:])
        return t

    Bt = block_row(0, 0)
    broadcast_(Bt, 0, group)
    count["broadcast"] += 1
    gi_last = None
    for p in range(nblk):
        j0, gi = p * NB, (p + 1) * NB
        mp = n - gi
        if mp <= 0:
            break
        nb_live = -(-mp // NB)                                     # blocks p + 1 .. nblk - 1
        # ---- 0. the next block row, stale, on its way while this panel is worked on
        stale = block_row(p + 1, gi)
        handle = _bcast_async(stale, (p + 1) % R, group)
        count["broadcast"] += 1
        # ---- 1. panel QR (replicated) on block row p = [D | panel^T]
        pan = Bt[:, NB:].t().contiguous()                          # [mp, 64]
        Vt, tau, betas, T = kernels.panel_qr_(pan)
        A[j0:gi, j0:gi] = Bt[:, :NB]                               # diagonal block of the band
        top = min(NB, mp)
        A[gi:gi + top, j0:gi] = torch.triu(pan[:top], 1) + torch.diag(betas)[:top]   # R: the sub-diagonal block of the band
        A[j0:gi, gi:] = Vt                                         # reflector rows (dead upper triangle)
        tau1[j0:gi] = tau
        # ---- 2. P = A22 V on the own live blocks, one all-gather
        lb0 = 0 if p < q else (p - q) // R + 1                     # own blocks with global index <= p are finished
        live = Aloc[lb0 * NB:, gi:]                                # [nl, mp] view (row stride n); nl a multiple of 64
        nl = live.shape[0]
        kmax = -(-(nblk - 1 - p) // R)                             # most live blocks any rank has
        send = torch.zeros((kmax * NB, NB), dtype=f32, device=dev)
        if nl > 0:
            kernels.gemm_nt(live, Vt, out=send[:nl])
        gathered = all_gather_cat(send, group).view(R, kmax, NB, NB)
        count["all_gather"] += 1
        Ppad = torch.zeros((nb_live, NB, NB), dtype=f32, device=dev)          # P by blocks p + 1 .., padded to whole blocks
        for r in range(R):
            first = (r - (p + 1)) % R                                          # rank r's first live block, relative to block p + 1
            cnt = len(range(first, nb_live, R))
            if cnt:
                Ppad[first::R] = gathered[r, :cnt]
        P = Ppad.view(nb_live * NB, NB)[:mp]
        # ---- 3. W (replicated, 64-wide)
        X = kernels.gemm_nn(P, T)
        S2 = kernels.gemm_nn(Vt, X)
        Y = kernels.gemm_tn(T, S2)
        W = kernels.gemm_tn(Vt, Y, out=X, alpha=-0.5, beta=1.0)    # W = X - V Y / 2
        Rm = torch.cat([W, Vt.t()], 1).contiguous()                # [mp, 128] = [W | V]
        VW = torch.zeros((nb_live * NB, 2 * NB), dtype=f32, device=dev)       # [V | W] by blocks, padded to whole blocks
        VW[:mp, :NB] = Vt.t()
        VW[:mp, NB:] = W
        # ---- 4. own rows of the trailing matrix
        if nl > 0:
            first = (q - (p + 1)) % R
            Lm = VW.view(nb_live, NB, 2 * NB)[first::R].reshape(-1, 2 * NB)   # [nl, 128] = [V_l | W_l]
            kernels.gemm_nt(Lm, Rm, out=live, alpha=-1.0, beta=1.0)
        # ---- 5. block row p + 1 for the next panel: the stale copy + this panel's update, on every rank
        handle.wait()
        rows = stale.shape[0]
        kernels.gemm_nt(VW[:rows].contiguous(), Rm, out=stale, alpha=-1.0, beta=1.0)
        Bt = stale
        gi_last = gi
    # ---- what is left of the trailing matrix (at most one block): the last diagonal block of the band
    if gi_last is not None and gi_last < n:
        m = n - gi_last
        A[gi_last:, gi_last:] = Bt[:m, :m]
    count["total"] = count["all_gather"] + count["broadcast"]
    LAST_SHARDED_COLLECTIVES.clear()
    LAST_SHARDED_COLLECTIVES.update(count)
    return tau1


def symeig(G: torch.Tensor, group=None, overwrite: bool = False, sharded_reduction: Optional[bool] = None):
    """Eigenvalues (ascending) and column eigenvectors of the (replicated) symmetric ``G`` with the band reduction and
    the back-transformations sharded over the ranks of ``group``.

    Every rank must hold the same ``G`` (e.g. the result of :func:`sharded_gram`).  Returns
    ``(evals [n], evecs [n, n])`` like ``kernels.symeig(G, eigenvectors=True)``; ``evecs`` is the
    transposed view of the gathered row-major eigenvector matrix (``evecs[:, i]`` contiguous).
    ``sharded_reduction``: shard the full -> band reduction too (:func:`sy2sb_sharded_`); default: on for at least
    ``SHARDED_BAND_MIN_RANKS`` (4) ranks and ``n >= SHARDED_BAND_MIN_N`` (8192).  Bulge chasing and the tridiagonal solve stay
    replicated (L2-resident, launch-free: nothing to shard)."""
    world = world_size(group)
    if _alone(world):
        return kernels.symeig(G, eigenvectors=True, overwrite=overwrite)
    n = G.shape[0]
    rank = rank_of(group)
    per = -(-n // world)
    lo, hi = row_slices(n, world)[rank]
    if sharded_reduction is None:
        sharded_reduction = world >= SHARDED_BAND_MIN_RANKS and n >= SHARDED_BAND_MIN_N
    if sharded_reduction and n > 2 * kernels.BAND_NB:
        A = G if overwrite else G.clone()
        scal = kernels.symeig_prepare_(A)
        tau1 = sy2sb_sharded_(A, group)
        w, Zt_local = kernels.symeig_banded_rows(A, tau1, scal, lo, hi)
    else:
        w, Zt_local = kernels.symeig_rows(G, lo, hi, overwrite=overwrite)
    if hi - lo == per:
        send = Zt_local
    else:  # pad the short last slices: all_gather_into_tensor needs equal shapes
        send = torch.zeros((per, n), dtype=Zt_local.dtype, device=Zt_local.device)
        send[: hi - lo] = Zt_local
    Zt = all_gather_cat(send, group)
    return w, Zt[:n].T
