"""Multi-GPU Gram build: one process per GPU, RCCL (backend "nccl" on ROCm) over xGMI.

The Gram matrix ``G[(c,n),(d,m)] = sum_p V[c,n,p] V[d,m,p]`` couples every pair of samples, so a
batch shard does NOT yield a partial sum.  What is a sum is the contraction
over parameters -- the reference's own ``gram += gram_p`` (vivit/utils/gram.py:104-116).  Ranks
therefore own *column slices* of ``V`` (whole parameters or slices of a parameter's trailing
dims), each builds the ``[n, n]`` partial Gram of its slice on MFMA, and one all-reduce adds them.
The eigensolver's reduction and tridiagonal solve then run replicated (deterministic, no atomics:
every rank holds bit-identical intermediate results, nothing is broadcast); its back-transformations
act on every eigenvector independently, so rank r back-transforms only the r-th slice of the
eigenvectors (``vivit_symeig_rows_f32``) and one all-gather delivers them to everybody (`symeig`).
"""
from typing import Iterable, Optional

import torch
import torch.distributed as dist

from vivit_amd import kernels


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
    """Full Gram matrix on every rank from parameter-sharded factors (partial Gram + all-reduce)."""
    G = partial_gram(local_factors, start_dim=start_dim, out=out)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(G, op=dist.ReduceOp.SUM, group=group)
    return G


def column_slices(num_columns: int, world_size: int):
    """Balanced contiguous column ranges ``[(lo, hi)] * world_size`` of a ``[n, num_columns]`` factor."""
    return [((num_columns * r) // world_size, (num_columns * (r + 1)) // world_size) for r in range(world_size)]


def row_slices(n: int, world_size: int):
    """Equal-length eigenvector ranges (the last ones may be shorter / empty): ``[(lo, hi)] * world_size``."""
    per = -(-n // world_size)
    return [(min(r * per, n), min((r + 1) * per, n)) for r in range(world_size)]


def symeig(G: torch.Tensor, group=None, overwrite: bool = False):
    """Eigenvalues (ascending) and column eigenvectors of the (replicated) symmetric ``G`` with the
    back-transformations sharded over the ranks of ``group``.

    Every rank must hold the same ``G`` (e.g. the result of :func:`sharded_gram`).  Returns
    ``(evals [n], evecs [n, n])`` like ``kernels.symeig(G, eigenvectors=True)``; ``evecs`` is the
    transposed view of the gathered row-major eigenvector matrix (``evecs[:, i]`` contiguous).
    """
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    if world == 1:
        return kernels.symeig(G, eigenvectors=True, overwrite=overwrite)
    n = G.shape[0]
    rank = dist.get_rank(group)
    per = -(-n // world)
    lo, hi = row_slices(n, world)[rank]
    w, Zt_local = kernels.symeig_rows(G, lo, hi, overwrite=overwrite)
    if hi - lo == per:
        send = Zt_local
    else:  # pad the short last slices: all_gather_into_tensor needs equal shapes
        send = torch.zeros((per, n), dtype=Zt_local.dtype, device=Zt_local.device)
        send[: hi - lo] = Zt_local
    Zt = torch.empty((world * per, n), dtype=send.dtype, device=send.device)
    dist.all_gather_into_tensor(Zt, send.contiguous(), group=group)
    return w, Zt[:n].T
