"""Gram-type contractions on the MI355X kernels (API of ``vivit.utils.gram``).

The reference builds einsum strings (vivit/utils/gram.py:206-232); every such contraction is
"flatten the leading dims, contract all trailing dims", i.e. one NT GEMM (or a SYRK when both
operands are the same tensor), which is what runs here on MFMA through libvivit_hip.so.
"""
import math
from typing import Iterable, Tuple

from torch import Tensor, cat

from vivit_amd import kernels

MAX_LETTERS = 26  # the reference's einsum-letter cap is observable behaviour (gram.py:50-55)


def get_letters(num_letters: int) -> str:
    """Kept for API parity: ``num_letters`` unique letters, ValueError beyond the alphabet."""
    if num_letters > MAX_LETTERS:
        raise ValueError(f"Requested too many letters {num_letters}>{MAX_LETTERS}")
    return "".join(chr(ord("a") + i) for i in range(num_letters))


def _numel(shape) -> int:
    out = 1
    for s in shape:
        out *= int(s)
    return out


def reshape_as_square(tensor: Tensor) -> Tensor:
    """View any tensor with a square number of elements as ``[dim, dim]`` (gram.py:58-69)."""
    dim = int(math.sqrt(tensor.numel()))
    return tensor.reshape(dim, dim)


def partial_contract(tensor: Tensor, other: Tensor, start_dims: Tuple[int, int], out: Tensor = None,
                     alpha: float = 1.0, beta: float = 0.0) -> Tensor:
    """Contract all dims from ``start_dims`` on; result has ``sum(start_dims)`` dims.

    ``(2, 2)`` with ``other is tensor`` is the Gram build K1 (SYRK), ``(2, 1)`` is ``V^T g`` (K2),
    ``(1, 1)`` the gradient Gram.  ``out``/``alpha``/``beta`` (extension of the reference
    signature) fuse the ``gram += gram_p`` accumulation of gram.py:104-116 into the kernel.
    """
    o1, o2 = start_dims
    f1, f2 = tensor.dim() - o1, other.dim() - o2
    if f1 != f2:
        raise ValueError("Trailing dimensions don't match.")
    get_letters(o1 + o2 + f1)
    if tuple(tensor.shape[o1:]) != tuple(other.shape[o2:]):
        raise RuntimeError(
            f"trailing shapes differ: {tuple(tensor.shape[o1:])} vs {tuple(other.shape[o2:])}"
        )
    lead1, lead2 = tuple(tensor.shape[:o1]), tuple(other.shape[:o2])
    A = tensor.detach().reshape(_numel(lead1), -1)
    out2d = None if out is None else out.view(_numel(lead1), _numel(lead2))
    if other is tensor and o1 == o2:
        res = kernels.gram_syrk(A, out=out2d, alpha=alpha, beta=beta)
    else:
        B = other.detach().reshape(_numel(lead2), -1)
        res = kernels.gemm_nt(A, B, out=out2d, alpha=alpha, beta=beta)
    return res.view(*lead1, *lead2)


def pairwise_dot(tensor: Tensor, start_dim: int = 1, flatten: bool = True, out: Tensor = None,
                 beta: float = 0.0) -> Tensor:
    """Pairwise scalar products of the slices selected by ``start_dim`` (gram.py:9-35)."""
    result = partial_contract(tensor, tensor, (start_dim, start_dim), out=out, beta=beta)
    return reshape_as_square(result) if flatten else result


def compute_gram_mat(parameters: Iterable, savefield: str, start_dim: int, flatten: bool = True) -> Tensor:
    """Sum of per-parameter Grams of ``p.<savefield>`` (gram.py:72-116), accumulated in-kernel."""
    gram = None
    for p in parameters:
        if gram is None:
            gram = pairwise_dot(getattr(p, savefield), start_dim=start_dim, flatten=False)
        else:
            pairwise_dot(getattr(p, savefield), start_dim=start_dim, flatten=False, out=gram, beta=1.0)
    if gram is not None and flatten:
        gram = reshape_as_square(gram)
    return gram


def mVp(V_t: Tensor, mat: Tensor, start_dim: int) -> Tensor:
    """``V^T @ mat``: ``[F, *param] x [*start, *param] -> [F, *start]`` (gram.py:182-203; K9)."""
    lead = tuple(V_t.shape[:start_dim])
    A = mat.detach().reshape(mat.shape[0], -1)
    B = V_t.detach().reshape(_numel(lead), -1)
    return kernels.gemm_nt(A, B).view(mat.shape[0], *lead)


def sqrt_gram_mat_prod(mat: Tensor, parameters: Iterable, savefield: str, start_dim: int, concat: bool = False):
    """Multiply the columns of ``mat`` with the Gram square root ``U`` (gram.py:119-179)."""
    if mat.dim() != 2:
        raise NotImplementedError("Can only multiply with matrices")
    result = []
    for p in parameters:
        sqrt = getattr(p, savefield).detach()
        lead = _numel(sqrt.shape[:start_dim])
        U_t = sqrt.reshape(lead, -1)  # [n, P]
        # out[P, J] = U_t^T @ mat
        res = kernels.gemm_tn(U_t, mat.detach())
        result.append(res.view(*sqrt.shape[start_dim:], mat.shape[1]))
    if concat:
        result = cat([r.flatten(end_dim=r.dim() - 2) for r in result])
    return result


def split_list(sequence, lengths):
    """Consecutive sub-lists of the given lengths (gram.py:235-256)."""
    if len(sequence) != sum(lengths):
        raise ValueError("Sub-list lengths don't sum to length of the full list.")
    start = 0
    for length in lengths:
        yield sequence[start : start + length]
        start += length
