"""CPU ORACLE for the ViViT low-rank GGN path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain torch-on-CPU restatement of the reference algorithm (f-dangel/vivit @ v1), every
function citing the reference file:line (relative to /root/reference) it follows.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg import this
module, and only as the checker or as the timed CPU baseline; nothing under ``vivit_amd/``
imports it.

Pinning status: PINNED.  ``tests/golden/make_golden.py`` imports the reference's own
``vivit/utils/{gram,ggn,eig,hooks,checks}.py`` and, through a stub ``backpack`` package plus a
``Tensor.symeig -> torch.linalg.eigh`` shim (the method was removed from torch), its four
Computation classes, runs them on seeded hand-made factors and commits inputs + outputs under
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks this oracle against every one of
those vectors.  What the reference cannot provide here is BackPACK's factor materialisation
(backpack-for-pytorch >=1.5.0,<2.0.0, setup.cfg:36, absent from this image): the factor oracle at
the bottom of this file restates its published definition (per-sample Jacobian times the
loss-Hessian square root) and is pinned through the reference tests' own properties
(``V V^T == GGN``; test/extensions/secondorder/vivit/test_vivit_ggn.py:22-76).
"""
import math
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
from torch import Tensor, einsum

# ---------------------------------------------------------------------------------------------
# vivit/utils/gram.py
# ---------------------------------------------------------------------------------------------


def _letters(num: int) -> str:
    """vivit/utils/gram.py:38-55 (at most 26 einsum letters, ValueError beyond)."""
    if num > 26:
        raise ValueError(f"Requested too many letters {num}>26")
    return "".join(chr(ord("a") + i) for i in range(num))


def partial_contract(tensor: Tensor, other: Tensor, start_dims: Tuple[int, int]) -> Tensor:
    """Contract all trailing dims. vivit/utils/gram.py:206-232 (K1 with (2,2), K2 with (2,1))."""
    o1, o2 = start_dims
    f1, f2 = tensor.dim() - o1, other.dim() - o2
    if f1 != f2:
        raise ValueError("Trailing dimensions don't match.")
    let = _letters(o1 + o2 + f1)
    a, b, s = let[:o1], let[o1 : o1 + o2], let[o1 + o2 :]
    return einsum(f"{a}{s},{b}{s}->{a}{b}", tensor, other)


def reshape_as_square(tensor: Tensor) -> Tensor:
    """vivit/utils/gram.py:58-69."""
    dim = int(math.sqrt(tensor.numel()))
    return tensor.reshape(dim, dim)


def pairwise_dot(tensor: Tensor, start_dim: int = 1, flatten: bool = True) -> Tensor:
    """vivit/utils/gram.py:9-35."""
    out = partial_contract(tensor, tensor, (start_dim, start_dim))
    return reshape_as_square(out) if flatten else out


def compute_gram_mat(factors: Sequence[Tensor], start_dim: int, flatten: bool = True) -> Tensor:
    """Sum of per-parameter Grams, ``gram += gram_p``. vivit/utils/gram.py:72-116."""
    gram = None
    for f in factors:
        g = pairwise_dot(f, start_dim=start_dim, flatten=flatten)
        gram = g if gram is None else gram + g
    return gram


def mVp(V_t: Tensor, mat: Tensor, start_dim: int) -> Tensor:
    """``V^T @ mat``: [F,*param] x [*start,*param] -> [F,*start]. vivit/utils/gram.py:182-203."""
    let = _letters(V_t.dim() + 1)
    free, out, s = let[0], let[1 : start_dim + 1], let[start_dim + 1 :]
    return einsum(f"{free}{s},{out}{s}->{free}{out}", mat, V_t)


# ---------------------------------------------------------------------------------------------
# vivit/utils/ggn.py
# ---------------------------------------------------------------------------------------------


def Vmp(V_t: Tensor, mat: Tensor, start_dim: int) -> Tensor:
    """``V @ mat``: [F,*start] x [*start,*param] -> [F,*param]. vivit/utils/ggn.py:94-115."""
    let = _letters(V_t.dim() + 1)
    free, s, out = let[0], let[1 : start_dim + 1], let[start_dim + 1 :]
    return einsum(f"{free}{s},{s}{out}->{free}{out}", mat, V_t)


# ---------------------------------------------------------------------------------------------
# vivit/extensions/secondorder/vivit/linear.py:41-81 (factorised Linear-weight closures)
# ---------------------------------------------------------------------------------------------


def linear_weight_gram(s: Tensor, z: Tensor) -> Tensor:
    """[C,N,C,N] Gram of a Linear weight from s=[C,N,out], z=[N,in]. linear.py:66-75."""
    return einsum("nm,cndm->cndm", pairwise_dot(z, 1, False), pairwise_dot(s, 2, False))


def linear_weight_V_mat_prod(s: Tensor, z: Tensor, mat: Tensor) -> Tensor:
    """linear.py:44-53."""
    return einsum("cno,vcn,ni->voi", s, mat, z)


def linear_weight_V_t_mat_prod(s: Tensor, z: Tensor, mat: Tensor) -> Tensor:
    """linear.py:55-64."""
    return einsum("cno,voi,ni->vcn", s, mat, z)


# ---------------------------------------------------------------------------------------------
# vivit/utils/eig.py  (Tensor.symeig is gone from torch; its successor is torch.linalg.eigh)
# ---------------------------------------------------------------------------------------------


def tensor_symeig(mat: Tensor, eigenvectors: bool = False, upper: bool = True):
    """Old ``Tensor.symeig`` semantics: ascending eigenvalues, column eigenvectors (eig.py:24-26)."""
    uplo = "U" if upper else "L"
    if eigenvectors:
        return torch.linalg.eigh(mat, UPLO=uplo)
    return torch.linalg.eigvalsh(mat, UPLO=uplo), mat.new_empty(0)


def shift_diag(mat: Tensor, shift: float, inplace: bool = False) -> Tensor:
    """vivit/utils/eig.py:51-74."""
    if shift == 0.0:
        return mat
    out = mat if inplace else mat.clone()
    k = min(mat.shape)
    out[range(k), range(k)] += shift
    return out


def symeig_psd(mat, eigenvectors=False, upper=True, shift=0.0, shift_inplace=False):
    """vivit/utils/eig.py:6-48."""
    if mat.dim() != 2:
        raise ValueError(f"Input must have dimension 2. Got {mat.dim()}.")
    mat = shift_diag(mat, shift, inplace=shift_inplace)
    try:
        evals, evecs = tensor_symeig(mat, eigenvectors=eigenvectors, upper=upper)
    except RuntimeError as e:
        raise RuntimeError(f"Tensor contains NaNs: {torch.isnan(mat).any()}") from e
    if shift_inplace:
        mat = shift_diag(mat, -shift, inplace=True)
    evals -= shift
    return evals, evecs


def remove_zero_evals(evals, evecs, atol=1e-7, rtol=1e-5):
    """vivit/utils/eig.py:111-134."""
    nz = torch.isclose(evals, torch.zeros_like(evals), rtol=rtol, atol=atol).logical_not()
    evals = evals[nz]
    if evecs.numel() != 0:
        evecs = evecs[:, nz]
    return evals, evecs


def symeig(mat, eigenvectors=False, upper=True, atol=1e-7, rtol=1e-5):
    """vivit/utils/eig.py:77-108."""
    if mat.dim() != 2:
        raise ValueError("Input must be of dimension 2")
    evals, evecs = tensor_symeig(mat, eigenvectors=eigenvectors, upper=upper)
    return remove_zero_evals(evals, evecs, atol=atol, rtol=rtol)


# ---------------------------------------------------------------------------------------------
# vivit/linalg/utils.py:67-76
# ---------------------------------------------------------------------------------------------


def normalize(tensors: List[Tensor]) -> List[Tensor]:
    inv_norm = 1 / sum(einsum("i...->i", t**2) for t in tensors).sqrt()
    return [einsum("i,i...->i...", inv_norm, t) for t in tensors]


# ---------------------------------------------------------------------------------------------
# Group hooks of the four Computation classes, as pure functions of the factors.
# ---------------------------------------------------------------------------------------------


def eigvalsh_group(grams: Sequence[Tensor], batch_size: int, subsampling: Optional[List[int]]) -> Tensor:
    """vivit/linalg/eigvalsh.py:170-183 (accumulate) + :201-225 (group hook)."""
    acc = None
    for g in grams:
        acc = g if acc is None else acc + g
    gram = reshape_as_square(acc)
    if subsampling is not None:
        gram = gram * (batch_size / len(subsampling))
    evals, _ = tensor_symeig(gram, eigenvectors=False)
    return evals


def eigh_group(
    grams: Sequence[Tensor],
    V_mat_prods: Sequence[Callable[[Tensor], Tensor]],
    criterion: Callable[[Tensor], List[int]],
    batch_size: int,
    subsampling: Optional[List[int]],
) -> Tuple[Tensor, List[Tensor]]:
    """vivit/linalg/eigh.py:222-275. ``grams[p]``: [C,N,C,N]; ``V_mat_prods[p](mat[K,C,N]) -> [K,*p]``."""
    gram = 0.0
    for g in grams:
        gram = gram + g
    if subsampling is not None:
        gram = gram * (batch_size / len(subsampling))
    evals, evecs = tensor_symeig(reshape_as_square(gram), eigenvectors=True)
    keep = criterion(evals)
    evals, evecs = evals[keep], evecs[:, keep]
    evecs = evecs.transpose(0, 1).reshape(-1, *gram.shape[:2])
    group_evecs = [f(evecs) for f in V_mat_prods]
    return evals, normalize(group_evecs)


def _directional(V_list, g_list, criterion, batch_size):
    """Shared part of directional_damped_newton.py:304-351 / directional_derivatives.py:281-325."""
    V_t_V = None
    V_t_g = None
    for V, g in zip(V_list, g_list):
        a = partial_contract(V, V, (2, 2))  # directional_damped_newton.py:254
        b = partial_contract(V, g, (2, 1))  # :255
        V_t_V = a if V_t_V is None else V_t_V + a  # _accumulate :402-405
        V_t_g = b if V_t_g is None else V_t_g + b
    N = batch_size
    N_ggn = V_t_V.shape[1]
    V_correction = math.sqrt(N / N_ggn)  # :308
    gram = V_correction**2 * V_t_V  # :309
    C = gram.shape[0]
    evals, evecs = tensor_symeig(reshape_as_square(gram), eigenvectors=True)  # :315
    keep = criterion(evals)  # :317
    evals, evecs = evals[keep], evecs[:, keep]  # :321
    V_t_g_n = V_correction * N * V_t_g.flatten(start_dim=0, end_dim=1)  # :327-331
    gammas = einsum("in,id->nd", V_t_g_n, evecs) / evals.sqrt()  # :342
    V_n_T_V_e_d = math.sqrt(N_ggn) * einsum("cni,id->cnd", gram.flatten(start_dim=2), evecs)  # :348-350
    lambdas = (V_n_T_V_e_d**2).sum(0) / evals  # :351
    return evals, evecs, gammas, lambdas, V_correction, C, N_ggn


def directional_derivatives_group(V_list, g_list, criterion, batch_size):
    """vivit/optim/directional_derivatives.py:255-325 -> (gammas [N_grad,K], lambdas [N_ggn,K])."""
    _, _, gammas, lambdas, _, _, _ = _directional(V_list, g_list, criterion, batch_size)
    return gammas, lambdas


def damped_newton_group(V_list, g_list, criterion, damping, batch_size) -> List[Tensor]:
    """vivit/optim/directional_damped_newton.py:263-379 -> list of step tensors."""
    evals, evecs, gammas, lambdas, V_correction, C, N_ggn = _directional(V_list, g_list, criterion, batch_size)
    coefficients = -gammas.mean(0) / (lambdas.mean(0) + damping(evals, evecs, gammas, lambdas)) / evals.sqrt()  # :354-359
    v = einsum("id,d->i", evecs, coefficients) * V_correction  # :362-366
    v = v.reshape(C, N_ggn)  # :369
    return [einsum("cn,cn...->...", v, V) for V in V_list]  # :370-373


def gram_sqrt_ggn(V_list: Sequence[Tensor]) -> Tensor:
    """GramSqrtGGN{Exact,MC}: accumulated [NC x NC] Gram.
    vivit/extensions/secondorder/sqrt_ggn/gram_sqrt_ggn.py:38-74."""
    gram = None
    for V in V_list:
        g = pairwise_dot(V, start_dim=2).detach()
        gram = g if gram is None else gram + g
    return gram


def gram_batch_grad(g_list: Sequence[Tensor], center: bool) -> Tensor:
    """GramBatchGrad / CenteredGramBatchGrad.
    vivit/extensions/firstorder/batch_grad/gram_batch_grad.py:77-117."""
    gram = None
    for g in g_list:
        if center:
            g = g - g.mean(0)
        gg = pairwise_dot(g, start_dim=1).detach()
        gram = gg if gram is None else gram + gg
    return gram


# ---------------------------------------------------------------------------------------------
# Factor oracle: what BackPACK's SqrtGGN{Exact,MC} / BatchGrad / ViViTGGN* materialise.
# Third-party (backpack-for-pytorch >=1.5.0,<2.0.0; not in /root/reference).  Restated from the
# published definition and anchored on the reference's call sites
# (vivit/extensions/secondorder/vivit/base.py:84-92, linear.py:41-42) and property tests.
# ---------------------------------------------------------------------------------------------


def loss_hessian_sqrt_exact(output: Tensor, loss: str) -> Tensor:
    """S[v, n, c] with sum_v S[v,n,:] S[v,n,:]^T = Hessian of the MEAN loss wrt output[n]."""
    N, C = output.shape
    if loss == "ce":
        p = output.softmax(dim=1)
        sq = p.sqrt()
        # H_n = diag(p) - p p^T = S S^T with S[:, v] = sqrt(p_v) (e_v - p)   (per sample)
        S = einsum("nv,vc->vnc", sq, torch.eye(C, dtype=output.dtype)) - einsum("nv,nc->vnc", sq, p)
        return S / math.sqrt(N)
    if loss == "mse":
        # mean over N*C elements of (f - y)^2: H_n = 2/(N C) I
        S = torch.eye(C, dtype=output.dtype).unsqueeze(1).expand(C, N, C).clone()
        return S * math.sqrt(2.0 / (N * C))
    raise ValueError(loss)


def loss_hessian_sqrt_mc(output: Tensor, onehots: Tensor) -> Tensor:
    """MC factor for cross-entropy: S[m,n,:] = (p_n - onehot(y_mn)) / sqrt(M N), y ~ Cat(p_n).
    ``onehots``: [M, N, C] externally supplied samples (parity needs identical samples)."""
    M, N, C = onehots.shape
    p = output.softmax(dim=1)
    return (p.unsqueeze(0) - onehots) / math.sqrt(M * N)


def loss_hessian_sqrt_mc_mse(eps: Tensor) -> Tensor:
    """MC factor of the MEAN squared error (vivit/extensions/secondorder/vivit/__init__.py:84-86 -> BackPACK
    ``SqrtGGNMSELoss`` with ``LossHessianStrategy.SAMPLING``, :155-181): the Hessian w.r.t. output[n] is
    ``2 / (N C) I`` whatever the output is, so its sampled square root is a scaled standard-normal draw,
    ``S[m, n, :] = sqrt(2 / (M N C)) eps[m, n, :]`` with ``E[sum_m S S^T] = 2 / (N C) I``.
    ``eps``: [M, N, C] externally supplied N(0, 1) samples (parity needs identical samples)."""
    M, N, C = eps.shape
    return eps * math.sqrt(2.0 / (M * N * C))


def per_sample_jacobians(model: torch.nn.Module, X: Tensor) -> List[Tensor]:
    """J[p]: [N, C, *param.shape] by brute-force autograd (one backward per (n, c))."""
    params = [p for p in model.parameters() if p.requires_grad]
    out = model(X)
    N, C = out.shape[0], out[0].numel()
    out = out.reshape(N, C)
    jac = [torch.zeros(N, C, *p.shape, dtype=X.dtype) for p in params]
    for n in range(N):
        for c in range(C):
            grads = torch.autograd.grad(out[n, c], params, retain_graph=True, allow_unused=True)
            for j, g in zip(jac, grads):
                if g is not None:
                    j[n, c] = g
    return jac


def sqrt_ggn_factors(model, X, S: Tensor, subsampling: Optional[List[int]] = None) -> List[Tensor]:
    """V_t[p]: [V, N, *param.shape] = sum_c S[v,n,c] J[n,c,...] (BackPACK ``sqrt_ggn_exact`` /
    ``sqrt_ggn_mc`` layout, vivit/utils/ggn.py:14-19).  ``S`` is the loss-Hessian square root of
    the FULL batch (carrying 1/sqrt(N)); sub-sampling slices the sample axis afterwards."""
    jac = per_sample_jacobians(model, X)
    if subsampling is not None:
        S = S[:, subsampling]
        jac = [j[subsampling] for j in jac]
    return [einsum("vnc,nc...->vn...", S, j) for j in jac]


def batch_grads(model, X, y, lossfunc, subsampling: Optional[List[int]] = None) -> List[Tensor]:
    """BackPACK ``grad_batch``: [N, *param.shape], per-sample gradients of the MEAN loss (carry 1/N)."""
    params = [p for p in model.parameters() if p.requires_grad]
    N = X.shape[0]
    idx = range(N) if subsampling is None else subsampling
    out = []
    for n in idx:
        loss_n = lossfunc(model(X[n : n + 1]), y[n : n + 1]) / N
        out.append(torch.autograd.grad(loss_n, params))
    return [torch.stack([g[i] for g in out]) for i in range(len(params))]


def dense_ggn(model, X, loss: str) -> Tensor:
    """Dense GGN of the mean loss, sum_n J_n^T H_n J_n, for property checks."""
    jac = per_sample_jacobians(model, X)
    N, C = jac[0].shape[:2]
    J = torch.cat([j.reshape(N, C, -1) for j in jac], dim=2)
    out = model(X).reshape(N, C)
    S = loss_hessian_sqrt_exact(out.detach(), loss)
    H = einsum("vnc,vnd->ncd", S, S)
    return einsum("ncp,ncd,ndq->pq", J, H, J)
