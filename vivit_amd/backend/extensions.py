"""Extensions of the stand-in backend: BatchGrad, SqrtGGN{Exact,MC}, ViViTGGN{Exact,MC}.

Data contract (identical to BackPACK's):
  ``param.grad_batch``       [N_grad, *param.shape]      carries 1/N for ``reduction='mean'``
  ``param.sqrt_ggn_exact``   [C, N_ggn, *param.shape]    carries 1/sqrt(N)
  ``param.sqrt_ggn_mc``      [M, N_ggn, *param.shape]    carries 1/sqrt(M N)
  ``param.vivit_ggn_exact|mc``  dict with closures ``gram_mat() -> [C,N,C,N]``,
      ``V_mat_prod(mat[F,C,N]) -> [F,*param]``, ``V_t_mat_prod(mat[F,*param]) -> [F,C,N]``
      (vivit/extensions/secondorder/vivit/base.py:126-130); Linear weights keep the factorised
      form ``s=[C,N,out], z=[N,in]`` (vivit/extensions/secondorder/vivit/linear.py:41-81).
"""
import math
from typing import List, Optional

import torch
import torch.nn.functional as F
from torch import Tensor, nn

from vivit_amd import _lib, kernels
from vivit_amd.backend.custom_module import ActiveIdentity, Pad, ScaleModule, Slicing, SumModule
from vivit_amd.utils.ggn import Vmp
from vivit_amd.utils.gram import mVp, pairwise_dot

DP_ROWS_MAX_COLUMNS = 4096  # data parallel: narrower materialised factors are all-gathered (block rows), wider ones
                            # go through the all-to-all to parameter shards (vivit_amd.distributed.BatchShardedGram)
_LOSSES = (nn.CrossEntropyLoss, nn.MSELoss)
_BATCHNORM = (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d)
_CONVS = (nn.Conv1d, nn.Conv2d, nn.Conv3d, nn.ConvTranspose1d, nn.ConvTranspose2d, nn.ConvTranspose3d)


def subsample(tensor: Tensor, dim: int = 0, subsampling: Optional[List[int]] = None) -> Tensor:
    """``backpack.utils.subsampling.subsample`` (used at linear.py:42)."""
    if subsampling is None:
        return tensor
    idx = torch.as_tensor(subsampling, device=tensor.device, dtype=torch.long)
    return tensor.index_select(dim, idx)


class LinearFactor:
    """Factorised quantity of a Linear weight, ``T[..., n, o, i] = s[..., n, o] * z[n, i]``, never materialised.

    ``s: [C, N, out]`` for a sqrt-GGN factor (``V_t`` of vivit/extensions/secondorder/vivit/linear.py:41-42) or
    ``[N, out]`` for per-sample gradients (``grad_batch[n] = delta_n z_n^T``); ``z: [N, in]``.  Stored under the
    savefield in place of the tensor when an extension is created with ``factorised=True``: the only representation
    that exists at BASELINE config 5 (the tensor would be 2.7 TB).  ``vivit_amd.optim`` contracts it with two small
    GEMMs and the fused Hadamard kernel."""

    def __init__(self, s: Tensor, z: Tensor):
        self.s, self.z = s.contiguous(), z.contiguous()

    @property
    def shape(self):
        return tuple(self.s.shape) + (self.z.shape[1],)

    def dim(self):
        return self.s.dim() + 1

    def detach(self):
        return self

    def materialise(self) -> Tensor:
        """The explicit tensor (small problems / mixed representations only)."""
        if self.s.is_cuda and self.s.dtype == torch.float32:   # the HIP store-stream kernel of the weight rule
            if self.s.dim() == 2:
                return kernels.linear_weight_mjp(self.s.unsqueeze(0), self.z)[0]
            return kernels.linear_weight_mjp(self.s, self.z)
        if self.s.dim() == 2:
            return torch.einsum("no,ni->noi", self.s, self.z)
        return torch.einsum("cno,ni->cnoi", self.s, self.z)


class _Extension:
    savefield = None
    # Stream contract (backend/engine.py): an extension whose ``apply`` reads ``g_out`` (autograd's gradient) keeps
    # ``uses_grad = True`` -- the extensions' stream then waits for the backward pass at every hook and the gradient is
    # recorded on it.  Only extensions that provably ignore ``g_out`` (the sqrt-GGN family) may set it to False.
    uses_grad = True

    def __init__(self, subsampling: Optional[List[int]] = None):
        self._subsampling = subsampling

    def get_subsampling(self):
        return self._subsampling

    def apply(self, ctx, module, g_out):
        raise NotImplementedError


def _own_params(module):
    return [(name, p) for name, p in module._parameters.items() if p is not None and p.requires_grad]


def _spatial_sum(t: Tensor, keep: int) -> Tensor:
    """Sum all dims after the first ``keep`` ones (HIP: one wave per row, fixed order)."""
    if t.dim() <= keep:
        return t
    if t.is_cuda:
        lead = t.shape[:keep]
        return kernels.row_dot(t.reshape(-1, t[(0,) * keep].numel())).view(lead)
    return t.flatten(start_dim=keep).sum(keep)


def _spatial_sum3(M: Tensor) -> Tensor:
    """``_spatial_sum(M, 3)`` remembered ON the factor tensor: BatchNorm's bias rule and its weight rule both need it (one pass
    over M instead of two; the memo lives and dies with the tensor object)."""
    memo = getattr(M, "_vivit_ssum3", None)
    if memo is None:
        memo = _spatial_sum(M, 3)
        try:
            M._vivit_ssum3 = memo
        except AttributeError:   # (tensor subclasses without a __dict__)
            pass
    return memo


def _bn_constants(module):
    """``(rstd, scale)`` of a BatchNorm in eval mode, recomputed at every use (two tiny launches): the three rules of a module
    share them through the memo :func:`_bn_eval_rules` keeps on the factor tensor, i.e. for one backward pass.  Nothing is
    remembered on the module: writes through ``.data`` (``bn.weight.data = ...``, as the reference's tests re-initialise,
    test/utils.py:111) leave ``_version`` and ``data_ptr`` unchanged, so no key on the module can tell stale constants."""
    rstd = torch.rsqrt(module.running_var + module.eps)
    scale = rstd * module.weight.detach() if module.weight is not None else rstd
    return rstd, scale


def _bn_scale(module) -> Tensor:
    return _bn_constants(module)[1]


def _bn_eval_rules(module, M: Tensor, x: Tensor):
    """``(M scale_c, (sum_l M x - mean_c sum_l M) rstd_c, sum_l M)`` of a BatchNorm in eval mode from ONE pass over the factor ``M [V, N, C, *spatial]``
    (``vivit_bn_eval_rules_f32``), remembered on the tensor: the weight rule, the bias rule and the input rule of the module
    are three calls on the same ``M`` (round 4: three row reductions and one scaling pass, four reads of ``M``)."""
    memo = getattr(M, "_vivit_bn_rules", None)
    if memo is None or memo[0] is not module:
        Mc = M if M.dim() > 3 else M.unsqueeze(-1)
        xc = x if x.dim() > 2 else x.unsqueeze(-1)
        rstd, scale = _bn_constants(module)
        out, mx, ms = kernels.bn_eval_rules(Mc, xc, scale, module.running_mean, rstd)   # mx: the finished weight rule
        memo = (module, out.view(M.shape), mx, ms)
        try:
            M._vivit_bn_rules = memo
        except AttributeError:   # (tensor subclasses without a __dict__)
            pass
    return memo[1:]


def _param_factor(module, name: str, M: Tensor, x: Tensor) -> Tensor:
    """``param_mjp(..., sum_batch=False)``: ``M`` [V, N, *out] -> [V, N, *param.shape]."""
    if isinstance(module, nn.Linear):
        if name == "bias":
            return M if M.dim() == 3 else M.flatten(2, -2).sum(2)
        if M.dim() == 3:
            return kernels.linear_weight_mjp(M, x)           # "vno,ni->vnoi" (HIP store-stream kernel)
        Ma, xa = M.flatten(2, -2), x.flatten(1, -2)          # [V, N, A, O], [N, A, I]  (linear.py:52-81: extra dims are summed)
        if M.is_cuda and M.dtype == torch.float32:
            # a 1 x 1 convolution over the A positions: the HIP weight rule of the convolution, "vnoa,nia->vnoi"
            out = kernels.conv2d_weight_mjp(Ma.transpose(2, 3).unsqueeze(3).contiguous(), xa.transpose(1, 2).unsqueeze(2).contiguous(),
                                            (1, 1), (1, 1), (0, 0), (1, 1))
            return out.reshape(out.shape[:4])
        return torch.einsum("vnao,nai->vnoi", Ma, xa)
    if isinstance(module, _CONVS):
        if name == "bias":
            return _spatial_sum(M, 3)
        if (M.is_cuda and M.dtype == torch.float32 and isinstance(module.padding, tuple) and module.padding_mode == "zeros"):
            try:
                if isinstance(module, (nn.Conv3d, nn.ConvTranspose3d)):
                    return _hip_conv3d_weight_factor(module, M, x)
                return _hip_conv_weight_factor(module, M, x)
            except _lib.VivitHipError as exc:  # shapes outside the kernel's launch limits: the torch rule below
                if exc.status != _lib.VIVIT_E_UNSUPPORTED:
                    raise
        return _conv_weight_factor(module, M, x)
    if isinstance(module, _BATCHNORM):
        if module.training:
            raise NotImplementedError("BatchNorm must be in eval mode (as in the reference tests)")
        if M.is_cuda and M.dtype == torch.float32:
            # sum_l M xhat = (sum_l M x - mean_c sum_l M) rstd_c: both row reductions (and the input rule's scaling) come out of
            # one pass over M; the rest is [V, N, C]-sized
            _, Mw, Ms = _bn_eval_rules(module, M, x)
            return Ms if name == "bias" else Mw
        if name == "bias":
            return _spatial_sum3(M)
        rstd = torch.rsqrt(module.running_var + module.eps)
        shape = [1, -1] + [1] * (x.dim() - 2)
        xhat = (x - module.running_mean.view(shape)) * rstd.view(shape)
        return _spatial_sum(M * xhat.unsqueeze(0), 3)
    raise NotImplementedError(f"no parameter rule for {type(module).__name__}")


_CONV_FN = {nn.Conv1d: F.conv1d, nn.Conv2d: F.conv2d, nn.Conv3d: F.conv3d, nn.ConvTranspose1d: F.conv_transpose1d,
            nn.ConvTranspose2d: F.conv_transpose2d, nn.ConvTranspose3d: F.conv_transpose3d}


def _hip_conv_weight_factor(module, M: Tensor, x: Tensor) -> Tensor:
    """Weight rule of Conv1d/2d and ConvTranspose1d/2d (any ``groups``, zero padding) on ONE HIP kernel
    (csrc/jacobians.hip: unfold + "vnol,nkl->vnok", patch values gathered on the fly, no im2col buffer).

    * a 1-D convolution is the 2-D one with a single row;
    * ``groups > 1``: one launch per group on the group's channel slices (the weight's leading axis is the groups'
      output channels one after the other -- convnd.py:9-30 through BackPACK's grouped unfold);
    * a transposed convolution y = conv_transpose(x, W) is the adjoint of a convolution with the same stride / padding /
      dilation, so dL/dW[ci, co, k] = sum_q x[ci, q] M[co, q s - p + k d] is the SAME contraction with the roles swapped:
      ``M`` is the image the patches are gathered from, ``x`` the coefficient (convtransposend.py:9-30).  The kernel
      gathers from the operand without a slice axis, so the slice axis of ``M`` is folded into the batch and ``x`` is
      repeated over it."""
    one_d = isinstance(module, (nn.Conv1d, nn.ConvTranspose1d))
    transposed = isinstance(module, (nn.ConvTranspose1d, nn.ConvTranspose2d))
    if one_d:
        M, x = M.unsqueeze(3), x.unsqueeze(2)
        ks, st, pd, dl = (1, module.kernel_size[0]), (1, module.stride[0]), (0, module.padding[0]), (1, module.dilation[0])
    else:
        ks, st, pd, dl = module.kernel_size, module.stride, module.padding, module.dilation
    V, N, G = M.shape[0], M.shape[1], module.groups
    if transposed:
        coef = x.unsqueeze(0).expand(V, *x.shape).reshape(1, V * N, *x.shape[1:])     # [1, V N, Cin, Hq, Wq]
        image = M.reshape(V * N, *M.shape[2:])                                         # [V N, Cout, H', W']
    else:
        coef, image = M, x
    Cc, Ci = coef.shape[2] // G, image.shape[1] // G
    parts = []
    for g in range(G):
        c = coef if G == 1 else coef[:, :, g * Cc:(g + 1) * Cc]
        im = image if G == 1 else image[:, g * Ci:(g + 1) * Ci]
        parts.append(kernels.conv2d_weight_mjp(c.contiguous(), im.contiguous(), ks, st, pd, dl))
    out = parts[0] if G == 1 else torch.cat(parts, 2)     # [.., .., coefficient channels, image channels / G, kh, kw]
    if transposed:
        out = out.view(V, N, *out.shape[2:])
    return out.squeeze(4) if one_d else out


def _triple(v):
    return (v, v, v) if isinstance(v, int) else tuple(v)


def _depth_range(n_out: int, n_in: int, s: int, p: int, off: int):
    """Positions ``q`` in ``[0, n_out)`` whose partner ``q s - p + off`` lies in ``[0, n_in)``: ``(q_lo, q_hi, first partner)``
    (``q_hi < q_lo``: none)."""
    q_lo = max(0, -((off - p) // s))            # ceil((p - off) / s)
    q_hi = min(n_out - 1, (n_in - 1 + p - off) // s)
    return q_lo, q_hi, q_lo * s - p + off


def _conv3d_weight_contract(coef: Tensor, image: Tensor, ks, st, pd, dl) -> Tensor:
    """``out[v, n, c, i, kd, kh, kw] = sum_q coef[v, n, c, qd, qh, qw] image[n, i, qd sd - pd + kd dd, qh sh - ph + kh dh,
    qw sw - pw + kw dw]`` -- the weight rule of a three-dimensional convolution -- on the TWO-dimensional HIP kernel, one
    launch per depth tap ``kd``: the depth slices are stacked along the height of one tall image.  Behind each coefficient
    slice stand ``ceil((KH - 1) dh / sh)`` zero rows, so that a window that starts in a slice's last rows and runs into the
    next slice meets a zero coefficient; the image slices carry their height padding explicitly at the matching pitch.
    ``coef: [V, N, C, QD, QH, QW]``, ``image: [N, I, D, H, W]`` -> ``[V, N, C, I, KD, KH, KW]``."""
    Vc, Nc, Cc, QD, QH, QW = coef.shape
    _, Ci, D, H, W = image.shape
    (KD, KH, KW), (sd, sh, sw), (pdd, ph, pw), (dd, dh, dw) = ks, st, pd, dl
    extra = -(-((KH - 1) * dh) // sh)
    QHp = QH + extra
    Hp = QHp * sh                      # pitch of an image slice: window row qh sh + kh dh of slice qd is row qd Hp + ...
    tail = (KH - 1) * dh               # rows behind the last slice that windows of (zero) coefficient rows still address
    cs = coef.new_zeros((Vc, Nc, Cc, QD, QHp, QW))
    cs[..., :QH, :] = coef
    cs = cs.view(Vc, Nc, Cc, QD * QHp, QW)
    hrows = min(H, Hp - ph)            # image rows that fall inside a slice (the rest is never read by a live window)
    outs = []
    for kd in range(KD):
        xs = image.new_zeros((Nc, Ci, QD * Hp + tail, W))
        q_lo, q_hi, d_lo = _depth_range(QD, D, sd, pdd, kd * dd)
        if q_hi >= q_lo and hrows > 0:
            nq = q_hi - q_lo + 1
            xv = xs[:, :, :QD * Hp].view(Nc, Ci, QD, Hp, W)
            xv[:, :, q_lo:q_hi + 1, ph:ph + hrows] = image[:, :, d_lo:d_lo + (nq - 1) * sd + 1:sd, :hrows]
        outs.append(kernels.conv2d_weight_mjp(cs, xs, (KH, KW), (sh, sw), (0, pw), (dh, dw)))
    return torch.stack(outs, 4)


def _hip_conv3d_weight_factor(module, M: Tensor, x: Tensor) -> Tensor:
    """Weight rule of Conv3d / ConvTranspose3d (any ``groups``, zero padding; convnd.py:25-30, convtransposend.py:25-30)
    on the HIP kernel of the two-dimensional rule (:func:`_conv3d_weight_contract`).  As in two dimensions, the transposed
    convolution's rule is the same contraction with the roles of factor and input swapped, and a grouped layer is one
    contraction per group on its channel slices."""
    transposed = isinstance(module, nn.ConvTranspose3d)
    ks, st, pd, dl = module.kernel_size, module.stride, module.padding, module.dilation
    V, N, G = M.shape[0], M.shape[1], module.groups
    if transposed:
        coef = x.unsqueeze(0).expand(V, *x.shape).reshape(1, V * N, *x.shape[1:])     # [1, V N, Cin, Dq, Hq, Wq]
        image = M.reshape(V * N, *M.shape[2:])                                         # [V N, Cout, D', H', W']
    else:
        coef, image = M, x
    Cc, Ci = coef.shape[2] // G, image.shape[1] // G
    parts = []
    for g in range(G):
        c = coef if G == 1 else coef[:, :, g * Cc:(g + 1) * Cc]
        im = image if G == 1 else image[:, g * Ci:(g + 1) * Ci]
        parts.append(_conv3d_weight_contract(c, im, ks, st, pd, dl))
    out = parts[0] if G == 1 else torch.cat(parts, 2)
    return out.view(V, N, *out.shape[2:]) if transposed else out


def _hip_conv3d_jac_t(module, M: Tensor, x: Tensor) -> Tensor:
    """Input rule of Conv3d (any ``groups``, zero padding) on the two-dimensional HIP kernel: for every depth tap ``kd`` the
    output slices ``od`` are a batch of two-dimensional problems whose results are added into the input slices
    ``d = od sd - pd + kd dd`` (for a fixed tap that map is one strided slice)."""
    W = module.weight.detach()
    (KD, KH, KW), (sd, sh, sw), (pdd, ph, pw), (dd, dh, dw) = module.kernel_size, module.stride, module.padding, module.dilation
    V, N, Cout, OD, OH, OW = M.shape
    _, Cin, D, H, Wd = x.shape
    G = module.groups
    Co, Ci = Cout // G, Cin // G
    Mt = M.permute(0, 1, 3, 2, 4, 5).reshape(V, N * OD, Cout, OH, OW)
    g = M.new_zeros((V, N, Cin, D, H, Wd))
    for kd in range(KD):
        o_lo, o_hi, d_lo = _depth_range(OD, D, sd, pdd, kd * dd)
        if o_hi < o_lo:
            continue
        Wk = W[:, :, kd]
        if G == 1:
            res = kernels.conv2d_jac_t(Mt, Wk, (H, Wd), (sh, sw), (ph, pw), (dh, dw))
        else:
            res = torch.cat([kernels.conv2d_jac_t(Mt[:, :, i * Co:(i + 1) * Co].contiguous(), Wk[i * Co:(i + 1) * Co].contiguous(),
                                                  (H, Wd), (sh, sw), (ph, pw), (dh, dw)) for i in range(G)], 2)
        res = res.view(V, N, OD, Cin, H, Wd)[:, :, o_lo:o_hi + 1].permute(0, 1, 3, 2, 4, 5)
        g[:, :, :, d_lo:d_lo + (o_hi - o_lo) * sd + 1:sd] += res
    return g


def _hip_convtranspose3d_jac_t(module, M: Tensor, x: Tensor) -> Optional[Tensor]:
    """Input rule of ConvTranspose3d: a forward convolution of ``M`` with the weight,
    ``g[ci, q] = sum_{co, k} W[ci, co, k] M[co, q s - p + k d]`` -- per depth tap ``kd`` the two-dimensional rule of
    :func:`_convtranspose2d_input` on the slices ``M[.., qd sd - pd + kd dd]`` (a batch), summed over the taps."""
    W = module.weight.detach()
    (KD, KH, KW), (sd, sh, sw), (pdd, ph, pw), (dd, dh, dw) = module.kernel_size, module.stride, module.padding, module.dilation
    V, N, Cout, Dm, Hm, Wm = M.shape
    _, Cin, QD, QH, QW = x.shape
    g = M.new_zeros((V, N, Cin, QD, QH, QW))
    for kd in range(KD):
        q_lo, q_hi, d_lo = _depth_range(QD, Dm, sd, pdd, kd * dd)
        if q_hi < q_lo:
            continue
        nq = q_hi - q_lo + 1
        Mk = M[:, :, :, d_lo:d_lo + (nq - 1) * sd + 1:sd].permute(0, 1, 3, 2, 4, 5).reshape(V, N * nq, Cout, Hm, Wm)
        res = _convtranspose2d_input(Mk, W[:, :, kd], (KH, KW), (sh, sw), (ph, pw), (dh, dw), module.groups, (QH, QW))
        if res is None:
            return None
        g[:, :, :, q_lo:q_hi + 1] += res.view(V, N, nq, Cin, QH, QW).permute(0, 1, 3, 2, 4, 5)
    return g


def _hip_pool3d_jac_t(module, M: Tensor, x: Tensor) -> Optional[Tensor]:
    """MaxPool3d / AvgPool3d on the two-dimensional pooling kernels.  Both poolings are separable -- a window over
    (d, h, w) is a window over (h, w) in every depth slice followed by a window over d -- and so are their Jacobians:
    the depth stage is the two-dimensional kernel on the image ``[D, OH * OW]`` with a ``(kd, 1)`` window.  For the
    maximum the composition selects the same element as the three-dimensional scan (first maximum in d-major order),
    and -inf padding composes; the average (count_include_pad) divides by kh kw and then by kd."""
    V, N, C, OD, OH, OW = M.shape
    _, _, D, H, W = x.shape
    ks = _triple(module.kernel_size)
    st = _triple(module.stride if module.stride is not None else module.kernel_size)
    pd = _triple(module.padding)
    if isinstance(module, nn.MaxPool3d):
        if _triple(module.dilation) != (1, 1, 1) or module.ceil_mode or module.return_indices:
            return None
        x2 = x.reshape(N, C * D, H, W)
        y1 = F.max_pool2d(x2, ks[1:], st[1:], pd[1:])                                   # the in-plane maxima (forward values only)
        g1 = kernels.maxpool2d_jac_t(M.reshape(V, N, C, OD, OH * OW), y1.reshape(N, C, D, OH * OW), (ks[0], 1), (st[0], 1), (pd[0], 0))
        return kernels.maxpool2d_jac_t(g1.reshape(V, N, C * D, OH, OW), x2, ks[1:], st[1:], pd[1:]).view(V, N, C, D, H, W)
    if module.ceil_mode or not module.count_include_pad or module.divisor_override is not None:
        return None
    g1 = kernels.avgpool2d_jac_t(M.reshape(V, N, C, OD, OH * OW), (D, OH * OW), (ks[0], 1), (st[0], 1), (pd[0], 0))
    return kernels.avgpool2d_jac_t(g1.reshape(V, N, C * D, OH, OW), (H, W), ks[1:], st[1:], pd[1:]).view(V, N, C, D, H, W)


def _conv_weight_factor(module, M: Tensor, x: Tensor) -> Tensor:
    """Weight rule of the convolution family (Conv1d/3d, grouped Conv2d, ConvTranspose1d/2d/3d; the derivative classes
    of vivit/extensions/secondorder/vivit/convnd.py:9-30 and convtransposend.py:9-30): per sample ``n`` and slice ``v``
    the vector-Jacobian product of the functional convolution w.r.t. its weight, batched with ``torch.func.vmap``."""
    from torch.func import vjp, vmap

    fn = _CONV_FN[type(module)]
    kw = dict(stride=module.stride, padding=module.padding, dilation=module.dilation, groups=module.groups)
    if isinstance(module, (nn.ConvTranspose1d, nn.ConvTranspose2d, nn.ConvTranspose3d)):
        kw["output_padding"] = module.output_padding
    if getattr(module, "padding_mode", "zeros") != "zeros":
        raise NotImplementedError("only zero padding is supported by the stand-in backend")
    weight = module.weight.detach()

    def single(xn, mvn):  # xn: [*in], mvn: [*out]
        _, pull = vjp(lambda w: fn(xn.unsqueeze(0), w, None, **kw).squeeze(0), weight)
        return pull(mvn)[0]

    return vmap(vmap(single, in_dims=(0, 0)), in_dims=(None, 0))(x, M)


_ACTIVATIONS = {nn.ReLU: ("relu", None), nn.Sigmoid: ("sigmoid", None), nn.Tanh: ("tanh", None),
                nn.LeakyReLU: ("leaky_relu", "negative_slope"), nn.LogSigmoid: ("logsigmoid", None), nn.ELU: ("elu", "alpha"),
                nn.SELU: ("selu", None)}


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def _single(v):
    return v if isinstance(v, int) else tuple(v)[0]


def _hip_jac_t_mat_prod(module, M: Tensor, x: Tensor) -> Optional[Tensor]:
    """The layer rules that have a HIP kernel (csrc/jacobians.hip): activations, Flatten / Identity / ActiveIdentity / Dropout(eval), ScaleModule,
    Max/AvgPool1d/2d, Conv1d / Conv2d and ConvTranspose1d / 2d (any groups, zero padding), Pad / ZeroPad2d / Slicing,
    BatchNorm (eval), and -- on the same two-dimensional kernels -- Conv3d, ConvTranspose3d, MaxPool3d, AvgPool3d.  ``None``: no
    kernel for this module (custom modules, unsupported options such as ceil_mode) -- the generic autograd rule takes over."""
    kind = _ACTIVATIONS.get(type(module))
    if kind is not None:
        return kernels.act_jac_t(M, x, kind[0], getattr(module, kind[1]) if kind[1] else 0.0)
    if isinstance(module, (nn.Flatten, nn.Identity, nn.Dropout, ActiveIdentity)):
        return M.reshape(M.shape[0], *x.shape)
    if isinstance(module, ScaleModule):   # SqrtGGNScaleModule (__init__.py:113-116): the Jacobian is ``weight * I``
        if module.weight == 1.0:
            return M
        Mc = M if M.dim() > 3 else M.unsqueeze(-1)
        return kernels.channel_scale(Mc, M.new_full((Mc.shape[2],), module.weight)).view(M.shape)
    # index modules: the transposed Jacobian of a zero / constant padding is a crop, that of a slicing a scatter into zeros
    # (SqrtGGNPad / SqrtGGNZeroPad2d / SqrtGGNSlicing, __init__.py:110-117) -- no recomputed forward, no autograd
    if isinstance(module, (Pad, nn.ZeroPad2d)):
        pad = tuple(module.padding) if isinstance(module, nn.ZeroPad2d) else tuple(module.pad)
        mode = "constant" if isinstance(module, nn.ZeroPad2d) else module.mode
        if mode == "constant" and len(pad) % 2 == 0 and all(p_ >= 0 for p_ in pad) and len(pad) // 2 <= x.dim():
            g = M
            for i in range(len(pad) // 2):          # F.pad: the last dimension first
                d = g.dim() - 1 - i
                g = g.narrow(d, pad[2 * i], g.shape[d] - pad[2 * i] - pad[2 * i + 1])
            return g.contiguous()
    if isinstance(module, Slicing):
        g = M.new_zeros((M.shape[0],) + tuple(x.shape))
        info = module.slice_info if isinstance(module.slice_info, tuple) else (module.slice_info,)
        g[(slice(None),) + tuple(info)] = M
        return g
    # one-dimensional pooling on the two-dimensional kernels (a row image), as the Conv1d rules do
    if isinstance(module, nn.MaxPool1d) and x.dim() == 3:
        if _single(module.dilation) == 1 and not module.ceil_mode and not module.return_indices:
            ks = _single(module.kernel_size)
            st = _single(module.stride if module.stride is not None else ks)
            g = kernels.maxpool2d_jac_t(M.unsqueeze(3), x.unsqueeze(2), (1, ks), (1, st), (0, _single(module.padding)))
            return g.squeeze(3)
    if isinstance(module, nn.AvgPool1d) and x.dim() == 3:
        if not module.ceil_mode and module.count_include_pad:
            ks = _single(module.kernel_size)
            st = _single(module.stride if module.stride is not None else ks)
            g = kernels.avgpool2d_jac_t(M.unsqueeze(3), (1, x.shape[2]), (1, ks), (1, st), (0, _single(module.padding)))
            return g.squeeze(3)
    if isinstance(module, nn.MaxPool2d) and x.dim() == 4:
        if _pair(module.dilation) == (1, 1) and not module.ceil_mode and not module.return_indices:
            ks = _pair(module.kernel_size)
            return kernels.maxpool2d_jac_t(M, x, ks, _pair(module.stride if module.stride is not None else ks), _pair(module.padding))
    if isinstance(module, nn.AvgPool2d) and x.dim() == 4:
        if not module.ceil_mode and module.count_include_pad and module.divisor_override is None:
            ks = _pair(module.kernel_size)
            return kernels.avgpool2d_jac_t(M, x.shape[2:], ks, _pair(module.stride if module.stride is not None else ks),
                                           _pair(module.padding))
    if isinstance(module, (nn.Conv2d, nn.Conv1d)) and isinstance(module.padding, tuple) and module.padding_mode == "zeros":
        # (a shape neither kernel takes comes back as VIVIT_E_UNSUPPORTED: the caller falls through to the generic rule)
        # a Conv1d is the Conv2d with one row: same kernel; groups > 1: one launch per group on its channel slices
        one_d = isinstance(module, nn.Conv1d)
        W = module.weight.detach()
        if one_d:
            M, W, hw = M.unsqueeze(3), W.unsqueeze(2), (1, x.shape[2])
            st, pd, dl = (1, module.stride[0]), (0, module.padding[0]), (1, module.dilation[0])
        else:
            hw, st, pd, dl = x.shape[2:], module.stride, module.padding, module.dilation
        G = module.groups
        Co = module.out_channels // G
        if G == 1:
            g = kernels.conv2d_jac_t(M, W, hw, st, pd, dl)
        else:
            g = torch.cat([kernels.conv2d_jac_t(M[:, :, i * Co:(i + 1) * Co].contiguous(), W[i * Co:(i + 1) * Co].contiguous(), hw, st, pd, dl)
                           for i in range(G)], 2)
        return g.squeeze(3) if one_d else g
    if (isinstance(module, (nn.ConvTranspose2d, nn.ConvTranspose1d)) and isinstance(module.padding, tuple)
            and getattr(module, "padding_mode", "zeros") == "zeros"):
        g = _hip_convtranspose_jac_t(module, M, x)
        if g is not None:
            return g
    # three-dimensional layers on the two-dimensional kernels (one launch per depth tap / separable pooling)
    if isinstance(module, nn.Conv3d) and x.dim() == 5 and isinstance(module.padding, tuple) and module.padding_mode == "zeros":
        return _hip_conv3d_jac_t(module, M, x)
    if (isinstance(module, nn.ConvTranspose3d) and x.dim() == 5 and isinstance(module.padding, tuple)
            and getattr(module, "padding_mode", "zeros") == "zeros"):
        g = _hip_convtranspose3d_jac_t(module, M, x)
        if g is not None:
            return g
    if isinstance(module, (nn.MaxPool3d, nn.AvgPool3d)) and x.dim() == 5:
        g = _hip_pool3d_jac_t(module, M, x)
        if g is not None:
            return g
    if isinstance(module, _BATCHNORM) and x.dim() >= 2:
        if _own_params(module):   # the parameter rules want the two row sums of the same pass
            return _bn_eval_rules(module, M, x)[0]
        return kernels.channel_scale(M if M.dim() > 3 else M.unsqueeze(-1), _bn_scale(module)).view(M.shape)
    return None


def _convtranspose2d_input(M: Tensor, W: Tensor, ks, st, pd, dl, G: int, hq) -> Optional[Tensor]:
    """``g[ci, q] = sum_{co, k} W[ci, co, k] M[co, q s - p + k d]`` for ``M [V, N, Cout, H', W']``, ``W [Cin, Cout / G, kh,
    kw]`` -> ``[V, N, Cin, *hq]`` on the HIP kernel of the convolution's input rule.

    With the kernel index reversed (k' = K - 1 - k) this is the stride-1 input rule
    ``sum_{o, k'} W'[o, ci, k'] M[o, y + p' - k' d]`` of a convolution with weight ``W'[co, ci, k'] = W[ci, co, K - 1 - k']``
    and padding ``p' = (K - 1) d - p``, evaluated at ``y = q s`` -- the kernel computes every y and the stride picks every
    s-th (s^2 of the work is discarded for s > 1; transposed convolutions are rare on this path).  ``None`` when p' would be
    negative or the kernel's channel limit is exceeded."""
    pad2 = tuple((k - 1) * d - p for k, d, p in zip(ks, dl, pd))
    Cin = W.shape[0]
    Co, Ci = W.shape[1], Cin // G
    if min(pad2) < 0 or Co * ks[0] * ks[1] > 1024:
        return None
    # input size of the stride-1 rule whose output size is M's:  H' = full + 2 p' - d (K - 1)  <=>  full = H' + 2 p - d (K - 1)
    full = tuple(h + 2 * p - d * (k - 1) for h, p, d, k in zip(M.shape[3:], pd, dl, ks))
    if any(f < (q - 1) * s_ + 1 for f, q, s_ in zip(full, hq, st)):
        return None
    parts = []
    for g in range(G):
        Wg = W[g * Ci:(g + 1) * Ci].transpose(0, 1).flip(2, 3).contiguous()            # [Co, Ci, kh, kw], reversed taps
        Mg = M if G == 1 else M[:, :, g * Co:(g + 1) * Co].contiguous()
        parts.append(kernels.conv2d_jac_t(Mg, Wg, full, (1, 1), pad2, dl))
    out = parts[0] if G == 1 else torch.cat(parts, 2)
    return out[..., ::st[0], ::st[1]][..., :hq[0], :hq[1]].contiguous()


def _hip_convtranspose_jac_t(module, M: Tensor, x: Tensor) -> Optional[Tensor]:
    """Input rule of ConvTranspose1d/2d (convtransposend.py:9-30): the transposed Jacobian of y = conv_transpose(x, W) is a
    forward convolution of ``M`` with ``W`` (:func:`_convtranspose2d_input`); a 1-D layer is the 2-D one with a single row."""
    one_d = isinstance(module, nn.ConvTranspose1d)
    W = module.weight.detach()
    if one_d:
        M, W = M.unsqueeze(3), W.unsqueeze(2)
        ks, st, pd, dl = (1, module.kernel_size[0]), (1, module.stride[0]), (0, module.padding[0]), (1, module.dilation[0])
        hq = (1, x.shape[2])
    else:
        ks, st, pd, dl = module.kernel_size, module.stride, module.padding, module.dilation
        hq = tuple(x.shape[2:])
    out = _convtranspose2d_input(M, W, ks, st, pd, dl, module.groups, hq)
    if out is None:
        return None
    return out.squeeze(3) if one_d else out


def _jac_t_mat_prod(module, M: Tensor, x: Tensor) -> Tensor:
    """Apply the transposed input-Jacobian of ``module`` to ``M`` [V, N, *out] -> [V, N, *in]."""
    if isinstance(module, nn.Linear):   # (any number of extra dimensions between batch and features: linear.py:38-39)
        if M.is_cuda and M.dtype == torch.float32:
            O = M.shape[-1]
            return kernels.gemm_nn(M.reshape(-1, O), module.weight.detach()).view(*M.shape[:-1], -1)
        return M @ module.weight.detach()
    if isinstance(module, nn.Dropout) and module.training and module.p > 0:
        raise NotImplementedError("Dropout must be in eval mode")
    if isinstance(module, _BATCHNORM) and module.training:
        raise NotImplementedError("BatchNorm must be in eval mode")
    if M.is_cuda and M.dtype == torch.float32:
        try:
            g = _hip_jac_t_mat_prod(module, M, x)
        except _lib.VivitHipError as exc:  # a shape beyond a kernel's launch limits: the generic rule below
            if exc.status != _lib.VIVIT_E_UNSUPPORTED:
                raise
            g = None
        if g is not None:
            return g
    # generic rule: batched vector-Jacobian product through a recomputed forward (bypasses hooks)
    with torch.enable_grad():
        xi = x.detach().requires_grad_(True)
        y = module.forward(xi)
        (g,) = torch.autograd.grad(y, xi, grad_outputs=M.reshape(M.shape[0], *y.shape), is_grads_batched=True)
    return g


# ---------------------------------------------------------------------------------------------
class BatchGrad(_Extension):
    """Per-sample gradients of the mini-batch loss (BackPACK ``BatchGrad``)."""

    savefield = "grad_batch"
    uses_grad = True   # reads autograd's gradient: the extensions' stream waits for the backward pass at every hook (engine.py)

    def __init__(self, subsampling: Optional[List[int]] = None, factorised: bool = False):
        super().__init__(subsampling)
        self._factorised = factorised

    def apply(self, ctx, module, g_out):
        params = _own_params(module)
        if not params or isinstance(module, _LOSSES):
            return
        sub = self.get_subsampling()
        g = subsample(g_out.detach(), 0, sub).unsqueeze(0)  # [1, N, *out]
        x = subsample(module.input0.detach(), 0, sub)
        for name, p in params:
            if self._factorised and isinstance(module, nn.Linear) and name == "weight" and g.dim() == 3:
                setattr(p, self.savefield, LinearFactor(g[0], x))
            else:
                setattr(p, self.savefield, _param_factor(module, name, g, x)[0])


# ---------------------------------------------------------------------------------------------
def _loss_hessian_sqrt(module, strategy: str, mc_samples: int, samples: Optional[Tensor]) -> Tensor:
    """Symmetric factor S [V, N, C] of the loss Hessian w.r.t. the model output."""
    out = module.input0.detach()
    if out.dim() != 2:
        raise NotImplementedError("loss input must be [N, C]")
    N, C = out.shape
    red = module.reduction
    if red not in ("mean", "sum"):
        raise NotImplementedError(f"reduction={red}")
    if isinstance(module, nn.CrossEntropyLoss) and out.is_cuda and out.dtype == torch.float32 and C * 4 <= 60 * 1024:
        norm = N if red == "mean" else 1
        if strategy == "exact":
            return kernels.ce_sqrt_hessian(out, 1.0 / math.sqrt(norm))     # softmax + sqrt(p_v)(delta_vc - p_c) in one kernel
        if samples is None:
            idx = torch.multinomial(out.softmax(dim=1), mc_samples, replacement=True)  # [N, M]
            samples = F.one_hot(idx.t(), C).to(out.dtype)  # [M, N, C]
        samples = samples.to(out.device, out.dtype)
        return kernels.ce_sqrt_hessian(out, 1.0 / math.sqrt(samples.shape[0] * norm), onehot=samples)
    if isinstance(module, nn.CrossEntropyLoss):
        p = out.softmax(dim=1)
        norm = N if red == "mean" else 1
        if strategy == "exact":
            sq = p.sqrt()
            # H_n = diag(p_n) - p_n p_n^T = sum_v S[v,n,:] S[v,n,:]^T,  S[v,n,c] = sqrt(p_nv)(delta_vc - p_nc)
            S = torch.einsum("nv,vc->vnc", sq, torch.eye(C, dtype=out.dtype, device=out.device)) - torch.einsum(
                "nv,nc->vnc", sq, p
            )
            return S / math.sqrt(norm)
        if samples is None:
            idx = torch.multinomial(p, mc_samples, replacement=True)  # [N, M]
            samples = F.one_hot(idx.t(), C).to(out.dtype)  # [M, N, C]
        return (p.unsqueeze(0) - samples.to(out.device, out.dtype)) / math.sqrt(samples.shape[0] * norm)
    if isinstance(module, nn.MSELoss):
        norm = N * C if red == "mean" else 1
        if strategy == "exact":
            S = torch.eye(C, dtype=out.dtype, device=out.device).unsqueeze(1).expand(C, N, C)
            return S * math.sqrt(2.0 / norm)
        if samples is None:
            samples = torch.randn(mc_samples, N, C, dtype=out.dtype, device=out.device)
        return samples.to(out.device, out.dtype) * math.sqrt(2.0 / (samples.shape[0] * norm))
    raise NotImplementedError(type(module).__name__)


class _SqrtGGN(_Extension):
    strategy = "exact"
    uses_grad = False   # reads forward activations and the back-propagated factor only, never ``g_out``

    def __init__(self, subsampling=None, mc_samples: int = 1, samples: Optional[Tensor] = None, factorised: bool = False):
        super().__init__(subsampling)
        self._mc_samples = mc_samples
        self._samples = samples  # externally supplied MC one-hots [M, N, C] (parity needs them)
        self._factorised = factorised

    def get_num_mc_samples(self) -> int:
        return self._mc_samples

    def _store(self, module, name, param, M, x):
        if self._factorised and isinstance(module, nn.Linear) and name == "weight" and M.dim() == 3:
            setattr(param, self.savefield, LinearFactor(M, x))
        else:
            setattr(param, self.savefield, _param_factor(module, name, M, x))

    def apply(self, ctx, module, g_out):
        sub = self.get_subsampling()
        if isinstance(module, _LOSSES):
            S = _loss_hessian_sqrt(module, self.strategy, self._mc_samples, self._samples)
            ctx.put(self, module.input0, subsample(S, 1, sub))
            return
        M = ctx.pop(self, module.output)
        if M is None:
            return
        if isinstance(module, SumModule):  # identity Jacobian w.r.t. every summand (SqrtGGNSumModule)
            for inp in module.inputs:
                if inp.requires_grad:
                    ctx.put(self, inp, M)
            return
        x = subsample(module.input0.detach(), 0, sub)
        for name, p in _own_params(module):
            self._store(module, name, p, M, x)
        if module.input0.requires_grad:
            ctx.put(self, module.input0, _jac_t_mat_prod(module, M, x))


class SqrtGGNExact(_SqrtGGN):
    """Materialised ``V_t`` with the exact loss Hessian (BackPACK ``SqrtGGNExact``)."""

    savefield = "sqrt_ggn_exact"
    strategy = "exact"

    def __init__(self, subsampling=None, factorised: bool = False):
        super().__init__(subsampling, factorised=factorised)


class SqrtGGNMC(_SqrtGGN):
    """Materialised ``V_t`` with an MC-sampled loss Hessian (BackPACK ``SqrtGGNMC``)."""

    savefield = "sqrt_ggn_mc"
    strategy = "sampling"

    def __init__(self, mc_samples: int = 1, subsampling=None, samples: Optional[Tensor] = None, factorised: bool = False):
        super().__init__(subsampling, mc_samples, samples, factorised=factorised)


# ---------------------------------------------------------------------------------------------
def _materialised_closures(V_t: Tensor):
    """Closures over a materialised ``V_t`` (vivit/extensions/secondorder/vivit/base.py:96-130)."""

    def gram_mat(out=None, beta=0.0):
        return pairwise_dot(V_t, start_dim=2, flatten=False, out=out, beta=beta)

    def dp_add(acc):  # data parallel: this rank's rows of V_t into a vivit_amd.distributed.BatchShardedGram
        if V_t[0, 0].numel() < DP_ROWS_MAX_COLUMNS:
            acc.add_factor_rows(V_t)
        else:
            acc.add_factor(V_t)

    return {
        "V_mat_prod": lambda mat: Vmp(V_t, mat, 2),
        "V_t_mat_prod": lambda mat: mVp(V_t, mat, 2),
        "gram_mat": gram_mat,
        "dp_add": dp_add,
        # explicit factor [C, N, *param] and its leading shape: used when the parameter side of a group is the
        # smaller one (vivit_amd.linalg.utils.parameter_side_symeig)
        "factor": lambda: V_t,
        "shape_cn": tuple(V_t.shape[:2]),
    }


def _linear_weight_closures(s: Tensor, z: Tensor):
    """Factorised closures for a Linear weight: ``V_t[c,n,o,i] = s[c,n,o] z[n,i]`` never exists.

    vivit/extensions/secondorder/vivit/linear.py:44-75.  Gram: two small SYRKs plus the fused
    Hadamard kernel (K1'); products: one tall GEMM with ``z`` plus a length-C contraction.
    """
    C, N, O = s.shape
    s = s.contiguous()
    z = z.contiguous()

    def gram_mat(out=None, beta=0.0):
        Gz = kernels.gram_syrk(z)                      # [N, N]
        Gs = kernels.gram_syrk(s.reshape(C * N, O))    # [CN, CN]
        out2d = None if out is None else out.view(C * N, C * N)
        G = kernels.gram_hadamard(Gz, Gs, C, N, out=out2d, alpha=1.0, beta=beta)
        return G.view(C, N, C, N)

    def V_mat_prod(mat):  # [F, C, N] -> [F, O, I]     "cno,vcn,ni->voi"
        Fdim = mat.shape[0]
        T = kernels.class_contract(mat, s).view(Fdim * O, N)
        return kernels.gemm_nn(T, z).view(Fdim, O, z.shape[1])

    def V_t_mat_prod(mat):  # [F, O, I] -> [F, C, N]   "cno,voi,ni->vcn"
        Fdim = mat.shape[0]
        U = kernels.gemm_nt(mat.reshape(Fdim * O, -1), z).view(Fdim, O, N)
        return kernels.class_expand(s, U)

    def factor():  # the explicit V_t[c,n,o,i] = s[c,n,o] z[n,i]; only asked for when O*I is small
        return LinearFactor(s, z).materialise()

    return {"V_mat_prod": V_mat_prod, "V_t_mat_prod": V_t_mat_prod, "gram_mat": gram_mat, "factor": factor,
            "shape_cn": (C, N), "dp_add": lambda acc: acc.add_linear(s, z)}


class _ViViTGGN(_SqrtGGN):
    """Functional access to ``V``, ``V^T`` and the Gram matrix
    (vivit/extensions/secondorder/vivit/__init__.py:63-133)."""

    def _store(self, module, name, param, M, x):
        if isinstance(module, nn.Linear) and name == "weight" and M.dim() == 3:
            closures = _linear_weight_closures(M, x)
        else:
            closures = _materialised_closures(_param_factor(module, name, M, x))
        setattr(param, self.savefield, closures)


class ViViTGGNExact(_ViViTGGN):
    savefield = "vivit_ggn_exact"
    strategy = "exact"

    def __init__(self, subsampling=None):
        super().__init__(subsampling)


class ViViTGGNMC(_ViViTGGN):
    savefield = "vivit_ggn_mc"
    strategy = "sampling"

    def __init__(self, mc_samples: int = 1, subsampling=None, samples: Optional[Tensor] = None):
        super().__init__(subsampling, mc_samples, samples)
