"""Host-side launchers: torch tensors in, HIP kernels (libvivit_hip.so) out.

PyTorch is plumbing here (device memory, streams); all arithmetic happens in the hand-written
gfx950 kernels behind the C ABI of ``include/vivit_hip.h``.  Every function raises
``RuntimeError`` for tensors that are not fp32 HIP-device tensors: there is deliberately no CPU
path and no backend switch (host-logic tests monkeypatch these functions from ``tests/helpers.py``).
"""
import functools
from typing import Optional, Tuple

import torch

from vivit_amd import _lib

_WORKSPACES = {}


def _require_device(*tensors):
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(
                "vivit_amd kernels need HIP-device tensors (got a CPU tensor); there is no CPU fallback"
            )
        if t.dtype != torch.float32:
            raise RuntimeError(f"vivit_amd kernels are fp32 only (got {t.dtype})")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f"all operands must live on one device (got {dev} and {t.device})")


def _first_device(args, kwargs):
    for a in list(args) + list(kwargs.values()):
        if isinstance(a, torch.Tensor):
            return a.device if a.is_cuda else None
        if isinstance(a, (list, tuple)) and a and isinstance(a[0], torch.Tensor):
            return a[0].device if a[0].is_cuda else None
    return None


def _launcher(fn):
    """Run ``fn`` with the HIP device of its first tensor operand current.

    The C side launches on the *current* device (``<<<>>>`` on the given stream, the dynamic-LDS attribute bitmap is
    keyed on ``hipGetDevice``), so a process that drives several GPUs must switch before every call -- torch ops do
    that implicitly, a raw ctypes call does not."""

    @functools.wraps(fn)
    def wrapped(*args, **kwargs):
        dev = _first_device(args, kwargs)
        if dev is None or dev.index == torch.cuda.current_device():
            return fn(*args, **kwargs)
        with torch.cuda.device(dev):
            return fn(*args, **kwargs)

    return wrapped


def _launcher_method(fn):
    """``_launcher`` for methods whose device is that of ``self.evals``."""

    @functools.wraps(fn)
    def wrapped(self, *args, **kwargs):
        dev = self.evals.device
        if not dev.type == "cuda" or dev.index == torch.cuda.current_device():
            return fn(self, *args, **kwargs)
        with torch.cuda.device(dev):
            return fn(self, *args, **kwargs)

    return wrapped


def _check_out(out, m, n, *operands):
    """``out`` is written by the kernel with a single leading dimension: shape, strides and aliasing must be right,
    otherwise the launch would write out of bounds or into an operand it is still reading."""
    if out is None:
        return
    if tuple(out.shape) != (m, n):
        raise ValueError(f"out must have shape {(m, n)}, got {tuple(out.shape)}")
    if out.numel() > 0 and (out.stride(1) != 1 or (m > 1 and out.stride(0) < n)):
        raise ValueError(f"out must be row-major with unit column stride (strides {out.stride()})")
    for t in operands:
        if t is not None and t.numel() > 0 and out.numel() > 0 and t.untyped_storage().data_ptr() == out.untyped_storage().data_ptr():
            lo, hi = out.data_ptr(), out.data_ptr() + 4 * ((m - 1) * out.stride(0) + n)
            tlo = t.data_ptr()
            thi = tlo + 4 * (sum((sz - 1) * st for sz, st in zip(t.shape, t.stride())) + 1)
            if tlo < hi and lo < thi:
                raise ValueError("out must not alias an input operand")


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _workspace(nbytes: int, like: torch.Tensor):
    """Stream-ordered scratch buffer (grown on demand, cached per device and stream)."""
    if nbytes == 0:
        return None, 0
    key = (like.device, _stream(like))
    buf = _WORKSPACES.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=like.device)
        _WORKSPACES[key] = buf
    return buf.data_ptr(), buf.numel()


def _as2d(t: torch.Tensor) -> torch.Tensor:
    if t.dim() != 2:
        raise ValueError(f"expected a matrix, got {t.dim()} dimensions")
    if t.numel() > 0 and t.stride(1) != 1:
        t = t.contiguous()
    if t.numel() > 0 and t.stride(0) < t.shape[1]:
        t = t.contiguous()
    return t


def _ld(t):
    return max(t.stride(0), t.shape[1], 1) if t.shape[0] > 1 else max(t.shape[1], 1)


@_launcher
def gram_syrk(A: torch.Tensor, out: Optional[torch.Tensor] = None, alpha: float = 1.0, beta: float = 0.0):
    """``out = alpha * A @ A.T + beta * out`` for ``A: [n, p]`` (K1, MFMA SYRK)."""
    _require_device(A, out)
    A = _as2d(A)
    n, p = A.shape
    _check_out(out, n, n, A)
    if out is None:
        out = torch.empty((n, n), dtype=torch.float32, device=A.device)
        beta = 0.0
    lib = _lib.load()
    ws, wsb = _workspace(lib.vivit_gram_syrk_f32_workspace_bytes(n, p), A)
    st = lib.vivit_gram_syrk_f32(A.data_ptr(), n, p, _ld(A), out.data_ptr(), _ld(out), alpha, beta, ws, wsb, _stream(A))
    _lib.check(st, "vivit_gram_syrk_f32")
    return out


def _gemm(name, A, B, m, n, k, out, alpha, beta):
    lib = _lib.load()
    _check_out(out, m, n, A, B)
    if out is None:
        out = torch.empty((m, n), dtype=torch.float32, device=A.device)
        beta = 0.0
    ws, wsb = _workspace(lib.vivit_gemm_f32_workspace_bytes(m, n, k), A)
    st = getattr(lib, name)(
        A.data_ptr(), B.data_ptr(), out.data_ptr(), m, n, k, _ld(A), _ld(B), _ld(out), alpha, beta, ws, wsb, _stream(A)
    )
    _lib.check(st, name)
    return out


@_launcher
def gemm_nt(A, B, out=None, alpha: float = 1.0, beta: float = 0.0):
    """``out = alpha * A @ B.T + beta * out``; ``A: [m, k]``, ``B: [n, k]`` (K2/K9)."""
    _require_device(A, B, out)
    A, B = _as2d(A), _as2d(B)
    if A.shape[1] != B.shape[1]:
        raise ValueError("Trailing dimensions don't match.")
    return _gemm("vivit_gemm_nt_f32", A, B, A.shape[0], B.shape[0], A.shape[1], out, alpha, beta)


@_launcher
def gemm_nn(A, B, out=None, alpha: float = 1.0, beta: float = 0.0):
    """``out = alpha * A @ B + beta * out``; ``A: [m, k]``, ``B: [k, n]`` (K6/K7/K8)."""
    _require_device(A, B, out)
    A, B = _as2d(A), _as2d(B)
    if A.shape[1] != B.shape[0]:
        raise ValueError("Inner dimensions don't match.")
    return _gemm("vivit_gemm_nn_f32", A, B, A.shape[0], B.shape[1], A.shape[1], out, alpha, beta)


@_launcher
def gemm_tn(A, B, out=None, alpha: float = 1.0, beta: float = 0.0):
    """``out = alpha * A.T @ B + beta * out``; ``A: [k, m]``, ``B: [k, n]`` (K5)."""
    _require_device(A, B, out)
    A, B = _as2d(A), _as2d(B)
    if A.shape[0] != B.shape[0]:
        raise ValueError("Leading dimensions don't match.")
    return _gemm("vivit_gemm_tn_f32", A, B, A.shape[1], B.shape[1], A.shape[0], out, alpha, beta)


@_launcher
def gram_hadamard(Gz, Gs, C: int, N: int, out=None, alpha: float = 1.0, beta: float = 0.0):
    """``out[c,n,d,m] = alpha * Gz[n,m] * Gs[c,n,d,m] + beta * out`` (K1')."""
    _require_device(Gz, Gs, out)
    Gz, Gs = Gz.contiguous(), Gs.contiguous()
    if out is None:
        out = torch.empty((C * N, C * N), dtype=torch.float32, device=Gz.device)
        beta = 0.0
    if not out.is_contiguous() or tuple(out.shape) != (C * N, C * N):
        raise ValueError(f"out must be a contiguous [{C * N}, {C * N}] matrix")
    if tuple(Gz.shape) != (N, N) or Gs.numel() != (C * N) ** 2:
        raise ValueError("Gz must be [N, N] and Gs [C*N, C*N]")
    st = _lib.load().vivit_gram_hadamard_f32(Gz.data_ptr(), Gs.data_ptr(), out.data_ptr(), C, N, alpha, beta, _stream(Gz))
    _lib.check(st, "vivit_gram_hadamard_f32")
    return out


@_launcher
def gram_hadamard_block(Gz, Gs, Cr: int, Nr: int, Cc: int, Nc: int, out=None, alpha: float = 1.0, beta: float = 0.0):
    """``out[(c,n),(d,m)] = alpha * Gz[n,m] * Gs[(c,n),(d,m)] + beta * out`` for a rectangular block (rows
    ``Cr x Nr``, columns ``Cc x Nc``): the block row of a batch shard, and ``V^T g`` of a factorised Linear weight."""
    _require_device(Gz, Gs, out)
    Gz, Gs = Gz.contiguous(), Gs.contiguous()
    rows, cols = Cr * Nr, Cc * Nc
    if tuple(Gz.shape) != (Nr, Nc) or Gs.numel() != rows * cols:
        raise ValueError(f"Gz must be [{Nr}, {Nc}] and Gs [{rows}, {cols}]")
    if out is None:
        out = torch.empty((rows, cols), dtype=torch.float32, device=Gz.device)
        beta = 0.0
    _check_out(out, rows, cols, Gz, Gs)
    st = _lib.load().vivit_gram_hadamard_block_f32(Gz.data_ptr(), Gs.data_ptr(), out.data_ptr(), Cr, Nr, Cc, Nc, _ld(out),
                                                   alpha, beta, _stream(Gz))
    _lib.check(st, "vivit_gram_hadamard_block_f32")
    return out


@_launcher
def class_contract(mat, s):
    """``T[f,o,n] = sum_c mat[f,c,n] s[c,n,o]``; ``mat: [F,C,N]``, ``s: [C,N,O]`` (first half of linear.py:53)."""
    _require_device(mat, s)
    mat, s = mat.contiguous(), s.contiguous()
    F, C, N = mat.shape
    if tuple(s.shape[:2]) != (C, N):
        raise ValueError(f"s must be [{C}, {N}, O], got {tuple(s.shape)}")
    O = s.shape[2]
    T = torch.empty((F, O, N), dtype=torch.float32, device=mat.device)
    st = _lib.load().vivit_class_contract_f32(mat.data_ptr(), s.data_ptr(), T.data_ptr(), F, C, N, O, _stream(mat))
    _lib.check(st, "vivit_class_contract_f32")
    return T


@_launcher
def class_expand(s, U):
    """``R[f,c,n] = sum_o s[c,n,o] U[f,o,n]``; ``s: [C,N,O]``, ``U: [F,O,N]`` (second half of linear.py:64)."""
    _require_device(s, U)
    s, U = s.contiguous(), U.contiguous()
    C, N, O = s.shape
    F = U.shape[0]
    if tuple(U.shape[1:]) != (O, N):
        raise ValueError(f"U must be [F, {O}, {N}], got {tuple(U.shape)}")
    R = torch.empty((F, C, N), dtype=torch.float32, device=s.device)
    st = _lib.load().vivit_class_expand_f32(s.data_ptr(), U.data_ptr(), R.data_ptr(), F, C, N, O, _stream(s))
    _lib.check(st, "vivit_class_expand_f32")
    return R


@_launcher
def symeig(G: torch.Tensor, eigenvectors: bool = False, overwrite: bool = False, info_out: Optional[list] = None
           ) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """Eigenvalues (ascending) and, optionally, column eigenvectors of symmetric ``G`` (K3/K4).

    Raises ``RuntimeError`` if the solver reports unconverged eigenvalues (the reference's
    behaviour for a failing ``Tensor.symeig``, vivit/utils/eig.py:37-40).  That check reads the device-side
    ``info`` word, i.e. synchronises with the stream; a caller that wants to stay asynchronous passes a list as
    ``info_out``: the ``info`` tensor is appended to it instead and nothing is read back (check it with
    :func:`check_info` when convenient).
    """
    _require_device(G)
    if G.dim() != 2 or G.shape[0] != G.shape[1]:
        raise ValueError(f"Input must be a square matrix. Got shape {tuple(G.shape)}.")
    n = G.shape[0]
    lib = _lib.load()

    def solve():
        A = _as2d(G)
        if A.data_ptr() == G.data_ptr() and not overwrite:
            A = A.clone()  # the solver destroys its input (reflectors are stored in it)
        w = torch.empty(n, dtype=torch.float32, device=G.device)
        Z = torch.empty((n, n), dtype=torch.float32, device=G.device) if eigenvectors else None
        info = torch.zeros(1, dtype=torch.int32, device=G.device)
        ws, wsb = _workspace(lib.vivit_symeig_f32_workspace_bytes(n, 1 if eigenvectors else 0), G)
        st = lib.vivit_symeig_f32(
            A.data_ptr(), n, _ld(A), w.data_ptr(), Z.data_ptr() if eigenvectors else None, n, ws, wsb, info.data_ptr(), _stream(G)
        )
        _lib.check(st, "vivit_symeig_f32")
        if info_out is not None:
            info_out.append(info)
        else:
            check_info(info)  # device->host sync; the reference syncs here too (criterion callback)
        return w, Z

    return _retry_on_launch_chain(solve, intact=not overwrite, G=G)


@_launcher
def linear_weight_mjp(s, z):
    """``V_t[c,n,o,i] = s[c,n,o] z[n,i]`` materialised (``param_mjp`` of a Linear weight, einsum "vno,ni->vnoi")."""
    _require_device(s, z)
    s, z = s.contiguous(), z.contiguous()
    C, N, O = s.shape
    if z.shape[0] != N:
        raise ValueError(f"z must be [{N}, in], got {tuple(z.shape)}")
    I = z.shape[1]
    V = torch.empty((C, N, O, I), dtype=torch.float32, device=s.device)
    st = _lib.load().vivit_linear_weight_mjp_f32(s.data_ptr(), z.data_ptr(), V.data_ptr(), C, N, O, I, _stream(s))
    _lib.check(st, "vivit_linear_weight_mjp_f32")
    return V


@_launcher
def conv2d_weight_mjp(M, x, kernel_size, stride, padding, dilation):
    """``param_mjp`` of a Conv2d weight (groups = 1, zero padding): ``M [V, N, Cout, OH, OW]``, ``x [N, Cin, H, W]`` ->
    ``[V, N, Cout, Cin, KH, KW]`` (unfold + einsum "vnol,nkl->vnok" without the im2col buffer)."""
    _require_device(M, x)
    M, x = M.contiguous(), x.contiguous()
    Vd, N, Cout, OH, OW = M.shape
    Nx, Cin, H, W = x.shape
    if Nx != N:
        raise ValueError(f"x must have batch size {N}, got {Nx}")
    KH, KW = kernel_size
    out = torch.empty((Vd, N, Cout, Cin, KH, KW), dtype=torch.float32, device=M.device)
    st = _lib.load().vivit_conv2d_weight_mjp_f32(
        M.data_ptr(), x.data_ptr(), out.data_ptr(), Vd * N, N, Cin, H, W, Cout, KH, KW, OH, OW, stride[0], stride[1],
        padding[0], padding[1], dilation[0], dilation[1], _stream(M))
    _lib.check(st, "vivit_conv2d_weight_mjp_f32")
    return out


ACTIVATION_KINDS = {"relu": 0, "sigmoid": 1, "tanh": 2, "leaky_relu": 3, "logsigmoid": 4, "elu": 5, "selu": 6}


@_launcher
def act_jac_t(M, x, kind: str, param: float = 0.0):
    """``out[v, n, ...] = M[v, n, ...] * f'(x[n, ...])`` for an elementwise activation ``f`` (``kind`` in
    :data:`ACTIVATION_KINDS`): the transposed input-Jacobian of SqrtGGN{ReLU,Sigmoid,Tanh,...}."""
    _require_device(M, x)
    M, x = M.contiguous(), x.contiguous()
    if tuple(M.shape[1:]) != tuple(x.shape):
        raise ValueError(f"M must be [V, *x.shape], got {tuple(M.shape)} for x {tuple(x.shape)}")
    out = torch.empty_like(M)
    st = _lib.load().vivit_act_jac_t_f32(M.data_ptr(), x.data_ptr(), out.data_ptr(), M.shape[0], x.numel(), ACTIVATION_KINDS[kind],
                                        float(param), _stream(M))
    _lib.check(st, "vivit_act_jac_t_f32")
    return out


@_launcher
def channel_scale(M, scale):
    """``out[v, n, c, ...] = M[v, n, c, ...] * scale[c]`` (BatchNorm in eval mode)."""
    _require_device(M, scale)
    M, scale = M.contiguous(), scale.contiguous()
    C = scale.numel()
    if M.dim() < 3 or M.shape[2] != C:
        raise ValueError(f"M must be [V, N, {C}, ...], got {tuple(M.shape)}")
    L = M[0, 0, 0].numel()
    out = torch.empty_like(M)
    st = _lib.load().vivit_channel_scale_f32(M.data_ptr(), scale.data_ptr(), out.data_ptr(), M.shape[0] * M.shape[1], C, L, _stream(M))
    _lib.check(st, "vivit_channel_scale_f32")
    return out


@_launcher
def maxpool2d_jac_t(M, x, kernel_size, stride, padding):
    """Transposed input-Jacobian of ``MaxPool2d`` (no dilation, no ceil_mode): ``M [V, N, C, OH, OW]``, ``x [N, C, H, W]``."""
    _require_device(M, x)
    M, x = M.contiguous(), x.contiguous()
    Vd, N, C, OH, OW = M.shape
    if tuple(x.shape[:2]) != (N, C):
        raise ValueError(f"x must be [{N}, {C}, H, W], got {tuple(x.shape)}")
    H, W = x.shape[2:]
    out = torch.empty((Vd, N, C, H, W), dtype=torch.float32, device=M.device)
    idx = torch.empty(N * C * OH * OW, dtype=torch.int32, device=M.device)
    st = _lib.load().vivit_maxpool2d_jac_t_f32(M.data_ptr(), x.data_ptr(), out.data_ptr(), idx.data_ptr(), Vd, N * C, H, W, OH, OW,
                                              kernel_size[0], kernel_size[1], stride[0], stride[1], padding[0], padding[1], _stream(M))
    _lib.check(st, "vivit_maxpool2d_jac_t_f32")
    return out


@_launcher
def avgpool2d_jac_t(M, in_hw, kernel_size, stride, padding):
    """Transposed input-Jacobian of ``AvgPool2d`` (count_include_pad, no ceil_mode): ``M [V, N, C, OH, OW]`` -> ``[V, N, C, H, W]``."""
    _require_device(M)
    M = M.contiguous()
    Vd, N, C, OH, OW = M.shape
    H, W = in_hw
    out = torch.empty((Vd, N, C, H, W), dtype=torch.float32, device=M.device)
    st = _lib.load().vivit_avgpool2d_jac_t_f32(M.data_ptr(), out.data_ptr(), Vd * N * C, H, W, OH, OW, kernel_size[0], kernel_size[1],
                                              stride[0], stride[1], padding[0], padding[1], _stream(M))
    _lib.check(st, "vivit_avgpool2d_jac_t_f32")
    return out


@_launcher
def conv2d_jac_t(M, weight, in_hw, stride, padding, dilation):
    """Transposed input-Jacobian of ``Conv2d`` (groups = 1, zero padding): ``M [V, N, Cout, OH, OW]``, ``weight
    [Cout, Cin, KH, KW]`` -> ``[V, N, Cin, H, W]``."""
    _require_device(M, weight)
    M, weight = M.contiguous(), weight.contiguous()
    Vd, N, Cout, OH, OW = M.shape
    Co, Cin, KH, KW = weight.shape
    if Co != Cout:
        raise ValueError(f"weight must have {Cout} output channels, got {Co}")
    H, W = in_hw
    out = torch.empty((Vd, N, Cin, H, W), dtype=torch.float32, device=M.device)
    st = _lib.load().vivit_conv2d_jac_t_f32(M.data_ptr(), weight.data_ptr(), out.data_ptr(), Vd * N, Cin, H, W, Cout, KH, KW, OH, OW,
                                           stride[0], stride[1], padding[0], padding[1], dilation[0], dilation[1], _stream(M))
    _lib.check(st, "vivit_conv2d_jac_t_f32")
    return out


@_launcher
def bn_eval_rules(M, x, scale, mean=None, rstd=None):
    """All rules of a BatchNorm in eval mode in one pass over the factor (``vivit_bn_eval_rules_f32``): ``M [V, N, C, *spatial]``,
    ``x [N, C, *spatial]`` (the module's input), ``scale [C]`` -> ``(M * scale_c  [like M], sum_l M x  [V, N, C], sum_l M  [V, N, C])``;
    with ``mean`` / ``rstd`` ``[C]`` the second result is the finished weight rule ``(sum_l M x - mean_c sum_l M) rstd_c``."""
    _require_device(M, x, scale)
    M, x, scale = M.contiguous(), x.contiguous(), scale.contiguous()
    if (mean is None) != (rstd is None):
        raise ValueError("mean and rstd come together")
    if mean is not None:
        mean, rstd = mean.contiguous(), rstd.contiguous()
    if M.dim() < 3 or tuple(M.shape[1:]) != tuple(x.shape) or scale.numel() != M.shape[2]:
        raise ValueError(f"M must be [V, *x.shape] with {scale.numel()} channels, got {tuple(M.shape)} for x {tuple(x.shape)}")
    Vd, N, C = M.shape[:3]
    L = M[0, 0, 0].numel()
    out = torch.empty_like(M)
    mx = torch.empty((Vd, N, C), dtype=torch.float32, device=M.device)
    ms = torch.empty((Vd, N, C), dtype=torch.float32, device=M.device)
    st = _lib.load().vivit_bn_eval_rules_f32(M.data_ptr(), x.data_ptr(), scale.data_ptr(), out.data_ptr(), mx.data_ptr(), ms.data_ptr(),
                                            Vd * N * C, N * C, C, L, mean.data_ptr() if mean is not None else None,
                                            rstd.data_ptr() if rstd is not None else None, _stream(M))
    _lib.check(st, "vivit_bn_eval_rules_f32")
    return out, mx, ms


@_launcher
def row_dot(M, X=None, rows_x: int = 1):
    """``out[r] = sum_l M[r, l] * (X[r % rows_x, l] if X is given else 1)`` for ``M [rows, L]`` (fixed summation order)."""
    _require_device(M, X)
    M = _as2d(M.contiguous())
    if X is not None:
        X = _as2d(X.contiguous())
        if X.shape != (rows_x, M.shape[1]):
            raise ValueError(f"X must be [{rows_x}, {M.shape[1]}], got {tuple(X.shape)}")
    out = torch.empty(M.shape[0], dtype=torch.float32, device=M.device)
    st = _lib.load().vivit_row_dot_f32(M.data_ptr(), X.data_ptr() if X is not None else None, out.data_ptr(), M.shape[0], rows_x,
                                      M.shape[1], _stream(M))
    _lib.check(st, "vivit_row_dot_f32")
    return out


@_launcher
def ce_sqrt_hessian(logits, scale: float, onehot=None):
    """Cross-entropy loss-Hessian square root ``S [V, N, C]`` from ``logits [N, C]``: exact (``V = C``) or, with
    ``onehot [M, N, C]``, the sampled factor ``(p - onehot) * scale``."""
    _require_device(logits, onehot)
    logits = logits.contiguous()
    N, C = logits.shape
    Vd = C if onehot is None else onehot.shape[0]
    if onehot is not None:
        onehot = onehot.contiguous()
        if tuple(onehot.shape[1:]) != (N, C):
            raise ValueError(f"onehot must be [M, {N}, {C}], got {tuple(onehot.shape)}")
    S = torch.empty((Vd, N, C), dtype=torch.float32, device=logits.device)
    st = _lib.load().vivit_ce_sqrt_hessian_f32(logits.data_ptr(), onehot.data_ptr() if onehot is not None else None, S.data_ptr(), N, C,
                                              Vd, float(scale), _stream(logits))
    _lib.check(st, "vivit_ce_sqrt_hessian_f32")
    return S


@_launcher
def pack_lower(G: torch.Tensor) -> torch.Tensor:
    """Lower triangle (diagonal included) of the square matrix ``G`` as a packed ``n (n + 1) / 2`` vector (row-major)."""
    _require_device(G)
    G = _as2d(G)
    n = G.shape[0]
    packed = torch.empty(n * (n + 1) // 2, dtype=torch.float32, device=G.device)
    st = _lib.load().vivit_pack_lower_f32(G.data_ptr(), n, _ld(G), packed.data_ptr(), _stream(G))
    _lib.check(st, "vivit_pack_lower_f32")
    return packed


@_launcher
def unpack_lower_(packed: torch.Tensor, G: torch.Tensor) -> torch.Tensor:
    """Inverse of :func:`pack_lower` into the (row-major) ``G``, both triangles written (symmetric result)."""
    _require_device(packed, G)
    n = G.shape[0]
    if G.dim() != 2 or G.shape[1] != n or G.stride(1) != 1 or packed.numel() != n * (n + 1) // 2 or not packed.is_contiguous():
        raise ValueError("unpack_lower_ needs a square row-major G and a contiguous packed vector of n (n + 1) / 2 floats")
    st = _lib.load().vivit_unpack_lower_f32(packed.data_ptr(), n, G.data_ptr(), _ld(G), _stream(G))
    _lib.check(st, "vivit_unpack_lower_f32")
    return G


class SymeigPlan:
    """A symmetric matrix reduced to tridiagonal form with ALL eigenvalues known (``evals``, ascending), waiting for
    the caller to say which eigenvectors it wants: the two launches around the reference's ``criterion`` callback
    (vivit/linalg/eigh.py:248-253).  :meth:`select` returns ``evecs[:, keep]`` of the reference."""

    def __init__(self, evals, n, A=None, state=None, full=None):
        self.evals, self.n = evals, n
        self._A, self._state, self._full = A, state, full

    @_launcher_method
    def select(self, keep) -> torch.Tensor:
        """``[n, K]`` column eigenvectors of ``evals[keep]`` (``keep``: any order, list of ints or an integer tensor)."""
        n = self.n
        dev = self.evals.device
        keep = [int(k) for k in (keep.tolist() if isinstance(keep, torch.Tensor) else keep)]
        keep = [k + n if k < 0 else k for k in keep]
        if any(k < 0 or k >= n for k in keep):
            raise IndexError(f"eigenvector index out of range for n = {n}")
        K = len(keep)
        if self._full is not None:  # small problem: all vectors already there
            return self._full[:, keep]
        if K == 0:
            return torch.empty((n, 0), dtype=torch.float32, device=dev)
        uniq = sorted(set(keep))
        idx = torch.tensor(uniq, dtype=torch.int32, device=dev)
        Zt = torch.empty((len(uniq), n), dtype=torch.float32, device=dev)
        info = torch.zeros(1, dtype=torch.int32, device=dev)
        lib = _lib.load()
        ws, wsb = _workspace(lib.vivit_symeig_select_f32_workspace_bytes(n, len(uniq)), self.evals)
        st = lib.vivit_symeig_select_f32(
            self._A.data_ptr(), n, _ld(self._A), idx.data_ptr(), len(uniq), Zt.data_ptr(), n, self._state.data_ptr(),
            self._state.numel(), ws, wsb, info.data_ptr(), _stream(self.evals))
        _lib.check(st, "vivit_symeig_select_f32")
        check_info(info)
        if uniq != keep:  # caller's order / repeated indices
            pos = {k: i for i, k in enumerate(uniq)}
            Zt = Zt[torch.tensor([pos[k] for k in keep], dtype=torch.long, device=dev)]
        return Zt.T


@_launcher
def symeig_reduce(G: torch.Tensor, overwrite: bool = False) -> SymeigPlan:
    """Phase 1 of the selected-eigenvector solver: all eigenvalues of symmetric ``G`` (ascending, in ``plan.evals``)
    and a reduced state from which ``plan.select(keep)`` produces only the wanted eigenvectors
    (``vivit_symeig_reduce_f32`` / ``vivit_symeig_select_f32``).  Synchronises (the caller's criterion needs the
    eigenvalues anyway) and raises ``RuntimeError`` on non-convergence like :func:`symeig`."""
    _require_device(G)
    if G.dim() != 2 or G.shape[0] != G.shape[1]:
        raise ValueError(f"Input must be a square matrix. Got shape {tuple(G.shape)}.")
    n = G.shape[0]
    if n < SYMEIG_ROWS_MIN_N:  # single-workgroup solver: all vectors cost nothing extra
        w, Z = symeig(G, eigenvectors=True, overwrite=overwrite)
        return SymeigPlan(w, n, full=Z)
    lib = _lib.load()

    def solve():
        A = _as2d(G)
        if A.data_ptr() == G.data_ptr() and not overwrite:
            A = A.clone()
        w = torch.empty(n, dtype=torch.float32, device=G.device)
        info = torch.zeros(1, dtype=torch.int32, device=G.device)
        state = torch.empty(lib.vivit_symeig_reduce_f32_workspace_bytes(n) + 256, dtype=torch.uint8, device=G.device)
        st = lib.vivit_symeig_reduce_f32(A.data_ptr(), n, _ld(A), w.data_ptr(), state.data_ptr(), state.numel(),
                                         info.data_ptr(), _stream(G))
        _lib.check(st, "vivit_symeig_reduce_f32")
        check_info(info)
        return SymeigPlan(w, n, A=A, state=state)

    return _retry_on_launch_chain(solve, intact=not overwrite, G=G)


class PersistentKernelTimeout(RuntimeError):
    """A persistent kernel of the eigensolver (one-launch tridiagonalisation, panel QR, bulge chase) could not get its
    workgroups resident together -- twice -- or one of its exchanges stalled: ``info = VIVIT_INFO_PERSIST_TIMEOUT``
    (include/vivit_hip.h, "Persistent kernels").  Not a numerical failure: the input may be fine; something else held
    the GPU's compute units (another process on the card, a long collective).  The solve's outputs are garbage."""


def check_info(info: torch.Tensor):
    """Map the eigensolver's device-side status word to the reference's RuntimeError (synchronises);
    :class:`PersistentKernelTimeout` (a RuntimeError too) for the persistent kernels' own status."""
    nfail = int(info.item())
    if nfail == _lib.VIVIT_INFO_PERSIST_TIMEOUT:
        raise PersistentKernelTimeout(
            "symeig: a persistent kernel could not become resident (two attempts of 2 s) or stalled; "
            "VIVIT_SYTRD_PERSIST=0 VIVIT_QR_PERSIST=0 VIVIT_SB2ST_PERSIST=0 select the launch chains")
    if nfail != 0:
        raise RuntimeError(f"symeig: {nfail} eigenvalues did not converge")


class persistent_kernels:
    """``with persistent_kernels(False): ...`` -- run the enclosed solves on the launch chains
    (``vivit_persistent_kernels``; the previous setting is restored on exit)."""

    def __init__(self, on: bool):
        self._on = 1 if on else 0

    def __enter__(self):
        self._prev = _lib.load().vivit_persistent_kernels(self._on)
        return self

    def __exit__(self, *exc):
        _lib.load().vivit_persistent_kernels(self._prev)
        return False


_BACKUP_ALWAYS_BYTES = 256 << 20   # inputs up to this size (n <= 8192) are always backed up before an in-place solve


def _wants_backup(G: torch.Tensor) -> bool:
    """An in-place solve (``overwrite=True``) keeps a copy of its input for the launch-chain retry when that is cheap
    (<= 256 MB: 0.1 ms) or when the card is likely to be shared -- a multi-rank job, whose RCCL kernels compete with the
    persistent kernels for compute units -- or on request (``VIVIT_PERSIST_BACKUP=1``; ``=0`` never)."""
    import os

    env = os.environ.get("VIVIT_PERSIST_BACKUP")
    if env is not None:
        return env != "0"
    if G.numel() * 4 <= _BACKUP_ALWAYS_BYTES:
        return True
    import torch.distributed as dist

    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def _retry_on_launch_chain(solve, intact: bool, G: Optional[torch.Tensor] = None):
    """Run ``solve()``; if it ends with :class:`PersistentKernelTimeout` and the input is still there -- the solve worked
    on a copy (``intact``), or ``G`` is solved in place and was backed up (:func:`_wants_backup`) -- repeat it ONCE with the
    persistent kernels switched off: the same stages as launch chains, which need no co-residency.  Otherwise the error
    propagates (the input is gone)."""
    backup = G.clone() if (not intact and G is not None and _wants_backup(G)) else None
    try:
        return solve()
    except PersistentKernelTimeout:
        if not intact and backup is None:
            raise
        import warnings

        warnings.warn("symeig: persistent kernel timed out; repeating the solve on the launch chains", RuntimeWarning)
        if backup is not None:
            G.copy_(backup)
        with persistent_kernels(False):
            return solve()


SYMEIG_ROWS_MIN_N = 193  # below: single-workgroup solver, no row-range entry point


@_launcher
def symeig_rows(G: torch.Tensor, row_begin: int, row_end: int, overwrite: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """All eigenvalues (ascending) of symmetric ``G`` and the eigenvectors ``row_begin .. row_end-1`` as ROWS
    ``[row_end - row_begin, n]`` (``vivit_symeig_rows_f32``): the unit of work of one rank in the multi-GPU
    eigensolver (``vivit_amd.distributed.symeig``)."""
    _require_device(G)
    if G.dim() != 2 or G.shape[0] != G.shape[1]:
        raise ValueError(f"Input must be a square matrix. Got shape {tuple(G.shape)}.")
    n = G.shape[0]
    if not (0 <= row_begin <= row_end <= n):
        raise ValueError(f"invalid eigenvector range [{row_begin}, {row_end}) for n = {n}")
    if n < SYMEIG_ROWS_MIN_N:  # same HIP solver, slice afterwards
        w, Z = symeig(G, eigenvectors=True, overwrite=overwrite)
        return w, Z.T[row_begin:row_end].contiguous()
    lib = _lib.load()

    def solve():
        A = _as2d(G)
        if A.data_ptr() == G.data_ptr() and not overwrite:
            A = A.clone()
        w = torch.empty(n, dtype=torch.float32, device=G.device)
        Zt = torch.empty((max(row_end - row_begin, 1), n), dtype=torch.float32, device=G.device)
        info = torch.zeros(1, dtype=torch.int32, device=G.device)
        ws, wsb = _workspace(lib.vivit_symeig_f32_workspace_bytes(n, 1), G)
        st = lib.vivit_symeig_rows_f32(
            A.data_ptr(), n, _ld(A), w.data_ptr(), Zt.data_ptr(), n, row_begin, row_end, ws, wsb, info.data_ptr(), _stream(G)
        )
        _lib.check(st, "vivit_symeig_rows_f32")
        check_info(info)
        return w, Zt[: row_end - row_begin]

    return _retry_on_launch_chain(solve, intact=not overwrite, G=G)


@_launcher
def stedc(d: torch.Tensor, e: torch.Tensor, eigenvectors: bool = False):
    """Eigen-decomposition of the symmetric tridiagonal (d, e) -- stage 2 of ``symeig`` (testing)."""
    _require_device(d, e)
    n = d.numel()
    d, e = d.contiguous().clone(), e.contiguous().clone()
    w = torch.empty(n, dtype=torch.float32, device=d.device)
    Z = torch.empty((n, n), dtype=torch.float32, device=d.device) if eigenvectors else None
    info = torch.zeros(1, dtype=torch.int32, device=d.device)
    lib = _lib.load()
    ws, wsb = _workspace(lib.vivit_stedc_f32_workspace_bytes(n, 1 if eigenvectors else 0), d)
    st = lib.vivit_stedc_f32(
        d.data_ptr(), e.data_ptr(), n, w.data_ptr(), Z.data_ptr() if eigenvectors else None, n, ws, wsb, info.data_ptr(), _stream(d)
    )
    _lib.check(st, "vivit_stedc_f32")
    if int(info.item()) != 0:
        raise RuntimeError(f"stedc: {int(info.item())} eigenvalues did not converge")
    return w, Z


@_launcher
def sytrd(G: torch.Tensor):
    """Householder tridiagonalisation -- stage 1 of ``symeig`` (testing).
    Returns ``(d, e, tau, A)`` with the reflectors in the upper triangle of ``A``."""
    _require_device(G)
    n = G.shape[0]
    A = G.contiguous().clone()
    d = torch.empty(n, dtype=torch.float32, device=G.device)
    e = torch.empty(n - 1, dtype=torch.float32, device=G.device)
    tau = torch.empty(n, dtype=torch.float32, device=G.device)
    lib = _lib.load()
    ws, wsb = _workspace(lib.vivit_sytrd_f32_workspace_bytes(n), G)
    st = lib.vivit_sytrd_f32(A.data_ptr(), n, n, d.data_ptr(), e.data_ptr(), tau.data_ptr(), ws, wsb, _stream(G))
    _lib.check(st, "vivit_sytrd_f32")
    return d, e, tau, A


@_launcher
def sy2sb(G: torch.Tensor):
    """Full symmetric -> band (testing). Returns ``(AB, tau1, A)``: band in row-band layout, reflector
    scalars, and the overwritten matrix (band + reflector rows)."""
    _require_device(G)
    lib = _lib.load()
    n = G.shape[0]
    nb = lib.vivit_sb2st_half_bandwidth()
    A = G.contiguous().clone()
    AB = torch.empty((n, 2 * nb + 1), dtype=torch.float32, device=G.device)
    tau1 = torch.empty(n, dtype=torch.float32, device=G.device)
    ws, wsb = _workspace(lib.vivit_sy2sb_f32_workspace_bytes(n), G)
    st = lib.vivit_sy2sb_f32(A.data_ptr(), n, n, AB.data_ptr(), tau1.data_ptr(), ws, wsb, _stream(G))
    _lib.check(st, "vivit_sy2sb_f32")
    return AB, tau1, A


BAND_NB = 64  # half bandwidth of the two-stage reduction (vivit_sb2st_half_bandwidth())


@_launcher
def symeig_prepare_(A: torch.Tensor) -> torch.Tensor:
    """In place: LAPACK-style scaling of symmetric ``A`` (lower triangle read) and mirror into the upper triangle --
    the state the band reduction starts from.  Returns the device ``scal[16]`` block for
    :func:`symeig_banded_rows` (``vivit_symeig_prepare_f32``)."""
    _require_device(A)
    n = A.shape[0]
    scal = torch.zeros(16, dtype=torch.float32, device=A.device)
    lib = _lib.load()
    ws, wsb = _workspace(8 * n + 512, A)
    st = lib.vivit_symeig_prepare_f32(A.data_ptr(), n, _ld(A), scal.data_ptr(), ws, wsb, _stream(A))
    _lib.check(st, "vivit_symeig_prepare_f32")
    return scal


@_launcher
def panel_qr_(pan: torch.Tensor):
    """Householder QR of one sub-band panel ``pan [mp, 64]`` (contiguous, factored in place: R's strict upper triangle
    stays in its first 64 rows).  Returns ``(Vt [64, mp], tau [64], betas [64], T [64, 64])`` with ``Q = I - V T V^T``
    (``vivit_sy2sb_panel_qr_f32``): the step every rank of the sharded band reduction repeats."""
    _require_device(pan)
    mp = pan.shape[0]
    if pan.dim() != 2 or pan.shape[1] != BAND_NB or not pan.is_contiguous():
        raise ValueError(f"panel must be a contiguous [mp, {BAND_NB}] matrix, got {tuple(pan.shape)}")
    dev = pan.device
    Vt = torch.empty((BAND_NB, mp), dtype=torch.float32, device=dev)
    tau = torch.empty(BAND_NB, dtype=torch.float32, device=dev)
    betas = torch.empty(BAND_NB, dtype=torch.float32, device=dev)
    T = torch.empty((BAND_NB, BAND_NB), dtype=torch.float32, device=dev)
    lib = _lib.load()
    ws, wsb = _workspace(lib.vivit_sy2sb_panel_qr_f32_workspace_bytes(mp), pan)
    st = lib.vivit_sy2sb_panel_qr_f32(pan.data_ptr(), mp, Vt.data_ptr(), mp, tau.data_ptr(), betas.data_ptr(), T.data_ptr(), ws,
                                      wsb, _stream(pan))
    _lib.check(st, "vivit_sy2sb_panel_qr_f32")
    return Vt, tau, betas, T


@_launcher
def symeig_banded_rows(A: torch.Tensor, tau1: torch.Tensor, scal: torch.Tensor, row_begin: int, row_end: int):
    """The two-stage solver entered after the band reduction (``A``: band + first-stage reflectors as ``sy2sb`` leaves
    them, destroyed): all eigenvalues and the eigenvectors ``row_begin .. row_end-1`` as rows
    (``vivit_symeig_banded_rows_f32``)."""
    _require_device(A)
    n = A.shape[0]
    w = torch.empty(n, dtype=torch.float32, device=A.device)
    Zt = torch.empty((max(row_end - row_begin, 1), n), dtype=torch.float32, device=A.device)
    info = torch.zeros(1, dtype=torch.int32, device=A.device)
    lib = _lib.load()
    ws, wsb = _workspace(lib.vivit_symeig_f32_workspace_bytes(n, 1), A)
    st = lib.vivit_symeig_banded_rows_f32(A.data_ptr(), n, _ld(A), tau1.data_ptr(), scal.data_ptr(), w.data_ptr(), Zt.data_ptr(),
                                          n, row_begin, row_end, ws, wsb, info.data_ptr(), _stream(A))
    _lib.check(st, "vivit_symeig_banded_rows_f32")
    check_info(info)
    return w, Zt[: row_end - row_begin]


@_launcher
def sb2st(AB: torch.Tensor):
    """Band -> tridiagonal by bulge chasing (testing). ``AB``: [n, 2*NB+1] row-band layout.
    Returns ``(d, e, R2, tau2)``."""
    _require_device(AB)
    lib = _lib.load()
    n = AB.shape[0]
    nb = lib.vivit_sb2st_half_bandwidth()
    assert AB.shape[1] == 2 * nb + 1 and AB.is_contiguous()
    AB = AB.clone()
    d = torch.empty(n, dtype=torch.float32, device=AB.device)
    e = torch.empty(n, dtype=torch.float32, device=AB.device)
    R2 = torch.zeros((n, n), dtype=torch.float32, device=AB.device)
    nbytes = lib.vivit_sb2st_f32_workspace_bytes(n)
    ws = torch.zeros(nbytes + 256, dtype=torch.uint8, device=AB.device)
    st = lib.vivit_sb2st_f32(AB.data_ptr(), n, d.data_ptr(), e.data_ptr(), R2.data_ptr(), ws.data_ptr(), ws.numel(), _stream(AB))
    _lib.check(st, "vivit_sb2st_f32")
    off = (-ws.data_ptr()) % 256
    nk = -(-n // nb) + 1
    tau2 = ws[off : off + 4 * n * nk].view(torch.float32).view(n, nk).clone()
    return d, e[: n - 1], R2, tau2


@_launcher
def q2_apply_(Zt: torch.Tensor, R2: torch.Tensor, tau2: torch.Tensor, mode: int = -1) -> torch.Tensor:
    """``Zt <- Zt Q2^T`` in place (testing): the back-transformation through the bulge-chasing reflectors ``(R2, tau2)``
    of :func:`sb2st`.  ``mode``: 0 block steps (fp32 MFMA), 1 sliding window on the bf16 pipe, -1 the solver's choice."""
    _require_device(Zt, R2, tau2)
    n = R2.shape[0]
    assert Zt.dim() == 2 and Zt.shape[1] == n and Zt.stride(1) == 1 and R2.is_contiguous() and tau2.is_contiguous()
    lib = _lib.load()
    nbytes = lib.vivit_q2_apply_f32_workspace_bytes(n)
    ws, wsb = _workspace(nbytes, Zt)
    st = lib.vivit_q2_apply_f32(Zt.data_ptr(), Zt.stride(0), Zt.shape[0], n, R2.data_ptr(), n, tau2.data_ptr(), ws, wsb, int(mode),
                                _stream(Zt))
    _lib.check(st, "vivit_q2_apply_f32")
    return Zt


@_launcher
def dir_curvature(GE, evals, C: int, N: int, scale: float):
    """``lambdas[n,k] = scale * sum_c GE[(c,n),k]^2 / evals[k]`` (K6 epilogue)."""
    _require_device(GE, evals)
    GE, evals = GE.contiguous(), evals.contiguous()
    K = evals.numel()
    out = torch.empty((N, K), dtype=torch.float32, device=GE.device)
    st = _lib.load().vivit_dir_curvature_f32(GE.data_ptr(), evals.data_ptr(), out.data_ptr(), C, N, K, scale, _stream(GE))
    _lib.check(st, "vivit_dir_curvature_f32")
    return out


@_launcher
def scale_cols_rsqrt_(X, evals, pre: float = 1.0):
    """In place ``X[:, k] *= pre / sqrt(evals[k])`` (K5 epilogue)."""
    _require_device(X, evals)
    if X.stride(1) != 1:
        raise ValueError("X must have unit column stride")
    rows, K = X.shape
    st = _lib.load().vivit_scale_cols_rsqrt_f32(X.data_ptr(), evals.contiguous().data_ptr(), rows, K, _ld(X), pre, _stream(X))
    _lib.check(st, "vivit_scale_cols_rsqrt_f32")
    return X


@_launcher
def normalize_rows_(tensors):
    """Normalise ``K`` stacked vectors given in parameter-list format, in place (K10).

    ``tensors``: list of ``[K, *param.shape]`` contiguous tensors; afterwards, for each ``k``,
    ``sum_t ||tensors[t][k]||^2 == 1``.  Replaces vivit/linalg/utils.py:67-76.
    """
    if len(tensors) == 0:
        return tensors
    _require_device(*tensors)
    K = tensors[0].shape[0]
    lib = _lib.load()
    acc = torch.zeros(K, dtype=torch.float32, device=tensors[0].device)
    flat = []
    for t in tensors:
        if not t.is_contiguous():
            raise ValueError("normalize_rows_ needs contiguous tensors")
        length = t.numel() // max(K, 1)
        flat.append((t, length))
        ws, wsb = _workspace(lib.vivit_row_sqnorm_workspace_bytes(K, length), t)
        st = lib.vivit_row_sqnorm_acc_f32(t.data_ptr(), acc.data_ptr(), K, length, ws, wsb, _stream(t))
        _lib.check(st, "vivit_row_sqnorm_acc_f32")
    for t, length in flat:
        st = lib.vivit_scale_rows_rsqrt_f32(t.data_ptr(), acc.data_ptr(), K, length, _stream(t))
        _lib.check(st, "vivit_scale_rows_rsqrt_f32")
    return tensors
