"""The native layer rules of the factor back-propagation (csrc/jacobians.hip, csrc/factors.hip) one by one against plain
torch autograd of the layer itself -- the definition BackPACK's derivative classes implement for the reference
(vivit/extensions/secondorder/vivit/__init__.py:84-118, base.py:19,41,84-92) -- at shapes of the BASELINE networks and at
awkward ones (strides, paddings, dilation, ragged channel counts), plus the packed-triangle kernels of the multi-GPU
all-reduce.  fp32; tolerances are those of a length-L fp32 sum."""
import math

import pytest
import torch
import torch.nn.functional as F
from torch import nn

from oracle import vivit_oracle as oracle

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def jac_t_by_autograd(module, x, M):
    """[V, N, *out] -> [V, N, *in]: vector-Jacobian products of ``module`` at ``x`` for every slice of ``M``."""
    x = x.detach().requires_grad_(True)
    y = module(x)
    return torch.stack([torch.autograd.grad(y, x, grad_outputs=M[v], retain_graph=True)[0] for v in range(M.shape[0])])


def close(a, b, rtol=1e-5, atol=1e-6):
    torch.testing.assert_close(a, b, rtol=rtol, atol=atol)


@pytest.mark.parametrize("kind,module", [("relu", nn.ReLU()), ("sigmoid", nn.Sigmoid()), ("tanh", nn.Tanh()),
                                         ("leaky_relu", nn.LeakyReLU(0.1)), ("logsigmoid", nn.LogSigmoid()),
                                         ("elu", nn.ELU(0.7)), ("selu", nn.SELU())])
def test_activation_rules(kind, module):
    from vivit_amd import kernels

    g = torch.Generator(device=DEV).manual_seed(0)
    x = torch.randn(37, 5, 11, generator=g, device=DEV) * 2
    M = torch.randn(3, *x.shape, generator=g, device=DEV)
    param = {"leaky_relu": 0.1, "elu": 0.7}.get(kind, 0.0)
    close(kernels.act_jac_t(M, x, kind, param), jac_t_by_autograd(module, x, M), rtol=2e-5, atol=1e-6)


@pytest.mark.parametrize("shape,k,s,p", [((64, 6, 28, 28), 2, 2, 0), ((9, 5, 13, 11), 3, 2, 1), ((4, 3, 8, 8), 3, 1, 1),
                                         ((3, 2, 7, 9), (2, 3), (1, 2), (1, 0))])
def test_pooling_rules(shape, k, s, p):
    from vivit_amd import kernels
    from vivit_amd.backend.extensions import _pair

    g = torch.Generator(device=DEV).manual_seed(1)
    x = torch.randn(*shape, generator=g, device=DEV)
    x[0, 0, :3, :3] = 1.5   # ties inside a window: the first maximum in scan order must take the gradient (as torch)
    for mod in (nn.MaxPool2d(k, s, p), nn.AvgPool2d(k, s, p)):
        y = mod(x)
        M = torch.randn(2, *y.shape, generator=g, device=DEV)
        ref = jac_t_by_autograd(mod, x, M)
        if isinstance(mod, nn.MaxPool2d):
            got = kernels.maxpool2d_jac_t(M, x, _pair(k), _pair(s), _pair(p))
        else:
            got = kernels.avgpool2d_jac_t(M, x.shape[2:], _pair(k), _pair(s), _pair(p))
        close(got, ref)


@pytest.mark.parametrize("cin,cout,hw,k,s,p,d", [(6, 16, (14, 14), 5, 1, 0, 1),      # LeNet conv2
                                                 (16, 32, (32, 32), 3, 2, 1, 1),     # ResNet-32 down-sampling block
                                                 (64, 64, (8, 8), 3, 1, 1, 1),       # ResNet-32 last stage
                                                 (5, 7, (9, 11), (2, 3), (2, 1), (1, 2), (2, 1)),   # ragged everything
                                                 (19, 3, (6, 6), 3, 1, 0, 1),        # more than one group of 16 input channels
                                                 # filter slices beyond the scalar input rule's LDS budget (Cout KH KW > 1024): the
                                                 # matrix-pipe input rule -- one chunk with split contraction, ragged channels and
                                                 # positions, stride 2 (inserted zeros), chunks of output channels
                                                 (40, 128, (8, 8), 3, 1, 1, 1),
                                                 (37, 150, (7, 9), 3, 2, 1, 1),
                                                 (16, 520, (20, 20), 3, 1, 1, 1),
                                                 (8, 72, (12, 12), 5, 1, 2, 1)])
def test_conv2d_rules(cin, cout, hw, k, s, p, d):
    from vivit_amd import kernels

    g = torch.Generator(device=DEV).manual_seed(2)
    N, Vd = 5, 3
    conv = nn.Conv2d(cin, cout, k, stride=s, padding=p, dilation=d, bias=True).to(DEV)
    x = torch.randn(N, cin, *hw, generator=g, device=DEV)
    y = conv(x)
    M = torch.randn(Vd, *y.shape, generator=g, device=DEV)
    # input rule = transposed convolution
    got = kernels.conv2d_jac_t(M, conv.weight.detach(), x.shape[2:], conv.stride, conv.padding, conv.dilation)
    close(got, jac_t_by_autograd(conv, x, M), rtol=1e-4, atol=1e-5)
    # weight rule: per-sample, per-slice gradient of <M[v, n], conv(x[n])> w.r.t. the weight
    gotw = kernels.conv2d_weight_mjp(M, x, conv.kernel_size, conv.stride, conv.padding, conv.dilation)
    ref = torch.empty_like(gotw)
    for v in range(Vd):
        for n in range(N):
            xn = x[n:n + 1]
            ref[v, n] = torch.autograd.grad(F.conv2d(xn, conv.weight, None, conv.stride, conv.padding, conv.dilation),
                                            conv.weight, grad_outputs=M[v, n:n + 1])[0]
    L = y.shape[2] * y.shape[3]
    close(gotw, ref, rtol=1e-4, atol=1e-5 * math.sqrt(L))
    # bias rule: sum over the output positions
    close(kernels.row_dot(M.reshape(-1, L)).view(Vd, N, cout), M.flatten(3).sum(3), rtol=1e-5, atol=1e-5 * math.sqrt(L))


@pytest.mark.parametrize("cin,cout,L,k,s,p,d", [(4, 6, 33, 5, 1, 2, 1), (3, 8, 40, 3, 2, 1, 2), (17, 5, 12, 2, 1, 0, 1)])
def test_conv1d_rules(cin, cout, L, k, s, p, d):
    """Conv1d runs on the Conv2d kernels (one row): input rule and weight rule through the backend dispatch."""
    from vivit_amd.backend.extensions import _jac_t_mat_prod, _param_factor

    g = torch.Generator(device=DEV).manual_seed(6)
    N, Vd = 4, 3
    conv = nn.Conv1d(cin, cout, k, stride=s, padding=p, dilation=d).to(DEV)
    x = torch.randn(N, cin, L, generator=g, device=DEV)
    y = conv(x)
    M = torch.randn(Vd, *y.shape, generator=g, device=DEV)
    close(_jac_t_mat_prod(conv, M, x), jac_t_by_autograd(conv, x, M), rtol=1e-4, atol=1e-5)
    gotw = _param_factor(conv, "weight", M, x)
    ref = torch.empty_like(gotw)
    for v in range(Vd):
        for n in range(N):
            ref[v, n] = torch.autograd.grad(F.conv1d(x[n:n + 1], conv.weight, None, conv.stride, conv.padding, conv.dilation),
                                            conv.weight, grad_outputs=M[v, n:n + 1])[0]
    close(gotw, ref, rtol=1e-4, atol=1e-5 * math.sqrt(y.shape[2]))
    close(_param_factor(conv, "bias", M, x), M.sum(3), rtol=1e-5, atol=1e-5 * math.sqrt(y.shape[2]))


def test_linear_extra_dims_input_rule():
    from vivit_amd.backend.extensions import _jac_t_mat_prod

    g = torch.Generator(device=DEV).manual_seed(7)
    lin = nn.Linear(13, 9).to(DEV)
    x = torch.randn(5, 3, 4, 13, generator=g, device=DEV)
    M = torch.randn(2, 5, 3, 4, 9, generator=g, device=DEV)
    close(_jac_t_mat_prod(lin, M, x), jac_t_by_autograd(lin, x, M), rtol=1e-4, atol=1e-5)


def test_batchnorm_rules():
    from vivit_amd import kernels
    from vivit_amd.backend.extensions import _jac_t_mat_prod, _param_factor

    g = torch.Generator(device=DEV).manual_seed(3)
    bn = nn.BatchNorm2d(7).to(DEV).eval()
    bn.running_mean.uniform_(-0.5, 0.5)
    bn.running_var.uniform_(0.5, 1.5)
    bn.weight.data.uniform_(0.5, 1.5)
    bn.bias.data.uniform_(-0.5, 0.5)
    x = torch.randn(6, 7, 5, 4, generator=g, device=DEV)
    M = torch.randn(3, *x.shape, generator=g, device=DEV)
    close(_jac_t_mat_prod(bn, M, x), jac_t_by_autograd(bn, x, M))
    xhat = (x - bn.running_mean.view(1, -1, 1, 1)) / torch.sqrt(bn.running_var.view(1, -1, 1, 1) + bn.eps)
    close(_param_factor(bn, "weight", M, x), (M * xhat).flatten(3).sum(3), rtol=1e-4, atol=1e-5)
    close(_param_factor(bn, "bias", M, x), M.flatten(3).sum(3), rtol=1e-4, atol=1e-5)
    scale = bn.weight.detach() / torch.sqrt(bn.running_var + bn.eps)
    close(kernels.channel_scale(M, scale), M * scale.view(1, 1, -1, 1, 1))


@pytest.mark.parametrize("N,C", [(8, 10), (33, 100), (5, 1000)])
def test_cross_entropy_factors(N, C):
    from vivit_amd import kernels

    g = torch.Generator(device=DEV).manual_seed(4)
    logits = torch.randn(N, C, generator=g, device=DEV) * 3
    S = kernels.ce_sqrt_hessian(logits, 1.0 / math.sqrt(N))
    ref = oracle.loss_hessian_sqrt_exact(logits.cpu().double(), "ce")
    close(S.cpu().double(), ref, rtol=1e-4, atol=1e-6)
    # S S^T (over the slices) is the Hessian of the mean loss: diag(p) - p p^T, per sample
    p = logits.double().softmax(1)
    H = torch.einsum("vnc,vnd->ncd", S.double(), S.double())
    close(H, (torch.diag_embed(p) - p.unsqueeze(2) * p.unsqueeze(1)) / N, rtol=1e-4, atol=1e-6)
    # sampled factor with supplied one-hots
    idx = torch.multinomial(logits.softmax(1), 3, replacement=True, generator=g)
    onehot = F.one_hot(idx.t(), C).float()
    Smc = kernels.ce_sqrt_hessian(logits, 1.0 / math.sqrt(3 * N), onehot=onehot)
    close(Smc.cpu().double(), oracle.loss_hessian_sqrt_mc(logits.cpu().double(), onehot.cpu().double()), rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("n", [1, 5, 256, 257, 1000])
def test_pack_unpack_lower(n):
    from vivit_amd import kernels

    g = torch.Generator(device=DEV).manual_seed(5)
    A = torch.randn(n, n, generator=g, device=DEV)
    Gm = A + A.T
    packed = kernels.pack_lower(Gm)
    i, j = torch.tril_indices(n, n, device=DEV)
    assert torch.equal(packed, Gm[i, j])
    out = torch.full((n, n), float("nan"), device=DEV)
    kernels.unpack_lower_(packed, out)
    assert torch.equal(out, Gm)
    # a padded leading dimension on the source
    big = torch.zeros(n, n + 3, device=DEV)
    big[:, :n] = Gm
    assert torch.equal(kernels.pack_lower(big[:, :n]), packed)
