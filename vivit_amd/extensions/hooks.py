"""Stand-alone extension hooks (API of ``vivit.extensions.hooks``).

``GramSqrtGGN{Exact,MC}`` turn the materialised sqrt-GGN factors into the accumulated
``[NC, NC]`` Gram matrix (the simplest entry to the Gram kernel);
``GramBatchGrad`` / ``CenteredGramBatchGrad`` / ``CenteredBatchGrad`` do the same for
per-sample gradients.  All hooks are single-use per backward pass.
"""
from vivit_amd.utils.gram import pairwise_dot
from vivit_amd.utils.hooks import ParameterHook


class _GramAccumulator(ParameterHook):
    """Sum the Gram of ``param.<source>`` over all visited parameters, in-kernel (beta = 1)."""

    def __init__(self, savefield, source, start_dim, layerwise, free_source):
        super().__init__(savefield)
        self._source = source
        self._start_dim = start_dim
        self._layerwise = layerwise
        self._free_source = free_source
        self._gram_mat = None

    def _prepare(self, factor):
        return factor

    def param_hook(self, param):
        factor = self._prepare(getattr(param, self._source))
        layer_gram = None
        if self._layerwise or self._gram_mat is None:
            layer_gram = pairwise_dot(factor, start_dim=self._start_dim).detach()
            if self._gram_mat is None:
                self._gram_mat = layer_gram.clone() if self._layerwise else layer_gram
            else:
                self._gram_mat += layer_gram
        else:
            n = self._gram_mat.shape[0]
            lead = factor.shape[: self._start_dim]
            pairwise_dot(factor, start_dim=self._start_dim, flatten=False,
                         out=self._gram_mat.view(*lead, *lead), beta=1.0)
            assert self._gram_mat.shape == (n, n)
        if self._free_source:
            delattr(param, self._source)
        if self._layerwise:
            return layer_gram

    def get_result(self):
        """The accumulated Gram matrix after the backward pass."""
        return self._gram_mat


class GramSqrtGGNExact(_GramAccumulator):
    """Gram matrix ``[CN, CN]`` of the exact GGN factors ``param.sqrt_ggn_exact``
    (vivit/extensions/secondorder/sqrt_ggn/gram_sqrt_ggn.py:77-107)."""

    def __init__(self, savefield="gram_sqrt_ggn_exact", layerwise=False, free_sqrt_ggn=False):
        super().__init__(savefield, "sqrt_ggn_exact", 2, layerwise, free_sqrt_ggn)


class GramSqrtGGNMC(_GramAccumulator):
    """Gram matrix ``[MN, MN]`` of the MC GGN factors ``param.sqrt_ggn_mc``
    (gram_sqrt_ggn.py:110-142)."""

    def __init__(self, savefield="gram_sqrt_ggn_mc", layerwise=False, free_sqrt_ggn=False):
        super().__init__(savefield, "sqrt_ggn_mc", 2, layerwise, free_sqrt_ggn)


class CenteredBatchGrad(ParameterHook):
    """Store ``grad_batch - grad_batch.mean(0)`` under ``savefield``
    (vivit/extensions/firstorder/batch_grad/gram_batch_grad.py:7-37)."""

    _SAVEFIELD_GRAD_BATCH = "grad_batch"

    def __init__(self, savefield="centered_grad_batch"):
        super().__init__(savefield)

    def param_hook(self, param):
        grad_batch = getattr(param, self._SAVEFIELD_GRAD_BATCH)
        return grad_batch - grad_batch.mean(0)


class _GramBatchGradBase(_GramAccumulator):
    _SAVEFIELD_GRAD_BATCH = "grad_batch"

    def __init__(self, savefield, center, layerwise=False, free_grad_batch=False):
        super().__init__(savefield, self._SAVEFIELD_GRAD_BATCH, 1, layerwise, free_grad_batch)
        self._center = center

    def _prepare(self, factor):
        if self._center:
            factor -= factor.mean(0)  # in place, as gram_batch_grad.py:96-97
        return factor


class GramBatchGrad(_GramBatchGradBase):
    """Un-centred gradient Gram matrix ``[N, N]`` (gram_batch_grad.py:120-164)."""

    def __init__(self, savefield="gram_grad_batch", layerwise=False, free_grad_batch=False):
        super().__init__(savefield, center=False, layerwise=layerwise, free_grad_batch=free_grad_batch)


class CenteredGramBatchGrad(_GramBatchGradBase):
    """Centred gradient Gram matrix ``[N, N]`` (gram_batch_grad.py:167-213)."""

    def __init__(self, savefield="centered_gram_grad_batch", layerwise=False, free_grad_batch=False):
        super().__init__(savefield, center=True, layerwise=layerwise, free_grad_batch=free_grad_batch)
