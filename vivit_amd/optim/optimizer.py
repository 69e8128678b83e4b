"""A ``torch.optim``-style loop around the directionally damped Newton step.

The reference has no optimizer class; its example applies the step by hand after the backward pass
(docs/examples/basic_usage/example_directional_damped_newton.py:144-187).  This wrapper keeps that data flow -- one
``with backpack(...)`` backward pass per step whose extension hook computes the step, then ``p += lr * step`` -- behind
the usual ``Optimizer`` interface (``param_groups`` carry ViViT's ``'criterion'`` and ``'damping'`` callbacks)."""
from typing import Callable, Dict, Iterable, Optional

import torch

from vivit_amd.optim.directional_damped_newton import DirectionalDampedNewtonComputation


class DirectionalDampedNewton(torch.optim.Optimizer):
    """``optimizer.step(closure)``: ``closure()`` must run the forward pass on an ``extend``-ed model and return the
    (mean-reduced) loss WITHOUT calling ``backward``; the optimizer back-propagates it inside the ``backpack`` context
    with its own extensions and hook, then applies ``p += lr * newton_step`` for every group.

    ``params``: iterable of parameters or of dicts with ``'params'`` (+ optional ``'criterion'``, ``'damping'``, ``'lr'``);
    ``backpack``: the context manager to use (``vivit_amd.backend.backpack`` or BackPACK's own);
    remaining keyword arguments go to :class:`DirectionalDampedNewtonComputation`
    (``subsampling_grad``, ``subsampling_ggn``, ``mc_samples_ggn``, ``factorised``, ``data_parallel``, ...)."""

    def __init__(self, params: Iterable, criterion: Callable, damping: Callable, backpack, lr: float = 1.0,
                 **computation_kwargs):
        if lr <= 0.0:
            raise ValueError(f"Invalid learning rate: {lr}")
        super().__init__(params, dict(lr=lr, criterion=criterion, damping=damping))
        self._backpack = backpack
        self._kwargs = computation_kwargs
        self.last_steps: Dict[int, list] = {}

    @torch.no_grad()
    def step(self, closure: Optional[Callable] = None):
        if closure is None:
            raise ValueError("DirectionalDampedNewton.step needs a closure that returns the loss (forward pass only)")
        computation = DirectionalDampedNewtonComputation(**self._kwargs)  # hooks are single-use per backward pass
        with torch.enable_grad():
            loss = closure()
            with self._backpack(*computation.get_extensions(),
                                extension_hook=computation.get_extension_hook(self.param_groups)):
                loss.backward()
        for group in self.param_groups:
            steps = computation.get_result(group)
            self.last_steps[id(group)] = steps
            for p, s in zip(group["params"], steps):
                p.add_(s.to(p.dtype), alpha=group["lr"])
        return loss.detach()
