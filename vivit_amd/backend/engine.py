"""``extend`` / ``with backpack(...)``: run extensions and the extension hook during backward.

Mechanism: every extended leaf module records ``input0`` / ``output`` in a forward hook and
registers a tensor hook on its output.  During ``loss.backward()`` autograd calls those tensor
hooks in reverse execution order with the gradient w.r.t. the module output, which is exactly
the point at which BackPACK runs its extensions and then the user's ``extension_hook(module)``.
Tensor identity is preserved between consecutive modules (``prev.output is next.input0``), which
is how back-propagated quantities (the sqrt-GGN factor) are handed from a module to its
predecessor.
"""
from typing import Callable, Optional

import torch
from torch.nn import Module

_ACTIVE = None  # the innermost active ``backpack`` context


class backpack:
    """Context manager activating ``extensions`` (and an optional hook) for backward passes.

    Mirrors ``backpack.backpack(*exts, extension_hook=None)`` as used by the reference
    (docs/examples/basic_usage/example_eigh.py:103-107).
    """

    def __init__(self, *extensions, extension_hook: Optional[Callable[[Module], None]] = None):
        self.extensions = extensions
        self.extension_hook = extension_hook
        self.state = {}  # (id(extension), id(tensor)) -> quantity back-propagated to that tensor
        self._keepalive = []
        self._outer = None

    def __enter__(self):
        global _ACTIVE
        self._outer = _ACTIVE
        _ACTIVE = self
        return self

    def __exit__(self, *exc):
        global _ACTIVE
        _ACTIVE = self._outer
        self.state.clear()
        self._keepalive.clear()
        return False

    # back-propagated quantities are keyed by the identity of the activation tensor
    def put(self, ext, tensor, value):
        key = (id(ext), id(tensor))
        if key in self.state:  # fan-out: contributions add (accumulate_backpropagated_quantities)
            value = self.state[key] + value
        self.state[key] = value
        self._keepalive.append(tensor)

    def pop(self, ext, tensor):
        return self.state.pop((id(ext), id(tensor)), None)


def _make_output_hook(module: Module):
    def on_grad(grad):
        ctx = _ACTIVE
        if ctx is None:
            return None
        for ext in ctx.extensions:
            ext.apply(ctx, module, grad)
        if ctx.extension_hook is not None:
            ctx.extension_hook(module)
        return None

    return on_grad


def _forward_hook(module: Module, inputs, output):
    if not torch.is_grad_enabled() or not isinstance(output, torch.Tensor) or not output.requires_grad:
        return None
    replaced = None
    if any(output is t for t in inputs):
        # a module that hands its input through (nn.Identity, Dropout in eval mode): give the output its own identity,
        # otherwise the producer of that tensor would see its hook fire before this module contributed
        output = replaced = output.view_as(output)
    # what BackPACK stores for its extensions (vivit/linalg/utils.py:54 reads ``input0``)
    module.input0 = inputs[0]
    if len(inputs) > 1:
        module.input1 = inputs[1]
    module.inputs = tuple(inputs)
    module.output = output
    output.register_hook(_make_output_hook(module))
    return replaced


def extend(module: Module) -> Module:
    """Register the bookkeeping hooks on all leaf sub-modules of ``module`` (idempotent)."""
    for m in module.modules():
        if len(list(m.children())) > 0:
            continue  # containers: their children do the work
        if getattr(m, "_vivit_amd_extended", False):
            continue
        m.register_forward_hook(_forward_hook)
        m._vivit_amd_extended = True
    return module
