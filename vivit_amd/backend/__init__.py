"""Self-contained stand-in for the BackPACK protocol the ViViT hooks are driven by.

``backpack-for-pytorch`` (>=1.5,<2; reference setup.cfg:36) is not available on the build or the
GPU box, so the boundary the path sits behind -- ``extend(model)``, ``with backpack(*extensions,
extension_hook=hook): loss.backward()``, quantities attached to parameters under an
extension's ``savefield``, ``hook(module)`` called per module after its extensions ran -- is
provided here for feed-forward and residual nets (Linear, Conv1d/2d/3d, ConvTranspose1d/2d/3d, BatchNorm in
eval mode, element-wise activations, pooling, Flatten, Dropout in eval mode, and the branching modules
``Parallel`` / ``SumModule`` / ``Pad`` / ``Slicing`` / ``ScaleModule`` of ``backpack.custom_module``;
CrossEntropyLoss / MSELoss with ``reduction='mean'``).  When the real BackPACK is importable the Computation classes can be used
with it directly; this module is never imported in that case.

This is the FACTOR PROVIDER, i.e. the producer of the path's inputs
(``V_t``, ``grad_batch``), not the measured hot path; it uses torch ops for the per-layer
Jacobian rules and the HIP kernels only where a rule is GEMM-shaped.
"""
from vivit_amd.backend.custom_module import ActiveIdentity, Pad, Parallel, ScaleModule, Slicing, SumModule
from vivit_amd.backend.engine import backpack, extend
from vivit_amd.backend.extensions import (
    BatchGrad,
    SqrtGGNExact,
    SqrtGGNMC,
    ViViTGGNExact,
    ViViTGGNMC,
)

__all__ = ["ActiveIdentity", "Pad", "Parallel", "ScaleModule", "Slicing", "SumModule", "backpack", "extend", "BatchGrad", "SqrtGGNExact", "SqrtGGNMC", "ViViTGGNExact", "ViViTGGNMC"]
