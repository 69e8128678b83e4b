"""Import-guarded adapter for the REAL BackPACK (backpack-for-pytorch >= 1.5, < 2): ``SqrtGGN{Exact,MC}`` whose Linear
weights stay factorised, as the reference's ``ViViTGGNLinear`` keeps them
(vivit/extensions/secondorder/vivit/linear.py:41-81) -- ``V_t[c,n,o,i] = s[c,n,o] z[n,i]`` is never materialised, the
hooks get the closures ``gram_mat`` / ``V_mat_prod`` / ``V_t_mat_prod`` on the HIP kernels.

BackPACK is absent from this image and from the GPU box, so NOTHING here can be exercised by the tests:
the module is written against the public structure of BackPACK 1.5's ``sqrt_ggn`` package as the reference uses it
(vivit/extensions/secondorder/vivit/__init__.py:81-120, base.py:84-92) and is deliberately defensive -- any mismatch
makes :func:`factorised_sqrt_ggn` return ``None`` and the caller falls back to BackPACK's own (materialising)
``SqrtGGN{Exact,MC}``, which is today's behaviour.
"""
import warnings
from typing import Optional


def factorised_sqrt_ggn(mc_samples: int, subsampling) -> Optional[object]:
    """A BackPACK ``SqrtGGN{Exact,MC}`` instance whose ``nn.Linear`` module extension stores closures for the weight,
    or ``None`` if BackPACK (or the parts of it this needs) cannot be found."""
    try:
        from backpack.extensions import SqrtGGNExact, SqrtGGNMC
        from backpack.extensions.secondorder.sqrt_ggn.linear import SqrtGGNLinear
        from backpack.utils.subsampling import subsample
        from torch.nn import Linear

        from vivit_amd.backend.extensions import _linear_weight_closures

        class FactorisedSqrtGGNLinear(SqrtGGNLinear):
            """``weight`` returns the closure dict instead of ``param_mjp(..., sum_batch=False)`` (base.py:84-92)."""

            def weight(self, ext, module, g_inp, g_out, backproped):
                if backproped.dim() != 3 or not backproped.is_cuda:      # extra input dimensions / CPU: BackPACK's own rule
                    return super().weight(ext, module, g_inp, g_out, backproped)
                z = subsample(module.input0, subsampling=ext.get_subsampling())
                return _linear_weight_closures(backproped.detach(), z.detach())

        ext = SqrtGGNExact(subsampling=subsampling) if mc_samples == 0 else SqrtGGNMC(mc_samples=mc_samples, subsampling=subsampling)
        # the module-extension table of a BackpropExtension (name-mangled private attribute in BackPACK 1.x)
        table = None
        for name in ("_BackpropExtension__module_extensions", "_module_extensions", "module_extensions"):
            table = getattr(ext, name, None)
            if isinstance(table, dict):
                break
        if not isinstance(table, dict) or Linear not in table:
            raise AttributeError("module-extension table of BackpropExtension not found")
        table[Linear] = FactorisedSqrtGGNLinear()
        return ext
    except Exception as exc:  # noqa: BLE001 -- optional third-party integration: never fatal
        try:
            import backpack  # noqa: F401
        except Exception:
            return None       # BackPACK not installed: nothing to report
        warnings.warn(f"vivit_amd: BackPACK found but its Linear extension could not be made factorised ({exc!r}); "
                      "falling back to BackPACK's materialised SqrtGGN factors")
        return None
