"""``ViViTGGN{Exact,MC}``: functional access to ``V``, ``V^T`` and the Gram matrix
(vivit/extensions/secondorder/vivit/__init__.py:136-181), provided by the stand-in backend."""
from vivit_amd.backend.extensions import ViViTGGNExact, ViViTGGNMC

__all__ = ["ViViTGGNExact", "ViViTGGNMC"]
