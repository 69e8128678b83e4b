"""``vivit.extensions.secondorder.vivit``: ``ViViTGGN{Exact,MC}``
(vivit/extensions/secondorder/vivit/__init__.py:136-181).  The closures they attach to the parameters
(``gram_mat`` / ``V_mat_prod`` / ``V_t_mat_prod``) run on the HIP kernels; the sqrt-GGN back-propagation that
feeds them is the stand-in backend's (BackPACK is absent on the build and the GPU box)."""
from vivit_amd.backend.extensions import ViViTGGNExact, ViViTGGNMC

__all__ = ["ViViTGGNExact", "ViViTGGNMC"]
