"""Second-order extensions (mirror of ``vivit.extensions.secondorder``)."""
from vivit_amd.extensions.secondorder import vivit
from vivit_amd.extensions.secondorder.vivit import ViViTGGNExact, ViViTGGNMC

__all__ = ["vivit", "ViViTGGNExact", "ViViTGGNMC"]
