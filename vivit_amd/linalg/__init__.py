"""Linear algebra with the low-rank GGN on MI355X (mirror of ``vivit.linalg``)."""
from vivit_amd.linalg.eigh import EighComputation
from vivit_amd.linalg.eigvalsh import EigvalshComputation

__all__ = ["EighComputation", "EigvalshComputation"]
