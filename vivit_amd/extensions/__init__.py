"""Extension hooks and factor-providing extensions (mirror of ``vivit.extensions``)."""
from vivit_amd.extensions import hooks

__all__ = ["hooks"]
