"""vivit_amd: MI355X-native low-rank GGN curvature path (Gram build + symmetric eigensolver)."""
__version__ = "0.1.0"
