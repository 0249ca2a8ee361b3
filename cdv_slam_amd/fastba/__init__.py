"""fastba operator surface (reference: cdvslam/fastba/__init__.py, ba.py:4-8)."""
from .ba import BA, neighbors, reproject

__all__ = ["BA", "neighbors", "reproject"]
