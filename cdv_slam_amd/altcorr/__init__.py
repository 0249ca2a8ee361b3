"""altcorr operator surface (reference: cdvslam/altcorr/__init__.py, correlation.py:51-75)."""
from .correlation import corr, patchify

__all__ = ["corr", "patchify"]
