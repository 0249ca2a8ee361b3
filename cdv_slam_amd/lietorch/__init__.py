"""lietorch operator surface (reference: cdvslam/lietorch/__init__.py:2).  SO3 and SE3 forward ops on
the HIP backend; RxSO3 / Sim3 and all backward ops belong to training / loop closure (out of scope)."""
from .groups import SE3, SO3, LieGroup, cat, stack

__all__ = ["SE3", "SO3", "LieGroup", "cat", "stack"]
