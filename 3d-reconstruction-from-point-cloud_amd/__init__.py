"""MI355X-native point-based detail transfer (k-NN attribute transfer) -- host-side package.

Holds only what the hot path needs: csrc/ (gfx950 HIP kernels + the C ABI), capi.py (ctypes
binding), transfer.py (host mirror of the reference's build/query/blend steps), sharding.py
(spatial-slab multi-GPU protocol over torch.distributed) and host/ (the C++ pointsTransfer CLI).
The directory name is not a Python identifier; load it with `__graft_entry__.load_package()`.
"""
from . import capi
from .capi import F32, F16, F64, NOIDX, MAX_K, BLEND_MEAN, BLEND_INV_D2, PtError
from .transfer import PointsTransfer, POINT_DTYPE, K_REFERENCE

__all__ = ["capi", "PointsTransfer", "POINT_DTYPE", "K_REFERENCE", "F32", "F16", "F64", "NOIDX", "MAX_K",
           "BLEND_MEAN", "BLEND_INV_D2", "PtError"]
