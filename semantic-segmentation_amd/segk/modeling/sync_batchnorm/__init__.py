"""Same import surface as /root/reference/modeling/sync_batchnorm/__init__.py."""
from .batchnorm import SynchronizedBatchNorm1d, SynchronizedBatchNorm2d, SynchronizedBatchNorm3d
from .replicate import DataParallelWithCallback, patch_replication_callback

__all__ = ["SynchronizedBatchNorm1d", "SynchronizedBatchNorm2d", "SynchronizedBatchNorm3d",
           "DataParallelWithCallback", "patch_replication_callback"]
