"""Import-compatible stand-ins for /root/reference/modeling/sync_batchnorm/replicate.py:47-88.

Every trainer of the reference imports and calls `patch_replication_callback(model)` after wrapping the
model in nn.DataParallel (e.g. train.py:11,109).  Parallelism here is one process per GPU, where BN
statistics are synchronised by an all-reduce inside the layer, so there is nothing to patch: the call is
accepted and does nothing on a single-device wrapper, and refuses a multi-device DataParallel, which
would silently train with unsynchronised statistics.
"""
from torch.nn.parallel.data_parallel import DataParallel

__all__ = ["CallbackContext", "execute_replication_callbacks", "DataParallelWithCallback",
           "patch_replication_callback"]


class CallbackContext(object):
    pass


def execute_replication_callbacks(modules):
    if len(modules) > 1:
        raise RuntimeError("thread-replicated SyncBN is replaced by one process per GPU; launch with torchrun")


def _refuse_multi_device(dp):
    ids = getattr(dp, "device_ids", None) or []
    if len(ids) > 1:
        raise RuntimeError("nn.DataParallel over %d devices is not supported: run one process per GPU "
                           "(python -m torch.distributed.run --nproc-per-node N ...)" % len(ids))


class DataParallelWithCallback(DataParallel):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        _refuse_multi_device(self)


def patch_replication_callback(data_parallel):
    assert isinstance(data_parallel, DataParallel)
    _refuse_multi_device(data_parallel)
