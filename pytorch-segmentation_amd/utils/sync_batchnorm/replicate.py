"""`DataParallelWithCallback` / `patch_replication_callback` of the reference
(utils/sync_batchnorm/replicate.py:50-94) re-hosted on one-process-per-GPU data parallelism.

The reference subclasses nn.DataParallel so that each replicated SyncBN module learns its replica id.
Here every rank owns exactly one replica on its own GPU, so the wrapper is
segmi.distributed.DistributedModel: `.module` access (base/base_trainer.py:47-51), gradient
all-reduce over RCCL instead of DataParallel's gather/reduce.  `device_ids` is accepted and ignored
(the launcher assigns one GPU per process: LOCAL_RANK).
"""
from segmi.distributed import DistributedModel


class DataParallelWithCallback(DistributedModel):
    def __init__(self, module, device_ids=None, process_group=None):
        super().__init__(module, process_group)
        self.device_ids = device_ids


def patch_replication_callback(data_parallel):
    """No-op: there is no in-process replication to hook (kept for source compatibility)."""
    return data_parallel
