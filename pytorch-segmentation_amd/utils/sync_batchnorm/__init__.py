"""Synchronized BatchNorm plugin surface (reference: utils/sync_batchnorm/__init__.py:11-13).

Same public names; the implementation is one process per GPU with RCCL collectives
(segmi.distributed) instead of the reference's thread/queue master-slave pipe inside nn.DataParallel
(utils/sync_batchnorm/comm.py).  Only the 2-D variant is on the segmentation hot path.
"""
from .batchnorm import SynchronizedBatchNorm2d, convert_model, patch_sync_batchnorm  # noqa: F401
from .replicate import DataParallelWithCallback, patch_replication_callback  # noqa: F401
