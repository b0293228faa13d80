"""segmi — Python binding of the MI355X kernel library (libsegmi.so) behind the drop-in modules.

    segmi.ops   functional ops + autograd glue (NHWC-backed tensors)
    segmi.nn    nn.Module subclasses with reference-compatible state_dict keys
    segmi.optim fused multi-tensor SGD (torch.optim.SGD semantics and state layout)
    segmi.graph hipGraph capture / replay of a whole training step
"""
from ._lib import LIB_PATH, SegmiError, lib  # noqa: F401  (raises if libsegmi.so is missing)
from . import ops, nn, optim, graph  # noqa: F401

__version__ = "0.1.0"
