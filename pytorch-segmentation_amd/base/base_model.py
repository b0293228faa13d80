"""BaseModel — the plugin base class every architecture derives from.

Same contract as the reference's base/base_model.py:6-22: a `logger`, `summary()` that logs the number
of trainable parameters and a `__str__` that appends it (train.py:27 prints the model through it).
"""
import logging

import torch.nn as nn


def _count_trainable(module):
    return sum(p.numel() for p in module.parameters() if p.requires_grad)


class BaseModel(nn.Module):
    def __init__(self):
        super().__init__()
        self.logger = logging.getLogger(self.__class__.__name__)

    def forward(self, *inputs):
        raise NotImplementedError

    def summary(self):
        self.logger.info("Nbr of trainable parameters: %d" % _count_trainable(self))

    def __str__(self):
        return super().__str__() + "\nNbr of trainable parameters: %d" % _count_trainable(self)
