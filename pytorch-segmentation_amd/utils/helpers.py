"""Constructor-time helpers of the plugin surface (reference: utils/helpers.py:12-22,44-57).

Only what model constructors need is kept: weight initialisation of decoder heads and the
`set_trainable` freeze helper.  Visualisation helpers (colorize_mask, get_upsampling_weight) are
outside the training hot path.
"""
import os

import torch.nn as nn


def dir_exists(path):
    os.makedirs(path, exist_ok=True)


def initialize_weights(*models):
    """Kaiming-normal conv filters, BN gamma=1 / beta=1e-4, tiny-normal linear layers."""
    for model in models:
        for m in model.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight.data, nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1.0)
                m.bias.data.fill_(1e-4)
            elif isinstance(m, nn.Linear):
                m.weight.data.normal_(0.0, 0.0001)
                m.bias.data.zero_()


def set_trainable(modules, flag):
    """Recursively set requires_grad (and a `.trainable` attribute) on modules / lists of modules."""
    stack = list(modules) if isinstance(modules, (list, tuple)) else [modules]
    while stack:
        m = stack.pop()
        if isinstance(m, (list, tuple)):
            stack.extend(m)
            continue
        m.trainable = flag
        for p in m.parameters(recurse=False):
            p.requires_grad = flag
        stack.extend(m.children())
