"""MI355X-native UNet segmentation train-step path (drop-in for the hot path of LorenzoFramba/Continual-Learning).

Import as ``continual_learning_amd`` (the directory name has a hyphen; ``continual_learning_amd.py`` at the repository
root is the import shim).  Everything compute-related goes through libclamd.so (hand-written HIP for gfx950).
"""
from . import _lib, ops, synth  # noqa: F401
from .unet import UNet, cpad, stage_table  # noqa: F401
from .loss import CrossEntropyLoss, DistillationCrossEntropy  # noqa: F401
from .optim import FusedAdam  # noqa: F401
from .metrics import argmax_confusion, eval_metrics, metrics_from_confusion  # noqa: F401
from .trainer import Trainer, default_config  # noqa: F401
from . import data, ddp  # noqa: F401

__all__ = ['UNet', 'CrossEntropyLoss', 'DistillationCrossEntropy', 'FusedAdam', 'Trainer', 'default_config',
           'argmax_confusion', 'eval_metrics', 'metrics_from_confusion', 'data', 'ddp', 'synth']
