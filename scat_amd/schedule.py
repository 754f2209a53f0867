"""Host-side training glue the reference takes from outside its own tree.

* ``warmup_lr`` restates what ``train.py:61-63,134`` asks of the third-party ``warmup_scheduler`` package
  (``GradualWarmupScheduler(optimizer, multiplier=1, total_epoch=15, after_scheduler=StepLR(step_size=10, gamma=1))``,
  stepped with ``epoch + 1``): with multiplier 1 the package ramps the rate linearly from 0 to the base rate over
  ``total_epoch`` epochs and then hands over to the after-scheduler, whose gamma of 1 keeps it flat.  The package is
  not pinned by the reference (no version in requirements.txt) and is absent here: **parity unpinned** — this is a
  host scalar, off the kernel path.
* ``load_pretrained_backbone`` ingests an ImageNet ResNet checkpoint from a LOCAL file the way
  ``models/resnet.py:192-195`` ingests the downloaded one: ``load_state_dict(strict=False)`` — every conv/bn key
  matches, the classifier ``fc.*`` has no counterpart (the reference's head is ``fc1`` 2048->1024, resnet.py:116) and
  is ignored, ``fc1`` keeps its initialisation.
"""
from __future__ import annotations

import torch


def warmup_lr(base_lr: float, epoch: int, total_epoch: int = 15, multiplier: float = 1.0) -> float:
    """Learning rate in effect after ``scheduler_warmup.step(epoch)`` (train.py calls it with epoch + 1)."""
    if epoch > total_epoch:
        return base_lr * multiplier
    if multiplier == 1.0:
        return base_lr * (float(epoch) / total_epoch)
    return base_lr * ((multiplier - 1.0) * epoch / total_epoch + 1.0)


class WarmupSchedule:
    """Drop-in for the two call sites: ``WarmupSchedule(optimizer, 15).step(epoch + 1)``.  Works with any object that
    has ``param_groups`` (torch optimizers) or an ``lr`` attribute (scat_amd.trainer.TrainStep's fused Adam)."""

    def __init__(self, optimizer, total_epoch=15, multiplier=1.0):
        self.opt, self.total, self.mult = optimizer, total_epoch, multiplier
        groups = getattr(optimizer, "param_groups", None)
        self.base = [g["lr"] for g in groups] if groups is not None else [optimizer.lr]

    def step(self, epoch):
        groups = getattr(self.opt, "param_groups", None)
        if groups is not None:
            for g, b in zip(groups, self.base):
                g["lr"] = warmup_lr(b, epoch, self.total, self.mult)
        else:
            self.opt.lr = warmup_lr(self.base[0], epoch, self.total, self.mult)


def load_pretrained_backbone(backbone: torch.nn.Module, path: str):
    """``backbone``: scat_amd.models.resnet.ResNet (e.g. ``net.main_encoder``); ``path``: a torchvision-style ResNet
    ``state_dict`` saved with torch.save.  Returns (missing_keys, unexpected_keys) like load_state_dict."""
    sd = torch.load(path, map_location="cpu")
    if "state_dict" in sd and isinstance(sd["state_dict"], dict):
        sd = sd["state_dict"]
    sd = {k[len("module."):] if k.startswith("module.") else k: v for k, v in sd.items()}
    own = backbone.state_dict()
    sd = {k: v for k, v in sd.items() if k not in own or own[k].shape == v.shape}   # drop shape clashes, keep the rest
    return backbone.load_state_dict(sd, strict=False)
