"""Oracle: Adam (coupled L2, torch.optim.Adam semantics) and the ema_pytorch-style EMA.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Adam call sites in the reference: ddpm.py:1053-1059, vqvae.py:207-214, wgan.py:183-195.
The update below is torch's documented single-tensor Adam (no amsgrad, no maximize,
eps=1e-8 default): the tests pin it against ``torch.optim.Adam`` itself.

EMA: ``ema_pytorch.EMA`` is a third-party, UNPINNED dependency of the reference
(environments/requirements.txt:20; call sites ddpm.py:998,1048) and is not vendored
under /root/reference nor installed here -> **parity unpinned** for the EMA schedule.
The restatement follows its published algorithm: step counter,
``update_every``, hard copy while ``step <= update_after_step`` (100), afterwards
``shadow.lerp_(online, 1 - decay)`` with
``decay = clamp(1 - (1 + max(step - update_after_step - 1, 0)/inv_gamma) ** -power, 0, beta)``,
inv_gamma = 1, power = 2/3.
"""
from __future__ import annotations

import math

import torch


def adam_step(p, g, m, v, step: int, lr: float, b1: float, b2: float,
              eps: float = 1e-8, weight_decay: float = 0.0):
    """One Adam update; ``step`` is the 1-based step count AFTER increment.
    Returns (p', m', v')."""
    if weight_decay != 0.0:
        g = g + weight_decay * p
    m = m + (g - m) * (1 - b1)           # torch: exp_avg.lerp_(grad, 1-beta1)
    v = v * b2 + (1 - b2) * g * g
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    step_size = lr / bc1
    denom = v.sqrt() / math.sqrt(bc2) + eps
    return p - step_size * (m / denom), m, v


def rmsprop_step(p, g, sq, lr: float, alpha: float = 0.99, eps: float = 1e-8, weight_decay: float = 0.0):
    """torch.optim.RMSprop with its defaults (momentum 0, not centred) - reference wgan.py:171-181.
    Returns (p', sq')."""
    if weight_decay != 0.0:
        g = g + weight_decay * p
    sq = sq * alpha + (1 - alpha) * g * g
    return p - lr * g / (sq.sqrt() + eps), sq


def ema_decay(step: int, beta: float, update_after_step: int = 100,
              inv_gamma: float = 1.0, power: float = 2.0 / 3.0, min_value: float = 0.0) -> float:
    epoch = max(step - update_after_step - 1, 0)
    if epoch <= 0:
        return 0.0
    value = 1 - (1 + epoch / inv_gamma) ** -power
    return min(max(value, min_value), beta)


class EmaState:
    """Scalar state machine of ema_pytorch.EMA.update(); tensors handled by the caller."""

    def __init__(self, beta=0.995, update_every=10, update_after_step=100):
        self.beta, self.update_every, self.update_after_step = beta, update_every, update_after_step
        self.step = 0
        self.initted = False

    def next_action(self):
        """Advance one call of update(); returns ('skip'|'copy'|'lerp', weight)."""
        step = self.step
        self.step += 1
        if step % self.update_every != 0:
            return "skip", None
        if step <= self.update_after_step:
            return "copy", None
        if not self.initted:
            # upstream copies and then falls through to a lerp of two equal tensors
            self.initted = True
            return "copy", None
        # get_current_decay() reads the already-incremented counter
        d = ema_decay(self.step, self.beta, self.update_after_step)
        return "lerp", 1.0 - d


def ema_apply(shadow: torch.Tensor, online: torch.Tensor, action: str, w):
    if action == "copy":
        return online.clone()
    if action == "lerp":
        return shadow + (online - shadow) * w
    return shadow
