"""LightningModule / Trainer surface used by the hot path.

If ``pytorch_lightning`` is importable it is used unchanged (the models are ordinary
LightningModules).  It is not installed in the build image, so this file also provides a
minimal stand-in with the subset of behaviour the reference relies on
(train.py:124-141, ddpm.py:983,1017-1027,1047, wgan.py:58-82, vqvae.py:184-194):

  save_hyperparameters / hparams, log / log_dict (kept in ``logged``), global_step (counts
  optimizer.step() calls, which is what makes WGAN's n_critic schedule work — wgan.py:64),
  optimizers(), manual_backward(), automatic vs manual optimisation, on_train_batch_end,
  one process per GPU with gradient averaging over torch.distributed (RCCL on ROCm).
"""
from __future__ import annotations

import inspect
import os
import time
from typing import Any, Dict, Iterable, List, Optional

import torch
import torch.distributed as dist
from torch import nn

try:  # pragma: no cover - not available in the build image
    import pytorch_lightning as _pl  # type: ignore
    HAVE_PL = True
except Exception:  # noqa
    _pl = None
    HAVE_PL = False


class _AttrDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    __setattr__ = dict.__setitem__


class _CountingOptimizer:
    """Proxy that advances the module's global_step on every step() (Lightning semantics)."""

    def __init__(self, opt, owner):
        self._opt, self._owner = opt, owner

    def step(self, *a, **k):
        r = self._opt.step(*a, **k)
        self._owner._global_step += 1
        return r

    def __getattr__(self, n):
        return getattr(self._opt, n)


class MiniLightningModule(nn.Module):
    def __init__(self):
        super().__init__()
        self._hparams = _AttrDict()
        self._global_step = 0
        self.automatic_optimization = True
        self.logged: Dict[str, Any] = {}
        self._optimizers: List[Any] = []
        self.trainer = None
        self.logger = None

    # -- hyper-parameters ---------------------------------------------------------------
    def save_hyperparameters(self):
        frame = inspect.currentframe().f_back
        args = inspect.getargvalues(frame)
        for name in args.args:
            if name != "self":
                self._hparams[name] = args.locals[name]
        if args.keywords and args.keywords in args.locals:
            self._hparams.update(args.locals[args.keywords])

    @property
    def hparams(self):
        return self._hparams

    # -- state --------------------------------------------------------------------------
    @property
    def global_step(self) -> int:
        return self._global_step

    @property
    def device(self):
        for p in self.parameters():
            return p.device
        for b in self.buffers():
            return b.device
        return torch.device("cpu")

    def log(self, name, value, **kw):
        self.logged[name] = value

    def log_dict(self, d, **kw):
        self.logged.update(d)

    def optimizers(self):
        if len(self._optimizers) == 1:
            return self._optimizers[0]
        return self._optimizers

    def manual_backward(self, loss, *a, **k):
        """Lightning semantics under DDP: gradients are averaged over ranks as part of the backward.  Only the
        flat buffers this backward actually wrote (a GAN's critic OR generator) are exchanged."""
        loss.backward(*a, **k)
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        if world > 1:
            for fp in _flat_grads_of(self):
                if not fp.fresh:
                    dist.all_reduce(fp.grad)
                    fp.grad.div_(world)
            for p in self.parameters():
                if getattr(p, "_lgm_flat", None) is None and p.grad is not None:
                    dist.all_reduce(p.grad)
                    p.grad.div_(world)

    # hooks (no-ops by default)
    def on_train_batch_end(self, outputs, batch, batch_idx):
        pass

    def configure_optimizers(self):
        raise NotImplementedError


LightningModule = _pl.LightningModule if HAVE_PL else MiniLightningModule


def _flat_grads_of(module: nn.Module):
    """Distinct FlatParams objects reachable from the module's parameters."""
    seen, out = set(), []
    for p in module.parameters():
        fp = getattr(p, "_lgm_flat", None)
        if fp is not None and id(fp) not in seen:
            seen.add(id(fp))
            out.append(fp)
    return out


class FlatGradSync:
    """Bucketed gradient exchange for one flat gradient buffer, overlapped with the backward pass.

    The hand-written backward fills the flat buffer in a known order; ``ready(lo, hi)`` is called
    as soon as the slice [lo, hi) is final and launches an asynchronous all-reduce on it (RCCL
    runs it on its own stream, ordered after the kernels already enqueued), ``finish()`` waits
    for all buckets.  The 1/world average is folded into the optimiser (``grad_scale``)."""

    def __init__(self, flat, group=None):
        self.flat, self.group = flat, group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.handles = []
        self.covered = 0

    def ready(self, lo: int, hi: int):
        if self.world == 1 or hi <= lo:
            return
        self.handles.append(dist.all_reduce(self.flat.grad[lo:hi], group=self.group, async_op=True))
        self.covered += hi - lo

    def finish(self):
        for h in self.handles:
            h.wait()
        self.handles = []
        if self.world > 1:
            assert self.covered == self.flat.total, \
                f"gradient buckets covered {self.covered} of {self.flat.total} elements"
        self.covered = 0

    @property
    def grad_scale(self) -> float:
        return 1.0 / self.world


def save_checkpoint(model, optimizers, path: str, epoch: int = 0):
    """Write a checkpoint with the layout of a PyTorch-Lightning ``.ckpt`` (reference train.py:41,
    113-117,140 resumes from / writes these): ``state_dict`` with the reference's keys (incl.
    ``ema.online_model.*`` / ``ema.ema_model.*`` / ``ema.initted`` / ``ema.step``), ``global_step``
    (= optimizer steps), ``optimizer_states`` in torch's own per-parameter format, hyper-parameters."""
    torch.save({"epoch": int(epoch), "global_step": int(model.global_step),
                "pytorch-lightning_version": "2.0.0+lgm_hip", "state_dict": model.state_dict(),
                "loops": {}, "callbacks": {}, "optimizer_states": [o.state_dict() for o in optimizers],
                "lr_schedulers": [], "hparams_name": "kwargs", "hyper_parameters": dict(model.hparams)}, path)


class MiniTrainer:
    """Single-node trainer: one process per GPU, optional DDP-style gradient averaging with a
    single all-reduce per flat gradient buffer (RCCL over xGMI when backend is nccl)."""

    def __init__(self, max_steps=-1, max_epochs=-1, accumulate_grad_batches=1, device=None,
                 default_root_dir=None, log_every=50, **_ignored):
        self.max_steps, self.max_epochs = max_steps, max_epochs
        self.accumulate = max(1, int(accumulate_grad_batches))
        self.device = device
        self.root = default_root_dir
        self.log_every = log_every
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank() if self.world > 1 else 0

    def allreduce_grads(self, module):
        if self.world == 1:
            return
        for fp in _flat_grads_of(module):
            dist.all_reduce(fp.grad)
            fp.grad.div_(self.world)
        # parameters that are not flat-bound (CPU plumbing models)
        for p in module.parameters():
            if getattr(p, "_lgm_flat", None) is None and p.grad is not None:
                dist.all_reduce(p.grad)
                p.grad.div_(self.world)

    def fit(self, model, datamodule=None, train_dataloader=None, ckpt_path=None):
        device = torch.device(self.device) if self.device is not None else (
            torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu"))
        model.to(device)
        model.trainer = self
        ckpt = None
        if ckpt_path:
            ckpt = torch.load(ckpt_path, map_location=device, weights_only=False)
            model.load_state_dict(ckpt["state_dict"])
            model._global_step = int(ckpt.get("global_step", 0))
        if hasattr(model, "prepare_hip"):
            model.prepare_hip(device)
        cfg = model.configure_optimizers()
        opts = cfg[0] if isinstance(cfg, tuple) else cfg
        if not isinstance(opts, (list, tuple)):
            opts = [opts]
        if ckpt is not None:
            # resume: Lightning layout, one torch-format optimizer state per optimizer, in order
            for o, osd in zip(opts, ckpt.get("optimizer_states", [])):
                o.load_state_dict(osd)
        model._optimizers = [_CountingOptimizer(o, model) for o in opts]
        loader = train_dataloader if train_dataloader is not None else datamodule.train_dataloader()
        model.train()
        epoch, done = 0, False
        t0 = time.time()
        takes_idx = "batch_idx" in inspect.signature(model.training_step).parameters
        while not done:
            for batch_idx, batch in enumerate(loader):
                batch = tuple(b.to(device, non_blocking=True) if torch.is_tensor(b) else b for b in batch)
                if model.automatic_optimization:
                    opt = model._optimizers[0]
                    loss = model.training_step(batch, batch_idx) if takes_idx else model.training_step(batch)
                    if self.accumulate > 1:
                        loss = loss / self.accumulate
                    loss.backward()
                    if (batch_idx + 1) % self.accumulate == 0:
                        self.allreduce_grads(model)
                        opt.step()
                        opt.zero_grad()
                else:
                    model.training_step(batch, batch_idx) if takes_idx else model.training_step(batch)
                model.on_train_batch_end(None, batch, batch_idx)
                if self.rank == 0 and self.log_every and model.global_step % self.log_every == 0:
                    msg = {k: (float(v) if torch.is_tensor(v) else v) for k, v in model.logged.items()}
                    print(f"[step {model.global_step}] {msg} ({time.time() - t0:.1f}s)", flush=True)
                if 0 < self.max_steps <= model.global_step:
                    done = True
                    break
            epoch += 1
            if 0 < self.max_epochs <= epoch:
                done = True
        if self.root and self.rank == 0:
            os.makedirs(self.root, exist_ok=True)
            save_checkpoint(model, list(model._optimizers), os.path.join(self.root, "last.ckpt"), epoch=epoch)
        return model
