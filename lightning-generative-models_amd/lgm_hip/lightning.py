"""LightningModule / Trainer surface used by the hot path.

The models derive from ``MiniLightningModule`` UNCONDITIONALLY and ``train.py`` always drives them with
``MiniTrainer`` — also when ``pytorch_lightning`` is importable (the reference's own environment): the
HIP engine needs ``prepare_hip``, flat gradient buffers and the graph-replayed step, none of which
``pl.Trainer`` knows about, and a ``pl.LightningModule`` routes ``global_step`` / ``optimizers()`` /
``log`` through a ``pl.Trainer`` that is not there.  ``HAVE_PL`` is informational only.

The stand-in provides the subset of behaviour the reference relies on
(train.py:113-141, ddpm.py:983,1017-1027,1047, wgan.py:58-82, vqvae.py:184-194):

  save_hyperparameters / hparams, log / log_dict (kept in ``logged``), global_step (counts
  optimizer.step() calls, which is what makes WGAN's n_critic schedule work — wgan.py:64),
  optimizers(), manual_backward(), automatic vs manual optimisation, on_train_batch_end,
  validation every ``check_val_every_n_epoch`` epochs, ``last.ckpt`` (ModelCheckpoint(save_last=True),
  train.py:113-117) written atomically at every epoch end / every ``ckpt_every_n_steps`` optimizer
  steps / on interruption, one process per GPU with gradient averaging over torch.distributed (RCCL).
"""
from __future__ import annotations

import importlib.util
import inspect
import os
import time
from typing import Any, Dict, Iterable, List, Optional

import torch
import torch.distributed as dist
from torch import nn

HAVE_PL = importlib.util.find_spec("pytorch_lightning") is not None     # informational (see above)


def multi_rank() -> bool:
    """What the reference passes as ``sync_dist`` (``torch.cuda.device_count() > 1``, i.e. "this is a DDP run"),
    stated for this engine: more than one rank in the process group."""
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


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

    def count_step(self):
        """An optimizer step that was issued slice by slice (FusedAdam.begin_step / step_slice) counts as one."""
        self._owner._global_step += 1

    def __getattr__(self, n):
        return getattr(self._opt, n)


class MiniLightningModule(nn.Module):
    def __init__(self):
        super().__init__()
        self._hparams = _AttrDict()
        self._global_step = 0
        self.automatic_optimization = True
        self.logged: Dict[str, Any] = {}
        self._sync_dist_names = set()
        self._grads_prescaled = False       # informational only: the exchange decides from the optimizers' grad_scale
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

    def log(self, name, value, sync_dist=False, **kw):
        """``sync_dist=True`` (reference: ``sync_dist=torch.cuda.device_count() > 1``, ddpm.py:1017-1023, wgan.py:77-82):
        the value REPORTED for this name is the mean over ranks.  Lazy and batched: nothing is exchanged here; the
        trainer calls ``synced_logs()`` once per logging interval, which averages all such scalars with ONE small
        all-reduce (Lightning issues one per logged scalar per step)."""
        self.logged[name] = value
        if sync_dist:
            self._sync_dist_names.add(name)

    def log_dict(self, d, sync_dist=False, **kw):
        self.logged.update(d)
        if sync_dist:
            self._sync_dist_names.update(d.keys())

    def synced_logs(self) -> Dict[str, float]:
        """Host floats of everything logged so far; names logged with ``sync_dist=True`` are averaged over the
        ranks that logged them.  Collective when world > 1: every rank must call it at the same step (the trainer
        does, on its logging interval).  Shape-safe: the exchanged tensor has a fixed (sum, count) pair per REGISTERED
        name, whether or not this rank logged it this interval, and the ranks first agree on the name list itself
        (a hash, one tiny all-reduce): a scalar logged with ``sync_dist=True`` under a rank-dependent condition raises
        the same error on every rank instead of hanging the job in mismatched collectives."""
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        vals = {k: v for k, v in self.logged.items() if torch.is_tensor(v) and v.numel() == 1 or isinstance(v, (int, float))}
        names = sorted(self._sync_dist_names)
        out = {k: float(v) for k, v in vals.items() if k not in self._sync_dist_names}
        if not names and world == 1:
            return out
        dev = self.device
        CAP = self._SYNC_SLOTS
        # ONE fixed-size float64 all-reduce: [h1, h1^2, h2, h2^2 | (sum, count) x CAP slots].  h1, h2 = 24-bit hashes of
        # the name list: every rank holds the same list  <=>  world * sum(h^2) == (sum h)^2 for both (exact in float64).
        import hashlib
        dg = hashlib.sha256("\n".join(names).encode()).digest()
        h1, h2 = float(int.from_bytes(dg[:3], "big")), float(int.from_bytes(dg[3:6], "big"))

        def pack(sub):
            row = []
            for n in sub:
                v = vals.get(n)
                row += [float(v), 1.0] if v is not None else [0.0, 0.0]
            return row
        head = [h1, h1 * h1, h2, h2 * h2] + pack(names[:CAP]) + [0.0, 0.0] * max(0, CAP - len(names))
        t = torch.tensor(head, dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(t)
        r = t.tolist()
        if world * r[1] != r[0] * r[0] or world * r[3] != r[2] * r[2]:
            raise RuntimeError("log(..., sync_dist=True) was called with different names on different ranks "
                               f"(this rank: {names}); log such scalars unconditionally on every rank")
        rest = r[4:]
        if len(names) > CAP:                         # the lists are known to be equal now: a second, sized exchange
            t2 = torch.tensor(pack(names[CAP:]), dtype=torch.float64, device=dev)
            if world > 1:
                dist.all_reduce(t2)
            rest = rest + t2.tolist()
        for i, n in enumerate(names):
            sm, cnt = rest[2 * i], rest[2 * i + 1]
            if cnt > 0:
                out[n] = sm / cnt
        return out

    _SYNC_SLOTS = 32

    def optimizers(self):
        if len(self._optimizers) == 1:
            return self._optimizers[0]
        return self._optimizers

    def manual_backward(self, loss, *a, **k):
        """Lightning semantics under DDP: gradients are averaged over ranks as part of the backward.  Only the
        flat buffers this backward actually wrote (a GAN's critic OR generator) are exchanged."""
        flats = _flat_grads_of(self)
        for fp in flats:
            fp.written = False               # set by FlatParams.begin_backward() of the passes that run now
        loss.backward(*a, **k)
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        if world > 1:
            for fp in flats:
                if fp.written:
                    dist.all_reduce(fp.grad)
                    if not _prescaled(self, world):  # else the fused optimizers multiply by 1/world (grad_scale)
                        fp.grad.div_(world)
                    fp.written = False
            for p in self.parameters():
                if getattr(p, "_lgm_flat", None) is None and p.grad is not None:
                    dist.all_reduce(p.grad)
                    p.grad.div_(world)

    # hooks (no-ops by default)
    def on_train_batch_end(self, outputs, batch, batch_idx):
        pass

    def configure_optimizers(self):
        raise NotImplementedError


LightningModule = MiniLightningModule


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
    for all buckets.  The 1/world average is folded into the optimiser (``grad_scale``).

    ``LGM_DDP_OVERLAP=0`` (A/B switch for the multi-GPU run): ``ready`` only notes the slice and ``finish`` exchanges what
    was noted - the whole buffer in ONE all-reduce when everything was - after the backward pass.  A collective kernel
    that sits on even one CU beside the backward makes every chip-filling persistent launch wait for a second round
    (+27 ... 35 % on those launches, tools/cu_hog_step.py), so overlap hides the exchange at a price; which
    side wins depends on the link time and can only be measured on more than one GPU."""

    @staticmethod
    def wanted(world: int) -> bool:
        """An exchange object is needed: more than one rank, or ``LGM_DDP_FORCE=1`` with a process group up - the ONE-rank
        rehearsal (tests/test_hip_rccl.py: RCCL initialised, collectives issued between the four graphs, captures beside
        the watchdog thread) that makes sure the first multi-GPU run is not the first time this code meets RCCL."""
        forced = os.environ.get("LGM_DDP_FORCE", "0") == "1" and dist.is_available() and dist.is_initialized()
        return world > 1 or forced

    def __init__(self, flat, group=None, overlap: Optional[bool] = None, beside_backward: bool = True):
        """``beside_backward=False``: the owner issues ``ready`` only after its whole backward pass (WGANFastStep), so the
        collective never shares the chip with a launch plan, whatever ``overlap`` says."""
        self.flat, self.group = flat, group
        self.beside_backward = bool(beside_backward)
        up = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if up else 1
        self.backend = str(dist.get_backend(group)) if up else None
        self.active = up and FlatGradSync.wanted(self.world)         # False: ready() / finish() issue nothing
        self.overlap = (os.environ.get("LGM_DDP_OVERLAP", "1") != "0") if overlap is None else bool(overlap)
        self.selection = self._select_kernels()
        self.handles = []
        self.pending = []
        self.covered = 0
        # self-description for the bench line: the collectives of the last finished step in element
        # counts, and - while ``measure`` is on - per step the time the launch stream (HIP events around the waits) and
        # the host (gloo's waits block it) spent waiting for the exchange = the communication that was NOT hidden
        self.measure = False
        self.last_buckets = []
        self._buckets = []
        self._spans = []

    def _select_kernels(self) -> dict:
        """The kernel selection of a rank follows from what is TRUE of its exchange, not from WORLD_SIZE: light
        F(4x4) workgroups and launch plans that leave 16 CUs free pay only while a collective's workgroups are resident
        beside the backward pass - the overlapped exchange on RCCL (``nccl``).  With ``LGM_DDP_OVERLAP=0`` (one all-reduce
        after the backward), under gloo (host-side reduction) or on one rank nothing shares the chip with a launch and the
        one-GPU rules are the faster ones (the rank defaults cost +4.9 % on config 5 alone).  An
        environment setting of a knob wins over this rule.  Every change clears the host's plan caches
        (lgm_hip/_lib.py: on_selection_change).  Returns the record the bench line prints."""
        resident = bool(self.active and self.overlap and self.beside_backward and self.backend == "nccl"
                        and self.flat.grad.is_cuda)
        rec = {"rule": ("overlapped exchange on RCCL: a collective is resident beside the backward" if resident else
                        "no collective resident beside the backward (" +
                        ("one rank" if not self.active else "LGM_DDP_OVERLAP=0" if not self.overlap
                         else "exchange after the backward" if not self.beside_backward
                         else f"backend {self.backend}") + "): one-GPU rules")}
        if not self.flat.grad.is_cuda:
            return rec
        from . import ops
        env = os.environ
        if resident:                                 # (not resident: nothing is touched - the library's defaults ARE the one-GPU rules)
            if "LGM_CU_MARGIN" not in env:
                ops.set_kernel_selection(cu_margin=16)
            if "LGM_WINO4_LIGHT" not in env and "LGM_WINO4_LIGHT_BELOW" not in env:
                ops.set_kernel_selection(light=1)
        rec["cu_margin"] = int(ops.lib().lgm_cu_margin())
        rec["light_f4x4_workgroups"] = ("all launches" if resident and "LGM_WINO4_LIGHT" not in env
                                        and "LGM_WINO4_LIGHT_BELOW" not in env else "environment / launch by launch")
        return rec

    def exposed_ms(self):
        """-> (mean stream-side wait per step in ms, mean host-side wait per step in ms, steps) over the steps finished
        while ``measure`` was on; synchronises the device.  Resets the record."""
        if not self._spans:
            return None, None, 0
        torch.cuda.synchronize()
        ev = [a.elapsed_time(b) for a, b, _ in self._spans if a is not None]
        host = [h * 1e3 for _, _, h in self._spans]
        n = len(self._spans)
        self._spans = []
        return (sum(ev) / len(ev) if ev else None), sum(host) / n, n

    def ready(self, lo: int, hi: int):
        """-> the asynchronous work handle (None on one rank, and when the exchange is deferred to ``finish``);
        ``finish()`` waits for whatever is still pending"""
        if not self.active or hi <= lo:
            return None
        self.covered += hi - lo
        if not self.overlap:
            self.pending.append((lo, hi))
            return None
        h = dist.all_reduce(self.flat.grad[lo:hi], group=self.group, async_op=True)
        self.handles.append(h)
        self._buckets.append(hi - lo)
        return h

    def finish(self):
        if self.pending:
            if self.covered == self.flat.total:
                spans = [(0, self.flat.total)]
            else:                                    # merge adjacent slices: as few collectives as the coverage allows
                spans = []
                for lo, hi in sorted(self.pending):
                    if spans and spans[-1][1] == lo:
                        spans[-1] = (spans[-1][0], hi)
                    else:
                        spans.append((lo, hi))
            self.pending = []
            for lo, hi in spans:
                self.handles.append(dist.all_reduce(self.flat.grad[lo:hi], group=self.group, async_op=True))
                self._buckets.append(hi - lo)
        if self.measure and self.handles:
            on_gpu = self.flat.grad.is_cuda
            e0 = torch.cuda.Event(enable_timing=True) if on_gpu else None
            e1 = torch.cuda.Event(enable_timing=True) if on_gpu else None
            if on_gpu:
                e0.record()
            t0 = time.perf_counter()
            for h in self.handles:
                h.wait()
            dt = time.perf_counter() - t0
            if on_gpu:
                e1.record()
            self._spans.append((e0, e1, dt))
        else:
            for h in self.handles:
                h.wait()
        self.handles = []
        if self._buckets:
            self.last_buckets, self._buckets = self._buckets, []
        if self.active:
            assert self.covered == self.flat.total, \
                f"gradient buckets covered {self.covered} of {self.flat.total} elements"
        self.covered = 0

    @property
    def grad_scale(self) -> float:
        return 1.0 / self.world


def _prescaled(module, world: int) -> bool:
    """True when the 1/world average of a SUM all-reduce is applied by the optimizers that will consume the gradients
    (``grad_scale`` folded into the fused kernels).  Decided from the optimizers the module holds NOW, not from a
    sticky module flag: fresh optimizers (grad_scale = 1) on a module that once trained through a prescaled path must
    get averaged gradients, not summed ones."""
    opts = getattr(module, "_optimizers", None) or []
    if not opts:
        return False
    want = 1.0 / world
    for o in opts:
        gs = getattr(o, "grad_scale", None)          # _CountingOptimizer forwards to the wrapped optimizer
        if gs is None or abs(gs - want) > 1e-12 * want:
            return False
    return True


class BufferSync:
    """DDP's ``broadcast_buffers=True`` for this engine (reference: DDPStrategy, utils/lightning_utils.py:37-43 —
    torch DDP re-broadcasts rank 0's module buffers at the start of every training forward, so BatchNorm running
    statistics and the VQ-EMA buffers never drift between ranks).

    The floating-point buffers the module's training writes (``module.ddp_buffers()`` when it offers one, else every
    floating-point buffer) are re-pointed at views of ONE flat tensor, so the broadcast is one small collective per
    step instead of one per buffer.  Buffers keep their names, shapes and values (state_dicts are unchanged)."""

    def __init__(self, module: nn.Module, group=None):
        self.module, self.group = module, group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.flat: Optional[torch.Tensor] = None
        self._owners = []
        self.pack()

    def _targets(self):
        if hasattr(self.module, "ddp_buffers"):
            wanted = {id(b) for b in self.module.ddp_buffers()}
        else:
            wanted = None
        out = []
        for mod in self.module.modules():
            for name, b in mod._buffers.items():
                if b is None or not b.dtype.is_floating_point:
                    continue
                if wanted is None or id(b) in wanted:
                    out.append((mod, name, b))
        return out

    def pack(self):
        tg = self._targets()
        self._owners = []
        if not tg:
            self.flat = None
            self.flats = []
            return
        # one flat tensor per dtype (a buffer keeps its dtype); a buffer registered in two modules stays ONE tensor
        by_dtype: Dict[torch.dtype, list] = {}
        seen = {}
        for mod, name, b in tg:
            if id(b) not in seen:
                seen[id(b)] = None
                by_dtype.setdefault(b.dtype, []).append(b)
        self.flats = []
        for dt, bufs in by_dtype.items():
            dev = bufs[0].device
            n = sum((b.numel() + 3) // 4 * 4 for b in bufs)
            flat = torch.zeros(n, dtype=dt, device=dev)
            off = 0
            for b in bufs:
                v = flat[off:off + b.numel()].view(b.shape)
                v.copy_(b.detach())
                seen[id(b)] = v
                off += (b.numel() + 3) // 4 * 4
            self.flats.append(flat)
        for mod, name, b in tg:
            v = seen[id(b)]
            mod._buffers[name] = v
            self._owners.append((mod, name, v.data_ptr()))
        self.flat = self.flats[0]

    def still_packed(self) -> bool:
        return all(mod._buffers[name] is not None and mod._buffers[name].data_ptr() == p for mod, name, p in self._owners)

    def broadcast(self):
        """Rank 0's buffer block to every rank (no-op on one rank)."""
        if self.world == 1 or self.flat is None:
            return
        if not self.still_packed():                  # module.to() / load with assign=True replaced the tensors
            self.pack()
        src = dist.get_global_rank(self.group, 0) if self.group is not None else 0
        for flat in self.flats:                      # one collective per dtype (fp32 only in every in-repo module)
            dist.broadcast(flat, src=src, group=self.group)


def is_device_error(e: BaseException) -> bool:
    """A launch / runtime / collective failure after which the GPU must not be used again (no state_dict(), no more
    launches).  By class where the class says it: ``LgmDeviceError`` (a C-ABI call returned a hipError_t; host-side
    argument rejections are ``LgmArgumentError`` and are NOT device errors), ``torch.AcceleratorError``,
    ``DistBackendError``; by message only for what torch raises as a plain RuntimeError."""
    from ._lib import LgmArgumentError, LgmDeviceError
    if isinstance(e, LgmArgumentError):
        return False
    if isinstance(e, LgmDeviceError):
        return True
    acc = getattr(torch, "AcceleratorError", None)
    if acc is not None and isinstance(e, acc):
        return True
    dbe = getattr(dist, "DistBackendError", None)
    if dbe is not None and isinstance(e, dbe):
        return True
    if isinstance(e, RuntimeError):
        msg = str(e)
        return any(s in msg for s in ("HIP error", "hipError", "CUDA error", "NCCL", "RCCL", "device-side assert",
                                      "an illegal memory access", "hipErrorLaunchFailure"))
    return False


def save_checkpoint(model, optimizers, path: str, epoch: int = 0):
    """Write a checkpoint with the layout of a PyTorch-Lightning ``.ckpt`` (reference train.py:41,
    113-117,140 resumes from / writes these): ``state_dict`` with the reference's keys (incl.
    ``ema.online_model.*`` / ``ema.ema_model.*`` / ``ema.initted`` / ``ema.step``), ``global_step``
    (= optimizer steps), ``optimizer_states`` in torch's own per-parameter format, hyper-parameters.
    Atomic: written to a temporary file in the same directory, then renamed over ``path``."""
    tmp = f"{path}.tmp.{os.getpid()}"
    try:
        torch.save({"epoch": int(epoch), "global_step": int(model.global_step),
                    "pytorch-lightning_version": "2.0.0+lgm_hip", "state_dict": model.state_dict(),
                    "loops": {}, "callbacks": {}, "optimizer_states": [o.state_dict() for o in optimizers],
                    "lr_schedulers": [], "hparams_name": "kwargs", "hyper_parameters": dict(model.hparams)}, tmp)
        os.replace(tmp, path)
    finally:
        if os.path.exists(tmp):                      # a failed write leaves nothing behind
            os.remove(tmp)


class MiniTrainer:
    """Single-node trainer: one process per GPU, DDP-style gradient averaging (RCCL over xGMI when the
    backend is nccl).  Models that offer ``make_fast_step`` (DDPM) are driven through it: bucketed
    all-reduce overlapped with the hand-written backward + HIP-graph replay of the step — the same objects
    bench.py times — with an in-process fallback to eager launches when capture is not possible."""

    def __init__(self, max_steps=-1, max_epochs=-1, accumulate_grad_batches=1, device=None,
                 default_root_dir=None, log_every=50, check_val_every_n_epoch=1, ckpt_every_n_steps=1000,
                 limit_val_batches=None, fast_path=True, **_ignored):
        self.max_steps, self.max_epochs = max_steps, max_epochs
        self.accumulate = max(1, int(accumulate_grad_batches))
        self.device = device
        self.root = default_root_dir
        self.log_every = log_every
        self.val_every = max(1, int(check_val_every_n_epoch or 1))
        self.ckpt_every = int(ckpt_every_n_steps or 0)
        self.limit_val_batches = limit_val_batches
        self.fast_path = fast_path
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank() if self.world > 1 else 0
        self.best_val = None
        self.best_path = None
        self.val_history: List[float] = []

    def allreduce_grads(self, module):
        if self.world == 1:
            return
        for fp in _flat_grads_of(module):
            dist.all_reduce(fp.grad)
            if not _prescaled(module, self.world):   # else 1/world is folded into the fused optimizers
                fp.grad.div_(self.world)
        # parameters that are not flat-bound (CPU plumbing models)
        for p in module.parameters():
            if getattr(p, "_lgm_flat", None) is None and p.grad is not None:
                dist.all_reduce(p.grad)
                p.grad.div_(self.world)

    # -- checkpoints (reference: ModelCheckpoint(dirpath, save_last=True, monitor="val_loss")) ------
    def _save_last(self, model, epoch):
        if self.root and self.rank == 0:
            os.makedirs(self.root, exist_ok=True)
            save_checkpoint(model, list(model._optimizers), os.path.join(self.root, "last.ckpt"), epoch=epoch)

    def _save_best(self, model, epoch, val_loss):
        if not (self.root and self.rank == 0) or val_loss is None:
            return
        if self.best_val is None or val_loss < self.best_val:
            path = os.path.join(self.root, f"epoch={epoch}-step={model.global_step}.ckpt")
            save_checkpoint(model, list(model._optimizers), path, epoch=epoch)
            if self.best_path and self.best_path != path and os.path.exists(self.best_path):
                os.remove(self.best_path)          # save_top_k = 1
            self.best_val, self.best_path = val_loss, path

    # -- validation (Lightning: eval mode, no grad, every check_val_every_n_epoch epochs) -----------
    def validate(self, model, loader, device):
        was_training = model.training
        model.eval()
        takes_idx = "batch_idx" in inspect.signature(model.validation_step).parameters
        seen = 0
        total = None
        with torch.no_grad():
            for batch_idx, batch in enumerate(loader):
                if self.limit_val_batches is not None and batch_idx >= self.limit_val_batches:
                    break
                batch = tuple(b.to(device, non_blocking=True) if torch.is_tensor(b) else b for b in batch)
                model.validation_step(batch, batch_idx) if takes_idx else model.validation_step(batch)
                v = model.logged.get("val_loss")
                if v is not None:
                    v = v.detach().float().reshape(()) if torch.is_tensor(v) else torch.tensor(float(v))
                    total = v.clone() if total is None else total + v
                seen += 1
        model.train(was_training)
        if total is None or seen == 0:
            return None
        val = total / seen
        if self.world > 1:                    # sync_dist-style mean over ranks
            val = val.to(device)
            dist.all_reduce(val)
            val = val / self.world
        val = float(val)
        self.val_history.append(val)
        return val

    def fit(self, model, datamodule=None, train_dataloader=None, ckpt_path=None, val_dataloader=None):
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
        if self.world > 1 and opts and all(hasattr(o, "grad_scale") for o in opts):
            # SUM all-reduce + 1/world folded into the fused optimizer kernels: no divide pass over the gradients
            for o in opts:
                o.grad_scale = 1.0 / self.world
            model._grads_prescaled = True
        # DDP side semantics (reference: DDPStrategy): rank 0's buffers re-broadcast before every training forward
        self.buffer_sync = BufferSync(model) if self.world > 1 else None
        loader = train_dataloader if train_dataloader is not None else datamodule.train_dataloader()
        if val_dataloader is None and datamodule is not None and hasattr(datamodule, "val_dataloader"):
            val_dataloader = datamodule.val_dataloader()
        model.train()
        fast = None
        if (self.fast_path and self.accumulate == 1 and device.type == "cuda" and hasattr(model, "make_fast_step")):
            # automatic optimisation: one optimizer; manual optimisation (GANs): the module's optimizer list
            fast = model.make_fast_step(model._optimizers[0] if model.automatic_optimization else model._optimizers,
                                        self.world)
        self.fast = fast
        epoch = int(ckpt.get("epoch", 0)) if ckpt is not None else 0
        done = False
        t0 = time.time()
        takes_idx = "batch_idx" in inspect.signature(model.training_step).parameters
        last_saved_step = model.global_step
        try:
            while not done:
                pending = 0                          # micro-batches whose gradients are not stepped yet
                stopped_mid_epoch = False
                for batch_idx, batch in enumerate(loader):
                    batch = tuple(b.to(device, non_blocking=True) if torch.is_tensor(b) else b for b in batch)
                    if self.buffer_sync is not None:
                        self.buffer_sync.broadcast()         # one flat broadcast of rank 0's buffer block
                    if fast is not None:
                        fast.step(batch, batch_idx)          # loss, backward, exchange, Adam, EMA hook
                    elif model.automatic_optimization:
                        opt = model._optimizers[0]
                        loss = model.training_step(batch, batch_idx) if takes_idx else model.training_step(batch)
                        if self.accumulate > 1:
                            loss = loss / self.accumulate
                        loss.backward()
                        pending += 1
                        if pending == self.accumulate:
                            self.allreduce_grads(model)
                            opt.step()
                            opt.zero_grad()
                            pending = 0
                        model.on_train_batch_end(None, batch, batch_idx)
                    else:
                        model.training_step(batch, batch_idx) if takes_idx else model.training_step(batch)
                        model.on_train_batch_end(None, batch, batch_idx)
                    if self.log_every and model.global_step % self.log_every == 0:
                        msg = model.synced_logs()            # collective (sync_dist scalars): every rank takes part
                        if self.rank == 0:
                            print(f"[step {model.global_step}] {msg} ({time.time() - t0:.1f}s)", flush=True)
                    if self.ckpt_every and model.global_step - last_saved_step >= self.ckpt_every:
                        self._save_last(model, epoch)
                        last_saved_step = model.global_step
                    if 0 < self.max_steps <= model.global_step:
                        done = stopped_mid_epoch = True
                        break
                if pending:                          # Lightning steps on the last batch of an epoch
                    self.allreduce_grads(model)
                    model._optimizers[0].step()
                    model._optimizers[0].zero_grad()
                if not stopped_mid_epoch:            # an epoch cut short by max_steps is not a completed epoch:
                    epoch += 1                       # a resume from its checkpoint runs that epoch again
                if 0 < self.max_epochs <= epoch:
                    done = True
                val = None
                # validation runs at the end of COMPLETED epochs only (a run cut short by max_steps stops here)
                if (val_dataloader is not None and hasattr(model, "validation_step") and not stopped_mid_epoch
                        and epoch % self.val_every == 0):
                    val = self.validate(model, val_dataloader, device)
                    if self.rank == 0 and val is not None:
                        print(f"[epoch {epoch}] val_loss {val:.6f}", flush=True)
                self._save_best(model, epoch, val)
                self._save_last(model, epoch)
                last_saved_step = model.global_step
        except BaseException as e:
            # Ctrl-C / SIGTERM-as-exception / a host-side error: keep the progress made so far.  NOT after a device
            # error: state_dict() copies from the GPU, and after a fault or a hang that can block for ever instead of
            # letting the process exit non-zero.
            device_error = is_device_error(e)
            if isinstance(e, (KeyboardInterrupt, SystemExit)) or not device_error:
                try:
                    self._save_last(model, epoch)
                finally:
                    raise
            raise
        return model
