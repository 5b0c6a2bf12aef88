"""HIP-graph replay of the DDPM training step.

At small per-GPU batches (strong scaling: 128 / N images per GPU) the ~640 kernel launches of a
step are host-bound when issued from Python.  The step is therefore captured ONCE into two HIP
graphs and replayed:

    graph 1 : t ~ randint, noise ~ randn, q_sample, UNet forward, loss, loss gradient,
              backward phase 1 (final -> up path -> middle)
    eager   : async all-reduce of gradient bucket [ups, mid, final]        (N > 1)
    graph 2 : backward phase 2 (down path, init conv, FiLM / time MLP)      -- overlaps the bucket above
    eager   : all-reduce of the two remaining buckets, wait, fused Adam (1 kernel), EMA (every 10th step)

Collectives are never captured (they are issued between the two replays), the optimiser's step
count lives on the host (one eager kernel), and the RNG is torch's graph-safe Philox generator.
The arithmetic is identical to the eager path (same kernels, same order).
"""
from __future__ import annotations

from typing import Optional

import torch

from . import ops


class GraphedDDPMStep:
    """``inject=True`` (parity tests): ``t`` / ``noise`` are static INPUT buffers the caller fills before each
    step instead of being drawn inside graph 1.  Either way ``self.t`` / ``self.noise`` hold the values the
    last replay used."""

    def __init__(self, model, opt, x: torch.Tensor, sync=None, warmup: int = 3, inject: bool = False):
        from models.generative.diffusion.ddpm import hip_loss_backward_phase1, hip_loss_forward
        self.model, self.opt, self.sync = model, opt, sync
        self.gd = model.ema.online_model
        self.net = self.gd.model
        self.x = x                                   # static input buffer (copy new batches into it)
        self.one = torch.ones(1, device=x.device)
        self._fwd, self._bwd1 = hip_loss_forward, hip_loss_backward_phase1
        self.net.grad_sync = None                    # collectives are issued by step(), never captured
        fp = self.net._flat
        self.t = torch.zeros(x.shape[0], dtype=torch.long, device=x.device) if inject else None
        self.noise = torch.zeros_like(x) if inject else None

        def part1():
            gd = self.gd
            if inject:
                t, noise = self.t, self.noise
            else:
                t = torch.randint(0, gd.num_timesteps, (x.shape[0],), device=x.device).long()
                noise = torch.randn_like(self.x)
                self.t, self.noise = t, noise
            loss, ctx = self._fwd(gd, self.x, t, noise, gd.auto_normalize, True)
            fp.zero_grad()
            return loss, self._bwd1(ctx, self.one)

        # warm-up and capture must not perturb the random stream: a run that captures at batch 0 and a run that
        # resumes from a checkpoint (and captures later) draw the same (t, noise) for the same seed
        rng_state = torch.cuda.get_rng_state(x.device)
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):                # eager warm-up (sizes workspaces, sets kernel attributes)
            for _ in range(warmup):
                _, st = part1()
                self.net.backward_phase2(st)
        cur.wait_stream(side)
        torch.cuda.synchronize()
        self.g1 = torch.cuda.CUDAGraph()
        # thread_local: only THIS thread is held to the capture rules -- the RCCL watchdog thread of a
        # multi-GPU job keeps polling its events while the step is being captured
        with torch.cuda.graph(self.g1, capture_error_mode="thread_local"):
            self.loss, st = part1()
        self.g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g2, pool=self.g1.pool(), capture_error_mode="thread_local"):
            self.net.backward_phase2(st)
        self._st = st                                # keeps the captured buffers alive
        torch.cuda.set_rng_state(rng_state, x.device)

    def step(self, batch_idx: int = 0):
        net, sync = self.net, self.sync
        self.g1.replay()
        if sync is not None:
            sync.ready(net._ups_start, net._flat.total)
        self.g2.replay()
        if sync is not None:
            sync.ready(net._head_end, net._ups_start)
            sync.ready(0, net._head_end)
            sync.finish()
        self.opt.step()
        self.opt.zero_grad()                         # host flag only: the next backward overwrites
        self.model.on_train_batch_end(None, None, batch_idx)
        return self.loss


class DDPMFastStep:
    """What ``MiniTrainer.fit`` drives for a ``DDPM`` module (``DDPM.make_fast_step``): the bucketed gradient
    exchange overlapped with the hand-written backward (``FlatGradSync``, N > 1) and the two-graph replay of
    the step, captured lazily at the first batch.  When capture is not possible (or a batch has another shape)
    the same step runs from eager launches, in the same process, with the same overlapped exchange.
    Logging (``train_loss``) and the reference's periodic in-training sampling (ddpm.py:1017-1027) stay
    outside the graphs."""

    def __init__(self, model, opt, world: int, use_graph: bool = True):
        from .lightning import FlatGradSync
        self.model, self.opt = model, opt
        self.net = model.ema.online_model.model
        self.sync = FlatGradSync(self.net._flat) if world > 1 else None
        inner = getattr(opt, "_opt", opt)            # MiniTrainer wraps optimizers in a step-counting proxy
        if self.sync is not None:
            inner.grad_scale = self.sync.grad_scale  # 1/N folded into Adam (no divide pass)
        self.use_graph = use_graph
        self.graphed: Optional[GraphedDDPMStep] = None
        self.mode = "eager"

    def _capture(self, x):
        try:
            self.graphed = GraphedDDPMStep(self.model, self.opt, x.clone(), self.sync)
            self.mode = "hipGraph replay (2 graphs/step)"
        except Exception as e:  # capture is an optimisation: fall back to eager launches
            import sys
            print(f"[lgm_hip] HIP-graph capture unavailable ({type(e).__name__}: {e}); eager launches",
                  file=sys.stderr, flush=True)
            self.use_graph = False
            self.graphed = None

    def step(self, batch, batch_idx: int = 0):
        from models.generative.diffusion.ddpm import _is_master
        m = self.model
        x = batch[0]
        if m.sample_every and m.global_step % m.sample_every == 0 and _is_master():
            m._log_sample()
        if self.use_graph and self.graphed is None:
            self._capture(x)
        if self.graphed is not None and x.shape == self.graphed.x.shape:
            self.graphed.x.copy_(x)
            loss = self.graphed.step(batch_idx)
        else:
            self.net.grad_sync = self.sync           # backward phases hand finished buckets to the exchange
            gd = m.ema.online_model
            loss = gd(x)
            loss.backward()
            if self.sync is not None:
                self.sync.finish()
            self.opt.step()
            self.opt.zero_grad()
            m.on_train_batch_end(None, batch, batch_idx)
            self.net.grad_sync = None
        m.log("train_loss", loss, prog_bar=True, logger=True, sync_dist=False)
        return loss


class ModuleFastStep:
    """Fast step for an automatic-optimisation module whose ``training_step`` is pure device work on one flat
    parameter buffer (VQ-VAE: ~145 launches of 5-15 us, host-bound when issued from Python):
    ``training_step`` + ``backward`` are captured ONCE into a HIP graph at the first batch and replayed; the
    gradient exchange (N > 1: one all-reduce of the flat buffer, 1/N folded into Adam), the fused Adam kernel and
    the module hooks stay eager.  Warm-up and capture must not advance the training state (EMA codebooks, running
    statistics, the random stream): parameters, buffers and the generator state are snapshotted and restored, so
    a run that captures is bit-identical to one that does not.  Falls back to eager launches in the same process
    when capture is not possible, a collective sits inside ``training_step`` (``collective_inside``) or a batch has
    another shape."""

    def __init__(self, model, opt, world: int = 1, use_graph: bool = True, collective_inside: bool = False):
        self.model, self.opt, self.world = model, opt, world
        self.flat = model._flat
        inner = getattr(opt, "_opt", opt)
        if world > 1:
            inner.grad_scale = 1.0 / world
        self.use_graph = use_graph and not (collective_inside and world > 1)
        self.graph = None
        self.static = None
        self.loss = None
        self._captured_logs = {}
        self.mode = "eager"

    # ---- one step from eager launches ---------------------------------------------------------------------------
    def _fwd_bwd(self, batch, batch_idx):
        loss = self.model.training_step(batch, batch_idx)
        self.flat.zero_grad()                        # host flag: this backward overwrites the gradient buffer
        loss.backward()
        return loss

    def _finish(self, batch, batch_idx):
        if self.world > 1:
            import torch.distributed as dist
            dist.all_reduce(self.flat.grad)
        self.opt.step()
        self.opt.zero_grad()
        self.model.on_train_batch_end(None, batch, batch_idx)

    def _capture(self, batch):
        m = self.model
        dev = self.flat.grad.device
        try:
            static = tuple(b.clone() if torch.is_tensor(b) else b for b in batch)
            keep = [t for t in list(m.parameters()) + list(m.buffers())]
            snap = [t.detach().clone() for t in keep]
            logged = dict(getattr(m, "logged", {}))
            rng_state = torch.cuda.get_rng_state(dev)
            cur = torch.cuda.current_stream()
            side = torch.cuda.Stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):            # eager warm-up: sizes workspaces, sets kernel attributes
                for _ in range(2):
                    self._fwd_bwd(static, 0)
            cur.wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            if hasattr(m, "logged"):
                m.logged.clear()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                loss = self._fwd_bwd(static, 0)
            # what training_step logged during capture lives in the graph's memory: every replay refreshes those
            # tensors in place, so they are what the module reports after each replayed step
            self._captured_logs = dict(getattr(m, "logged", {}))
            with torch.no_grad():                    # undo what the warm-up did to the training state
                for t, s in zip(keep, snap):
                    t.copy_(s)
            torch.cuda.set_rng_state(rng_state, dev)
            if hasattr(m, "logged"):
                m.logged.clear()
                m.logged.update(logged)
            self.graph, self.static, self.loss = g, static, loss
            self.mode = "hipGraph replay (1 graph/step)"
        except Exception as e:  # capture is an optimisation: fall back to eager launches
            import sys
            print(f"[lgm_hip] HIP-graph capture unavailable ({type(e).__name__}: {e}); eager launches",
                  file=sys.stderr, flush=True)
            self.use_graph = False
            self.graph = None

    def step(self, batch, batch_idx: int = 0):
        if self.use_graph and self.graph is None:
            self._capture(batch)
        g = self.graph
        same = g is not None and all((not torch.is_tensor(b)) or b.shape == s.shape for b, s in zip(batch, self.static))
        if same:
            for b, s in zip(batch, self.static):
                if torch.is_tensor(b):
                    s.copy_(b)
            g.replay()
            loss = self.loss
            if hasattr(self.model, "logged"):
                self.model.logged.update(self._captured_logs)
        else:
            loss = self._fwd_bwd(batch, batch_idx)
        self._finish(batch, batch_idx)
        return loss
