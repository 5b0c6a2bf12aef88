"""HIP-graph replay of the training steps.

At small per-GPU batches (strong scaling: 128 / N images per GPU) the ~400 kernel launches of a DDPM step are
host-bound when issued from Python.  The step is therefore captured ONCE and replayed:

    one rank : ONE graph - t ~ randint, noise ~ randn, q_sample, UNet forward, loss, the whole backward -
               then fused Adam (1 eager kernel) and EMA (every 10th step)
    N ranks  : FOUR graphs, cut where an exchange bucket of the flat gradient buffer becomes final
               (GraphedDDPMStep), with the asynchronous all-reduce of each bucket issued between the replays

Collectives are never captured, the optimiser's step count lives on the host, and the RNG is torch's graph-safe
Philox generator.  The arithmetic is identical to the eager path (same kernels, same order) - tested bit for bit.
ModuleFastStep (VQ-VAE: one graph) and WGANFastStep (critic graph / generator graph) follow the same rules: warm-up and
capture leave the training state untouched (_TrainingState), also when capture fails.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import ops
from .lightning import multi_rank


import os as _os

_ONE_GRAPH = _os.environ.get("LGM_ONE_GRAPH", "1") == "1"           # one rank: the whole step in ONE graph (A/B switch)
_STEP_PIPELINE = _os.environ.get("LGM_STEP_PIPELINE", "0") == "1"   # opt-in: weight passes of a bucket on a side stream


class _TrainingState:
    """Everything a warm-up / capture run of a training step may advance and a run that never tried to capture
    would not have: parameters and buffers (EMA codebooks, running statistics), the device random stream, the
    host-side BatchNorm batch counters and what the module logged.  ``restore()`` is called from a ``finally``: a
    capture that FAILS leaves the same state behind as one that succeeds, so the eager fallback is identical to a
    run that was eager from the start."""

    def __init__(self, modules, device):
        from .bn import BatchNorm2d
        self.device = device
        self.tensors = []
        seen = set()
        for m in modules:
            for t in list(m.parameters()) + list(m.buffers()):
                if id(t) not in seen:
                    seen.add(id(t))
                    self.tensors.append(t)
        self.snap = [t.detach().clone() for t in self.tensors]
        self.bns = [b for m in modules for b in m.modules() if isinstance(b, BatchNorm2d)]
        self.nbt = [b._nbt_pending for b in self.bns]
        self.logged = [(m, dict(m.logged)) for m in modules if hasattr(m, "logged")]
        self.rng = torch.cuda.get_rng_state(device)

    def restore(self):
        with torch.no_grad():
            for t, s in zip(self.tensors, self.snap):
                t.copy_(s)
        for b, n in zip(self.bns, self.nbt):
            b._nbt_pending = n
        for m, lg in self.logged:
            m.logged.clear()
            m.logged.update(lg)
        torch.cuda.set_rng_state(self.rng, self.device)


def _capture(fn, warmup: int, pool=None):
    """Eager warm-up of ``fn`` on a side stream (sizes workspaces, sets kernel attributes), then its capture into a
    HIP graph.  Returns (graph, fn's return value inside the capture, BatchNorm modules whose train-mode forward
    ran inside it — a replay does not pass through Python, so the step object advances their host-side batch
    counters itself).  thread_local: only THIS thread is held to the capture rules — the RCCL watchdog thread of a
    multi-GPU job keeps polling its events while the step is being captured."""
    from . import bn
    cur = torch.cuda.current_stream()
    side = torch.cuda.Stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        for _ in range(warmup):
            fn()
    cur.wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    bn.CAPTURE_TRACE = []
    try:
        kw = dict(capture_error_mode="thread_local")
        if pool is not None:
            kw["pool"] = pool
        with torch.cuda.graph(g, **kw):
            out = fn()
        trace = bn.CAPTURE_TRACE
    finally:
        bn.CAPTURE_TRACE = None
    return g, out, trace


class GraphedDDPMStep:
    """``inject=True`` (parity tests): ``t`` / ``noise`` are static INPUT buffers the caller fills before each
    step instead of being drawn inside graph 1.  Either way ``self.t`` / ``self.noise`` hold the values the
    last replay used.

    One rank: ONE graph for the whole forward + backward (``LGM_ONE_GRAPH=0``: two, forward + backward phase 1 | phase 2 -
    no measurable difference), one Adam launch.  With a gradient exchange
    (``sync``) the backward is cut at every bucket boundary - four graphs - so that each bucket's all-reduce is issued
    the moment its slice is final and runs beside everything that follows it:

        g1a  t, noise, q_sample, UNet forward, loss, backward of final block + up path   -> bucket 0 [ups], [final]
        g1b  backward of the middle blocks                                               -> bucket 1 [mid]
        g2a  backward of the down path + init conv                                       -> bucket 2 [init, downs]
        g2b  backward of the time MLP / FiLM projections                                 -> bucket 3 [FiLM, time]

    ``LGM_STEP_PIPELINE=1`` (opt-in, measured SLOWER on one GPU: 11.69 vs 11.44 ms at B = 128, 4.77 vs 4.65 ms at
    B = 16): four graphs on every rank count, and behind each of them on a SIDE stream the bucket's weight-sized passes -
    batched slab reduction, all-reduce, ITS slice of the Adam update (a bucket's weights are not read again by the
    backward once its gradients are final).  Bit-identical to the default (tested), but the streaming kernels' workgroups
    delay the one-workgroup-per-CU convolutions more than the overlap returns, and two more graph boundaries plus five
    Adam slices cost 0.2 ms by themselves.  Kept for multi-GPU experiments, where the Adam slices would run beside the
    later buckets' all-reduces."""

    def __init__(self, model, opt, x: torch.Tensor, sync=None, warmup: int = 3, inject: bool = False):
        from models.generative.diffusion.ddpm import hip_loss_backward_phase1a, hip_loss_forward
        self.model, self.opt, self.sync = model, opt, sync
        self.gd = model.ema.online_model
        self.net = self.gd.model
        self.x = x                                   # static input buffer (copy new batches into it)
        self.one = torch.ones(1, device=x.device)
        self.net.grad_sync = None                    # collectives are issued by step(), never captured
        fp = self.net._flat
        net = self.net
        self.t = torch.zeros(x.shape[0], dtype=torch.long, device=x.device) if inject else None
        self.noise = torch.zeros_like(x) if inject else None
        # (the pipelined variant applies a bucket's Adam slice right behind ITS all-reduce: only with the overlapped exchange)
        self.pipeline = _STEP_PIPELINE and (sync is None or getattr(sync, "overlap", True))
        split = sync is not None or self.pipeline

        def part1a():
            gd = self.gd
            if inject:
                t, noise = self.t, self.noise
            else:
                t = torch.randint(0, gd.num_timesteps, (x.shape[0],), device=x.device).long()
                noise = torch.randn_like(self.x)
                self.t, self.noise = t, noise
            loss, ctx = hip_loss_forward(gd, self.x, t, noise, gd.auto_normalize, True)
            fp.zero_grad()
            return loss, hip_loss_backward_phase1a(ctx, self.one)

        def whole():
            _, st = part1a()
            net.backward_phase2(net.backward_phase1b(st))

        # warm-up and capture must not perturb the random stream: a run that captures at batch 0 and a run that
        # resumes from a checkpoint (and captures later) draw the same (t, noise) for the same seed
        rng_state = torch.cuda.get_rng_state(x.device)
        try:
            # (the eager warm-up runs the WHOLE step: the pieces share workspaces and kernel attributes)
            cur = torch.cuda.current_stream()
            side = torch.cuda.Stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):            # eager warm-up (sizes workspaces, sets kernel attributes)
                for _ in range(warmup):
                    whole()
            cur.wait_stream(side)
            torch.cuda.synchronize()
            if self.pipeline:
                net._flush_collect = {}              # the phases hand their reduction rows over instead of launching
            try:
                if split:
                    g1a, (self.loss, st), _ = _capture(part1a, 0)
                    pool = g1a.pool()
                    g1b, st, _ = _capture(lambda: net.backward_phase1b(st), 0, pool)
                    g2a, st, _ = _capture(lambda: net.backward_phase2a(st), 0, pool)
                    g2b, _, _ = _capture(lambda: net.backward_phase2b(st), 0, pool)
                    self.graphs = [g1a, g1b, g2a, g2b]
                elif _ONE_GRAPH:
                    def everything():                # one rank: nothing has to happen between the phases
                        loss, st1 = part1a()
                        st2 = net.backward_phase1b(st1)
                        net.backward_phase2(st2)
                        return loss, st2
                    g1, (self.loss, st), _ = _capture(everything, 0)
                    self.graphs = [g1]
                else:
                    def part1():
                        loss, st1 = part1a()
                        return loss, net.backward_phase1b(st1)
                    g1, (self.loss, st), _ = _capture(part1, 0)
                    g2, _, _ = _capture(lambda: net.backward_phase2(st), 0, g1.pool())
                    self.graphs = [g1, g2]
                rows = net._flush_collect if self.pipeline else {}
            finally:
                net._flush_collect = None
            self.ranges = net.bucket_ranges()
            if self.pipeline:
                self.reducers = [ops.make_reducer(rows.get(k), x.device) for k in range(4)]
                self.side = torch.cuda.Stream()
                self.events = [torch.cuda.Event() for _ in range(4)]
            self._st = st                            # keeps the captured buffers alive
        finally:
            torch.cuda.set_rng_state(rng_state, x.device)

    def step(self, batch_idx: int = 0):
        if self.pipeline:
            return self._step_pipelined(batch_idx)
        sync = self.sync
        if sync is None:
            for g in self.graphs:
                g.replay()
        else:
            for k, g in enumerate(self.graphs):
                g.replay()
                for lo, hi in self.ranges[k]:
                    sync.ready(lo, hi)               # asynchronous, on RCCL's stream, behind the replay just enqueued
            sync.finish()
        self.opt.step()
        self.opt.zero_grad()                         # host flag only: the next backward overwrites
        self.model.on_train_batch_end(None, None, batch_idx)
        return self.loss

    def _step_pipelined(self, batch_idx: int):
        net, sync = self.net, self.sync
        fp = net._flat
        main, side = torch.cuda.current_stream(), self.side
        inner = getattr(self.opt, "_opt", self.opt)
        group, st = inner.begin_step(fp)
        side.wait_stream(main)                       # whatever touched the weights / gradients before this step
        for k, g in enumerate(self.graphs):
            g.replay()
            self.events[k].record(main)
            with torch.cuda.stream(side):
                side.wait_event(self.events[k])
                ops.launch_reducer(self.reducers[k])             # bucket k's gradients are final after this
                hs = [sync.ready(lo, hi) for lo, hi in self.ranges[k]] if sync is not None else []
                for h in hs:
                    if h is not None:
                        h.wait()                                 # the SIDE stream waits for the exchange, not the host
                for lo, hi in self.ranges[k]:
                    inner.step_slice(fp, group, st, lo, hi)
        main.wait_stream(side)
        if sync is not None:
            sync.finish()
        if hasattr(self.opt, "count_step"):
            self.opt.count_step()                    # MiniTrainer's proxy: one optimizer step
        self.opt.zero_grad()                         # host flag only: the next backward overwrites
        self.model.on_train_batch_end(None, None, batch_idx)
        return self.loss


class DDPMFastStep:
    """What ``MiniTrainer.fit`` drives for a ``DDPM`` module (``DDPM.make_fast_step``): the bucketed gradient
    exchange overlapped with the hand-written backward (``FlatGradSync``, N > 1) and the graph replay of
    the step, captured lazily at the first batch.  When capture is not possible (or a batch has another shape)
    the same step runs from eager launches, in the same process, with the same overlapped exchange.
    Logging (``train_loss``) and the reference's periodic in-training sampling (ddpm.py:1017-1027) stay
    outside the graphs."""

    def __init__(self, model, opt, world: int, use_graph: bool = True):
        from .lightning import FlatGradSync
        self.model, self.opt = model, opt
        self.net = model.ema.online_model.model
        self.sync = FlatGradSync(self.net._flat) if FlatGradSync.wanted(world) else None
        inner = getattr(opt, "_opt", opt)            # MiniTrainer wraps optimizers in a step-counting proxy
        if self.sync is not None:
            inner.grad_scale = self.sync.grad_scale  # 1/N folded into Adam (no divide pass)
        self.use_graph = use_graph
        self.graphed: Optional[GraphedDDPMStep] = None
        self.mode = "eager"

    def _capture(self, x):
        try:
            self.graphed = GraphedDDPMStep(self.model, self.opt, x.clone(), self.sync)
            self.mode = ("hipGraph replay (4 graphs/step, weight passes on a side stream)" if _STEP_PIPELINE else
                         f"hipGraph replay ({len(self.graphed.graphs)} graph{'s' if len(self.graphed.graphs) > 1 else ''}/step)")
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
        m.log("train_loss", loss, prog_bar=True, logger=True, sync_dist=multi_rank())
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
        self._bn_trace = []
        self._one = None
        self.mode = "eager"

    # ---- one step from eager launches ---------------------------------------------------------------------------
    def _fwd_bwd(self, batch, batch_idx):
        loss = self.model.training_step(batch, batch_idx)
        self.flat.zero_grad()                        # host flag: this backward overwrites the gradient buffer
        one = self._one                              # the backward's seed: made once (autograd would fill a new one per step)
        if one is None or one.device != loss.device or one.dtype != loss.dtype or one.shape != loss.shape:
            one = self._one = torch.ones_like(loss)
        loss.backward(one)
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
        state = None
        try:
            static = tuple(b.clone() if torch.is_tensor(b) else b for b in batch)
            state = _TrainingState([m], dev)
            if hasattr(m, "logged"):
                m.logged.clear()
            g, loss, trace = _capture(lambda: self._fwd_bwd(static, 0), 2)
            # what training_step logged during capture lives in the graph's memory: every replay refreshes those
            # tensors in place, so they are what the module reports after each replayed step
            self._captured_logs = dict(getattr(m, "logged", {}))
            self.graph, self.static, self.loss, self._bn_trace = g, static, loss, trace
            self.mode = "hipGraph replay (1 graph/step)"
        except Exception as e:  # capture is an optimisation: fall back to eager launches
            import sys
            print(f"[lgm_hip] HIP-graph capture unavailable ({type(e).__name__}: {e}); eager launches",
                  file=sys.stderr, flush=True)
            self.use_graph = False
            self.graph = None
        finally:
            if state is not None:                    # on BOTH paths: undo what warm-up / capture did to the training state
                state.restore()

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
            for b in self._bn_trace:                 # BatchNorm forwards inside the graph: host-side batch counters
                b._nbt_pending += 1
            if hasattr(self.model, "logged"):
                self.model.logged.update(self._captured_logs)
        else:
            loss = self._fwd_bwd(batch, batch_idx)
        self._finish(batch, batch_idx)
        return loss


class WGANFastStep:
    """Fast step for the WGAN / WGAN-GP module (manual optimisation, reference wgan.py:58-82): the critic update and
    the generator update are captured ONCE each into a HIP graph — z ~ randn, G forward, (alpha ~ rand, three critic
    forwards, the gradient-penalty double backward | critic forward + input-gradient sweep + G backward) — and
    replayed; the n_critic : 1 schedule (keyed on ``global_step``), the gradient exchange (N > 1: ONE all-reduce of
    the flat buffer the update wrote, 1/N folded into the fused optimizer), the optimizer kernel and logging stay on
    the host.  Warm-up and capture leave the training state (BatchNorm running statistics, random stream) untouched;
    eager fallback in the same process when capture is not possible or a batch has another shape."""

    def __init__(self, model, opts, world: int = 1, use_graph: bool = True):
        from .lightning import FlatGradSync
        self.model, self.world = model, world
        self.d_opt, self.g_opt = opts
        for o in opts:
            inner = getattr(o, "_opt", o)
            if world > 1:
                inner.grad_scale = 1.0 / world
        if world > 1:
            model._grads_prescaled = True
        self.sync = ({"d": FlatGradSync(model.D._flat, beside_backward=False),
                      "g": FlatGradSync(model.G._flat, beside_backward=False)} if world > 1 else None)
        self.use_graph = use_graph
        self.graphs = {}          # "d" / "g" -> (graph, static x, captured logs, BatchNorm trace)
        self.mode = "eager"

    # the two updates, without exchange / optimizer step (what a graph holds)
    def _critic(self, x):
        m = self.model
        x_hat = m.G.random_sample(x.size(0))
        ld = m._calculate_d_loss(x, x_hat)
        m.D._flat.zero_grad()
        ld["d_loss"].backward()
        return ld

    def _generator(self, x):
        m = self.model
        x_hat = m.G.random_sample(x.size(0))
        ld = m._calculate_g_loss(x_hat)
        m.G._flat.zero_grad()
        ld["g_loss"].backward()
        return ld

    def _capture(self, key, x):
        m = self.model
        state = None
        try:
            static = x.clone()
            fn = self._critic if key == "d" else self._generator
            state = _TrainingState([m], x.device)
            g, logs, trace = _capture(lambda: fn(static), 2)
            self.graphs[key] = (g, static, dict(logs), trace)
            self.mode = "hipGraph replay (critic graph / generator graph)"
        except Exception as e:  # capture is an optimisation: fall back to eager launches
            import sys
            print(f"[lgm_hip] HIP-graph capture unavailable ({type(e).__name__}: {e}); eager launches",
                  file=sys.stderr, flush=True)
            self.use_graph = False
            self.graphs.pop(key, None)
        finally:
            if state is not None:
                state.restore()

    def step(self, batch, batch_idx: int = 0):
        m = self.model
        x = batch[0]
        critic = (m.global_step + 1) % (m.hparams.n_critic + 1) != 0
        key = "d" if critic else "g"
        if self.use_graph and key not in self.graphs and m.training:
            self._capture(key, x)
        ent = self.graphs.get(key)
        if ent is not None and ent[1].shape == x.shape:
            g, static, logs, trace = ent
            static.copy_(x)
            g.replay()
            for b in trace:
                b._nbt_pending += 1
        else:
            logs = self._critic(x) if critic else self._generator(x)
        if self.sync is not None:
            sy = self.sync[key]
            sy.ready(0, sy.flat.total)
            sy.finish()
        opt = self.d_opt if critic else self.g_opt
        opt.step()                                   # the counting proxy advances global_step (the schedule's clock)
        opt.zero_grad()
        m.log_dict(logs, prog_bar=True, logger=True, sync_dist=self.world > 1)
        return logs
