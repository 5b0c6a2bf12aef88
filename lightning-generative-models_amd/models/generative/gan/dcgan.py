"""DCGAN generator / discriminator on the MI355X HIP engine — drop-in for the reference's
models/generative/gan/dcgan.py (same class names, constructor arguments, state_dict keys).

Generator  (reference :35-104): ConvTranspose2d 4x4 (+ BatchNorm2d train + ReLU) ..., final Tanh.
Discriminator (:107-164): Conv2d 4x4 s2 (+ BatchNorm2d train) + LeakyReLU(0.2) ..., final conv.
Both run as explicit forward / backward kernel sequences (implicit-GEMM MFMA convolutions,
lgm_bn_* BatchNorm primitives); the discriminator additionally exposes the hand-derived
second-order pass used by WGAN-GP (see wgan.py).
"""
from __future__ import annotations

from typing import List, Optional

import torch
from torch import nn

from lgm_hip import ops
from lgm_hip.bn import BatchNorm2d
from lgm_hip.flat import FlatParams, _r4
from lgm_hip.nn import Conv2d, ConvTranspose2d, GradCtx, param_kind
from models.generative.gan.gan import GAN

SLOPE = 0.2


def initialize_weights(model: nn.Module):
    """reference :23-32"""
    for m in model.modules():
        if isinstance(m, (Conv2d, ConvTranspose2d)):
            nn.init.normal_(m.weight.data, 0.0, 0.02)
        elif isinstance(m, BatchNorm2d):
            nn.init.normal_(m.weight.data, 1.0, 0.02)
            nn.init.constant_(m.bias.data, 0)
    return model


class _Net(nn.Module):
    """Common flat-storage plumbing of G and D."""

    def __init__(self):
        super().__init__()
        self._flat: Optional[FlatParams] = None

    def prepare_hip(self, device) -> FlatParams:
        device = torch.device(device)
        if self._flat is not None and self._flat.device == device and self._flat.still_bound():
            return self._flat
        self._flat = FlatParams([(n, p, param_kind(n, p)) for n, p in self.named_parameters()], device)
        # the 4x4 / stride-2 layers register with the non-fused Winograd engine (opt-in: LGM_WENG=1, lgm_hip/ops.py)
        self._weng_ptrs = []
        for m in self.modules():
            if isinstance(m, (Conv2d, ConvTranspose2d)) and m.k == 4 and m.stride == 2 and m.padding == 1:
                nw, cw = (m.cin, m.cout) if isinstance(m, ConvTranspose2d) else (m.cout, m.cin)   # Y side / X side channels
                wp = self._flat.ptr(m.weight)
                if ops.weng_register(wp, _r4(nw), _r4(cw), device):
                    self._weng_ptrs.append(wp)
        self._weng_hold = False
        return self._flat

    def weng_refresh(self, xy: bool = True, yx: bool = True):
        """The engine's transformed weights from the CURRENT weights (no-op unless layers registered).  Called at the start
        of every forward pass - the weights may have changed since the last one - unless a caller holds them fresh."""
        if getattr(self, "_weng_ptrs", None) and not self._weng_hold:
            ops.weng_refresh(self._weng_ptrs, xy, yx)

    def weng_fresh(self):
        """Context: refresh once, then skip the per-forward refreshes inside (several forward passes on unchanged weights:
        the critic's real / fake / interpolate passes of one loss)."""
        import contextlib

        @contextlib.contextmanager
        def hold():
            self.prepare_hip(self.device)
            self.weng_refresh()
            prev, self._weng_hold = self._weng_hold, True
            try:
                yield
            finally:
                self._weng_hold = prev
        return hold()

    def _anchor(self, device):
        self.prepare_hip(device)
        a = getattr(self, "_anchor_t", None)
        if a is None or a.device != torch.device(device):
            a = torch.zeros(1, device=device, requires_grad=True)
            self._anchor_t = a
        return a

    @property
    def device(self) -> torch.device:
        return next(self.parameters()).device


def _to_nhwc(x):
    B, C, H, W = x.shape
    t = ops.new((B, H, W, _r4(C)), x)
    ops.nchw_to_nhwc(x.detach().float().contiguous(), t)
    return t


class Generator(_Net):
    def __init__(self, img_size: int, img_channels: int, latent_dim: int) -> None:
        super().__init__()
        self.latent_dim, self.img_channels, self.img_size = latent_dim, img_channels, img_size
        if img_size == 64:
            spec = [(latent_dim, 1024, 4, 1, 0), (1024, 512, 4, 2, 1), (512, 256, 4, 2, 1), (256, 128, 4, 2, 1),
                    (128, img_channels, 4, 2, 1)]
        elif img_size == 28:
            spec = [(latent_dim, 256, 7, 1, 0), (256, 128, 4, 2, 1), (128, img_channels, 4, 2, 1)]
        else:
            raise ValueError("img_size must be 64 or 28")
        blocks = []
        for i, (ci, co, k, s, p) in enumerate(spec):
            final = i == len(spec) - 1
            blocks.append(nn.Sequential(ConvTranspose2d(ci, co, k, s, p, bias=False),
                                        BatchNorm2d(co) if not final else nn.Identity(), nn.Identity()))
        self.model = initialize_weights(nn.Sequential(*blocks))

    # ---- engine -----------------------------------------------------------------------------
    def fwd(self, z4, save: bool):
        self.weng_refresh(xy=False, yx=True)         # a ConvTranspose2d forward is the Y -> X direction
        tape = []
        h = z4
        n = len(self.model)
        for i, blk in enumerate(self.model):
            if i < n - 1:
                # train mode: the transposed convolution's epilogue leaves the batch statistics of its output
                # (lgm_conv_yx_stats), BatchNorm finishes them without reading the activation again
                a, st = blk[0].fwd(h, stats=True)
                hn, sv = blk[1].fwd(a, ops.ACT_RELU, 0.0, self.training, stats=st)
                tape.append((h, sv, hn))
            else:
                a = blk[0].fwd(h)
                hn = ops.new(a.shape, a)
                ops.act_fwd(a, None, None, hn, ops.ACT_TANH)
                tape.append((h, a, hn))
            h = hn
        return h, (tape if save else None)

    def bwd(self, tape, gout):
        self.weng_refresh(xy=True, yx=False)         # its input gradient the X -> Y direction (weights unchanged since fwd)
        gc = GradCtx(self._flat)
        n = len(self.model)
        g = gout
        sums = None      # BatchNorm backward sums of `g`, left by the epilogue of the convolution that produced it
        for i in range(n - 1, -1, -1):
            blk = self.model[i]
            h_in, sv, hn = tape[i]
            if i == n - 1:
                ga = ops.new(g.shape, g)
                ops.act_bwd(sv, None, g, ga, False, ops.ACT_TANH)        # sv = pre-activation here
            else:
                ga, _ = blk[1].apply_T(sv, g, gc, sums=sums)             # g arrives with relu'(hn) applied (below)
            # the ReLU in front of this layer's input (h_in = the previous block's output): its backward mask rides in
            # the input gradient's epilogue - and so do the reduction sums of the BatchNorm that produced h_in
            sums = self.model[i - 1][1].sums_request(tape[i - 1][1]) if i > 0 else None
            g = blk[0].bwd(gc, h_in, ga, need_gx=(i > 0), mask=(h_in if i > 0 else None), bn_sums=sums)
        self._flat.bind_grad_views()

    def forward(self, z: torch.Tensor) -> torch.Tensor:
        return _GenFn.apply(self._anchor(z.device), self, z)

    def random_sample(self, batch_size: int) -> torch.Tensor:
        z = torch.randn([batch_size, self.latent_dim, 1, 1], device=self.device)
        return self(z)


class _GenFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, net: Generator, z):
        save = bool(ctx.needs_input_grad[0])
        h, tape = net.fwd(_to_nhwc(z), save)
        B = z.shape[0]
        y = ops.new((B, net.img_channels, net.img_size, net.img_size), z)
        ops.nhwc_to_nchw(h, y)
        ctx.net, ctx.tape = net, tape
        return y

    @staticmethod
    def backward(ctx, gy):
        if ctx.tape is None:
            raise RuntimeError("Generator forward ran without saving activations")
        ctx.net.bwd(ctx.tape, _to_nhwc(gy))
        ctx.tape = None
        return None, None, None


class Discriminator(_Net):
    def __init__(self, img_size: int, img_channels: int) -> None:
        super().__init__()
        self.img_channels, self.img_size = img_channels, img_size
        if img_size == 64:   # (cin, cout, k, s, p, bn, final)
            spec = [(img_channels, 64, 4, 2, 1, False, False), (64, 128, 4, 2, 1, True, False),
                    (128, 256, 4, 2, 1, True, False), (256, 512, 4, 2, 1, True, False), (512, 1, 4, 1, 0, False, True)]
        elif img_size == 28:
            spec = [(img_channels, 64, 4, 2, 1, False, False), (64, 128, 4, 2, 1, True, False),
                    (128, 256, 7, 1, 0, True, False), (256, 1, 1, 1, 0, False, True)]
        else:
            raise ValueError("img_size must be 64 or 28")
        self.spec = spec
        blocks = []
        for (ci, co, k, s, p, bn, final) in spec:
            blocks.append(nn.Sequential(Conv2d(ci, co, k, s, p, bias=False), BatchNorm2d(co) if bn else nn.Identity(),
                                        nn.Identity()))
        self.model = initialize_weights(nn.Sequential(*blocks))

    # ---- engine: first-order ------------------------------------------------------------------
    def fwd(self, x4, save: bool = True):
        """x4: [B,H,W,r4(C)] -> scores [B,1,1,4] (column 0).  tape[k] = (h_in, a, bn_saved, h_out)."""
        self.weng_refresh()
        tape = []
        h = x4
        for blk, (ci, co, k, s, p, bn, final) in zip(self.model, self.spec):
            sv = None
            if bn:
                a, st = blk[0].fwd(h, stats=True)                # batch statistics from the convolution's epilogue
                hn, sv = blk[1].fwd(a, ops.ACT_LRELU, SLOPE, self.training, stats=st)
                tape.append((h, a, sv, hn))
                h = hn
                continue
            if not final:
                # Conv2d -> LeakyReLU without a BatchNorm between them (the critic's first layer): the activation rides
                # in the convolution's epilogue; the backward passes only ever need its OUTPUT (sign of hn)
                hn = blk[0].fwd(h, act=ops.ACT_LRELU, slope=SLOPE)
                a = hn
            else:
                a = blk[0].fwd(h)
                hn = a
            tape.append((h, a, sv, hn))
            h = hn
        return h, tape

    def bwd(self, gc: Optional[GradCtx], tape, gs, need_gx: bool):
        """Backward from score gradients gs [B,1,1,4].  gc None => input gradient only."""
        g = gs
        n = len(self.model)
        sums = None         # BatchNorm backward sums of `g` from the epilogue of the convolution that produced it
        for i in range(n - 1, -1, -1):
            blk = self.model[i]
            ci, co, k, s, p, bn, final = self.spec[i]
            h_in, a, sv, hn = tape[i]
            gn = g          # below the head g arrives with lrelu'(hn) applied: the mask rides in the producer's epilogue
            ga = blk[1].apply_T(sv, gn, gc, sums=sums)[0] if bn else gn
            last = i == 0
            mk = None if last else h_in                  # h_in = the previous block's LeakyReLU output
            sums = self.model[i - 1][1].sums_request(tape[i - 1][2]) if (i > 0 and self.spec[i - 1][5]) else None
            if gc is not None:
                g = blk[0].bwd(gc, h_in, ga, need_gx=(not last) or need_gx, mask=mk, mask_slope=SLOPE, bn_sums=sums)
            else:
                g = (blk[0].dgrad(ga, h_in.shape, mask=mk, mask_slope=SLOPE, bn_sums=sums)
                     if ((not last) or need_gx) else None)
        return g

    # ---- engine: gradient penalty with its second-order backward (wgan.py:117-156) ---------------
    def gradient_penalty(self, x4, lam: float, kind: str = "wgan"):
        """Forward on ``x4`` + first backward (dD/dx, grad_outputs = 1), then the penalty functional:
        kind "wgan": lam * mean_p (||g_p||_channels - 1)^2  (reference wgan.py:153-156, interpolates);
        kind "r1":   0.5 * mean_b sum_chw g^2               (reference r1gan.py:74-77, real data).
        Returns (penalty scalar tensor [1], state for gp_backward)."""
        L = ops.lib()
        scores, tape = self.fwd(x4, True)
        B = x4.shape[0]
        g = ops.new(scores.shape, x4)
        L.lgm_fill_col(g.data_ptr(), 4, B, 4, 0, 1.0, None, ops.stream())
        n = len(self.model)
        first = [None] * n
        sums = None
        for i in range(n - 1, -1, -1):
            blk = self.model[i]
            ci, co, k, s, p, bn, final = self.spec[i]
            h_in, a, sv, hn = tape[i]
            gn = g          # already multiplied by lrelu'(hn) in the epilogue of the layer above
            mvec = None
            if bn:
                ga, mvec = blk[1].apply_T(sv, gn, None, want_m=True, sums=sums)
            else:
                ga = gn
            first[i] = (gn, ga, mvec)
            sums = self.model[i - 1][1].sums_request(tape[i - 1][2]) if (i > 0 and self.spec[i - 1][5]) else None
            g = blk[0].dgrad(ga, h_in.shape, mask=(h_in if i > 0 else None), mask_slope=SLOPE, bn_sums=sums)
        Bx, H, W, Cp = x4.shape
        pen = ops.new((1,), x4)
        ws = ops.workspace(L.lgm_gp_penalty_workspace(Bx * H * W), x4.device)
        if kind == "r1":
            L.lgm_r1_penalty(g.data_ptr(), Bx * H * W, Bx, None, pen.data_ptr(), None, ws.data_ptr(), ops.stream())
        else:
            L.lgm_gp_penalty(g.data_ptr(), Bx * H * W, self.img_channels, lam, None, pen.data_ptr(), None,
                             ws.data_ptr(), ops.stream())
        return pen, (tape, first, g, lam, kind)

    def gp_backward(self, gc: GradCtx, state, gscale):
        """d(gscale * penalty)/d(theta_D): reverse sweep over the first-backward nodes, then over the
        forward nodes (LeakyReLU'' = 0, convolutions are linear, BatchNorm via adjoint_T)."""
        tape, first, gx, lam, kind = state
        L = ops.lib()
        n = len(self.model)
        B, H, W, Cp = gx.shape
        u = ops.new(gx.shape, gx)
        pen = ops.new((1,), gx)
        ws = ops.workspace(L.lgm_gp_penalty_workspace(B * H * W), gx.device)
        if kind == "r1":
            L.lgm_r1_penalty(gx.data_ptr(), B * H * W, B, gscale.data_ptr(), pen.data_ptr(), u.data_ptr(),
                             ws.data_ptr(), ops.stream())
        else:
            L.lgm_gp_penalty(gx.data_ptr(), B * H * W, self.img_channels, lam, gscale.data_ptr(), pen.data_ptr(),
                             u.data_ptr(), ws.data_ptr(), ops.stream())
        a_extra = [None] * n
        # ---- sweep 1: nodes of the first backward pass, in forward order --------------------------
        for i in range(n):
            blk = self.model[i]
            ci, co, k, s, p, bn, final = self.spec[i]
            h_in, a, sv, hn = tape[i]
            gn, ga, mvec = first[i]
            blk[0].wgrad(gc, ga, u)                      # node gx_i = W_i^T ga_i : adjoint w.r.t. W_i
            if final:
                break                                    # ga of the head is the constant grad_outputs = 1
            ga_bar = blk[0].linear(u)                    #                          adjoint w.r.t. ga_i
            if bn:
                gn_bar, a_extra[i] = blk[1].adjoint_T(sv, ga_bar, gn, mvec, gc)
            else:
                gn_bar = ga_bar
            if not final:
                ops.act_bwd(hn, None, gn_bar, gn_bar, False, ops.ACT_LRELU, SLOPE)   # mask is piecewise constant
            u = gn_bar
        # ---- sweep 2: forward nodes in reverse order --------------------------------------------
        a_bar = None
        for i in range(n - 1, -1, -1):
            blk = self.model[i]
            ci, co, k, s, p, bn, final = self.spec[i]
            h_in, a, sv, hn = tape[i]
            n_bar = None
            if i < n - 1 and a_bar is not None:
                nxt = self.model[i + 1]
                h_bar = nxt[0].dgrad(a_bar, hn.shape, mask=hn, mask_slope=SLOPE)    # a_{i+1} = W_{i+1} h_i, h_i = lrelu(n_i)
                nxt[0].wgrad(gc, a_bar, hn)
                n_bar = h_bar
            if bn:
                cur = a_extra[i]
                if n_bar is not None:
                    blk[1].apply_T(sv, n_bar, gc, out=cur, accumulate=True)
                a_bar = cur
            else:
                a_bar = n_bar
        if a_bar is not None:
            self.model[0][0].wgrad(gc, a_bar, tape[0][0])

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return _DiscFn.apply(self._anchor(x.device), self, x)


class _DiscFn(torch.autograd.Function):
    """scores = D(x).squeeze(); backward gives parameter gradients and (if needed) dL/dx."""

    @staticmethod
    def forward(ctx, anchor, net: Discriminator, x):
        save = bool(ctx.needs_input_grad[0]) or bool(ctx.needs_input_grad[2])
        scores, tape = net.fwd(_to_nhwc(x), True)
        ctx.net, ctx.tape, ctx.shape = net, (tape if save else None), x.shape
        ctx.params = bool(ctx.needs_input_grad[0])
        return scores.reshape(-1, 4)[:, 0].clone().squeeze()

    @staticmethod
    def backward(ctx, gs):
        net = ctx.net
        if ctx.tape is None:
            raise RuntimeError("Discriminator forward ran without saving activations")
        B = ctx.shape[0]
        g4 = torch.zeros((B, 1, 1, 4), device=gs.device)
        g4.view(B, 4)[:, 0] = gs.reshape(B)
        need_x = bool(ctx.needs_input_grad[2])
        gc = GradCtx(net._flat) if ctx.params else None
        gx = net.bwd(gc, ctx.tape, g4, need_x)
        if gc is not None:
            net._flat.bind_grad_views()
        ctx.tape = None
        out = None
        if need_x:
            out = ops.new(ctx.shape, gs)
            ops.nhwc_to_nchw(gx, out)
        return None, None, out


class DCGAN(GAN):
    """reference :167-245 (BCE losses on the critic logits; tiny [B]-sized host-side math)."""

    def __init__(self, img_channels: int, img_size: int, latent_dim: int, lr: float, b1: float, b2: float,
                 weight_decay: float, calculate_metrics: bool = False, metrics: List[str] = [], summary: bool = True):
        super().__init__(img_channels=img_channels, img_size=img_size, latent_dim=latent_dim, lr=lr, b1=b1, b2=b2,
                         weight_decay=weight_decay, calculate_metrics=calculate_metrics, metrics=metrics, summary=False)
        self.G = Generator(img_size=img_size, img_channels=img_channels, latent_dim=latent_dim)
        self.D = Discriminator(img_size=img_size, img_channels=img_channels)
        self.z = torch.randn([16, latent_dim, 1, 1])

    def _calculate_d_loss(self, x, x_hat):
        bce = torch.nn.functional.binary_cross_entropy_with_logits
        logits_real = self.D(x)
        d_loss_real = bce(logits_real, torch.ones_like(logits_real))
        logits_fake = self.D(x_hat.detach())
        d_loss_fake = bce(logits_fake, torch.zeros_like(logits_fake))
        d_loss = (d_loss_real + d_loss_fake) / 2
        return {"d_loss": d_loss, "d_loss_real": d_loss_real, "d_loss_fake": d_loss_fake,
                "logits_real": logits_real.mean(), "logits_fake": logits_fake.mean()}

    def _calculate_g_loss(self, x_hat):
        logits_fake = self.D(x_hat)
        g_loss = torch.nn.functional.binary_cross_entropy_with_logits(logits_fake, torch.ones_like(logits_fake))
        return {"g_loss": g_loss}
