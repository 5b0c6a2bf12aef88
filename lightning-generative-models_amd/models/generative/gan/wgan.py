"""WGAN / WGAN-GP on the MI355X HIP engine — drop-in for the reference's
models/generative/gan/wgan.py (constructor arguments, training_step schedule, loss names,
configure_optimizers).

The critic step (reference :84-110, :117-156) is ONE autograd Function: three critic forwards
(real, fake, interpolates), the input-gradient pass on the interpolates, the channel-norm
penalty, and a hand-derived backward that differentiates THROUGH that input-gradient pass
(convolutions are linear, LeakyReLU'' = 0, train-mode BatchNorm via its adjoint operators).
"""
from __future__ import annotations

from typing import List

import torch

from lgm_hip import ops
from lgm_hip.lightning import multi_rank
from lgm_hip.nn import GradCtx
from lgm_hip.optim import FusedAdam, FusedRMSprop
from models.generative.gan.dcgan import DCGAN, _to_nhwc


class WGAN(DCGAN):
    def __init__(self, img_channels: int = 3, img_size: int = 64, latent_dim: int = 100, lr: float = 0.00005,
                 weight_decay: float = 0, b1: float = 0.5, b2: float = 0.9, n_critic: int = 5,
                 clip_value: float = 0.01, grad_penalty: float = 10, constraint_method: str = "gp",
                 calculate_metrics: bool = False, metrics: List[str] = [], summary: bool = True) -> None:
        super().__init__(img_channels=img_channels, img_size=img_size, latent_dim=latent_dim, lr=lr, b1=b1, b2=b2,
                         weight_decay=weight_decay, calculate_metrics=calculate_metrics, metrics=metrics,
                         summary=summary)
        assert constraint_method in ["gp", "clip"], \
            "Either gradient penalty (gp) or weight clipping (clip) to enforce 1-Lipschitz constraint."
        self.clip_value = clip_value
        self.grad_penalty = grad_penalty
        self.constraint_method = constraint_method
        self.save_hyperparameters()

    def training_step(self, batch) -> None:
        """reference :58-82 — n_critic critic updates per generator update, keyed on global_step."""
        x, _ = batch
        x_hat = self.G.random_sample(x.size(0))
        d_optim, g_optim = self.optimizers()
        if (self.global_step + 1) % (self.hparams.n_critic + 1) != 0:
            loss_dict = self._calculate_d_loss(x, x_hat)
            d_optim.zero_grad(set_to_none=True)
            self.manual_backward(loss_dict["d_loss"])
            d_optim.step()
        else:
            loss_dict = self._calculate_g_loss(x_hat)
            g_optim.zero_grad(set_to_none=True)
            self.manual_backward(loss_dict["g_loss"])
            g_optim.step()
        self.log_dict(loss_dict, prog_bar=True, logger=True, sync_dist=multi_rank())

    def make_fast_step(self, opts, world: int = 1, use_graph: bool = True):
        """The step object ``MiniTrainer.fit`` drives instead of ``training_step``: the critic update and the generator
        update replayed from one HIP graph each, schedule / exchange / optimizer kernel on the host (what bench.py
        times for this workload)."""
        from lgm_hip.graph import WGANFastStep
        return WGANFastStep(self, opts, world, use_graph)

    def _calculate_d_loss(self, x, x_hat, alpha=None):
        """reference :84-110.  ``alpha`` may be injected (parity tests); default U[0,1) per sample."""
        with_gp = self.training and self.hparams.constraint_method == "gp"
        if alpha is None and with_gp:
            alpha = torch.rand(x.size(0), 1, 1, 1, device=x.device)
        d_loss, real, fake, gp = _CriticLossFn.apply(self.D._anchor(x.device), self.D, x, x_hat.detach(), alpha,
                                                     float(self.hparams.grad_penalty), with_gp)
        out = {"d_loss": d_loss, "d_loss_real": real, "d_loss_fake": fake}
        if with_gp:
            out["gradient_penalty"] = gp
        elif self.training:
            self._weight_clipping()          # reference :101-102 (after the loss, before its backward)
        return out

    def _weight_clipping(self):
        """reference :158-168: every critic parameter clamped to [-clip_value, clip_value] — one
        launch over the critic's flat parameter buffer (its padding lanes are zero and stay zero)."""
        self.D.prepare_hip(next(self.D.parameters()).device)
        ops.clamp_(self.D._flat.data, -float(self.hparams.clip_value), float(self.hparams.clip_value))
        self.D.weng_refresh()        # the backward pass that follows reads the CLIPPED weights (reference :101-102)

    def _calculate_g_loss(self, x_hat):
        """reference :112-115: g_loss = -D(x_hat).mean()"""
        return {"g_loss": _GenLossFn.apply(self.D, x_hat)}

    def configure_optimizers(self):
        """reference :170-197: RMSprop(lr) for the weight-clipping variant, Adam for gradient penalty"""
        if self.hparams.constraint_method == "clip":
            return [FusedRMSprop(self.D.parameters(), lr=self.hparams.lr),
                    FusedRMSprop(self.G.parameters(), lr=self.hparams.lr)], []
        kw = dict(lr=self.hparams.lr, betas=(self.hparams.b1, self.hparams.b2), weight_decay=self.hparams.weight_decay)
        return [FusedAdam(self.D.parameters(), **kw), FusedAdam(self.G.parameters(), **kw)], []


class _CriticLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, D, x, x_hat, alpha, lam, with_gp):
        ctx.set_materialize_grads(False)            # no zero tensors (fill launches) for the logged, non-differentiable outputs
        L = ops.lib()
        st = ops.stream()
        B = x.shape[0]
        x4, xh4 = _to_nhwc(x), _to_nhwc(x_hat)
        with D.weng_fresh():         # three forward passes on the same weights: the engine's weight transform runs once
            s_real, t_real = D.fwd(x4, True)
            s_fake, t_fake = D.fwd(xh4, True)
            vals = torch.zeros(4, device=x.device)
            L.lgm_mean_col(s_real.data_ptr(), 4, B, 1.0, vals[0:1].data_ptr(), st)
            L.lgm_mean_col(s_fake.data_ptr(), 4, B, 1.0, vals[1:2].data_ptr(), st)
            gp_state = None
            if with_gp:
                xi = ops.new(x4.shape, x4)
                a = alpha.detach().float().reshape(B).contiguous()
                L.lgm_lerp_rows(x4.data_ptr(), xh4.data_ptr(), a.data_ptr(), xi.data_ptr(), B, x4[0].numel(), st)
                pen, gp_state = D.gradient_penalty(xi, lam)
                vals[2:3].copy_(pen)
        L.lgm_wgan_dloss(vals.data_ptr(), st)
        ctx.stuff = (D, t_real, t_fake, gp_state, B)
        outs = (vals[3].clone(), vals[0].clone(), vals[1].clone(), vals[2].clone())
        ctx.mark_non_differentiable(*outs[1:])
        return outs

    @staticmethod
    def backward(ctx, gloss, *_):
        D, t_real, t_fake, gp_state, B = ctx.stuff
        L = ops.lib()
        st = ops.stream()
        gl = gloss.detach().float().reshape(1).contiguous()
        gc = GradCtx(D._flat)
        g4 = torch.empty((B, 1, 1, 4), device=gl.device)
        L.lgm_fill_col(g4.data_ptr(), 4, B, 4, 0, -1.0 / B, gl.data_ptr(), st)     # d(-mean real)
        D.bwd(gc, t_real, g4, False)
        g4b = torch.empty((B, 1, 1, 4), device=gl.device)
        L.lgm_fill_col(g4b.data_ptr(), 4, B, 4, 0, 1.0 / B, gl.data_ptr(), st)     # d(+mean fake)
        D.bwd(gc, t_fake, g4b, False)
        if gp_state is not None:
            D.gp_backward(gc, gp_state, gl)
        D._flat.bind_grad_views()
        ctx.stuff = None
        return None, None, None, None, None, None, None


class _GenLossFn(torch.autograd.Function):
    """g_loss = -mean(D(x_hat)); backward returns d g_loss / d x_hat only (the critic's own
    parameter gradients of this pass are discarded by the reference's next zero_grad anyway)."""

    @staticmethod
    def forward(ctx, D, x_hat):
        D.prepare_hip(x_hat.device)
        B = x_hat.shape[0]
        scores, tape = D.fwd(_to_nhwc(x_hat), True)
        out = torch.empty(1, device=x_hat.device)
        ops.lib().lgm_mean_col(scores.data_ptr(), 4, B, -1.0, out.data_ptr(), ops.stream())
        ctx.stuff = (D, tape, x_hat.shape)
        return out.view(())

    @staticmethod
    def backward(ctx, gloss):
        D, tape, shape = ctx.stuff
        B = shape[0]
        gl = gloss.detach().float().reshape(1).contiguous()
        g4 = torch.empty((B, 1, 1, 4), device=gl.device)
        ops.lib().lgm_fill_col(g4.data_ptr(), 4, B, 4, 0, -1.0 / B, gl.data_ptr(), ops.stream())
        gx = D.bwd(None, tape, g4, True)
        out = ops.new(shape, gl)
        ops.nhwc_to_nchw(gx, out)
        ctx.stuff = None
        return None, out
