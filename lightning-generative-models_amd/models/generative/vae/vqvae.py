"""VQ-VAE on the MI355X HIP engine — drop-in for the reference's models/generative/vae/vqvae.py
(+ models/modules/vector_quantizer.py, residual.py): same class names, constructor arguments,
state_dict keys and training_step / configure_optimizers surface.

Encoder / decoder convolutions run on the implicit-GEMM MFMA kernels (stride-2 4x4 conv and
conv-transpose included), the quantiser on lgm_vq_* (no [N,K] distance / one-hot matrices, int64
indices, deterministic segmented sums for the EMA statistics and the codebook gradient).
Forward and backward are explicit kernel sequences; autograd only sees one Function.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
from torch import nn

from lgm_hip import ops
from lgm_hip.flat import FlatParams, _r4
from lgm_hip.lightning import LightningModule, multi_rank
from lgm_hip.nn import Conv2d, ConvTranspose2d, GradCtx, param_kind
from lgm_hip.optim import FusedAdam
from models.modules.residual import ResidualBlock, ResidualStack  # noqa: F401  (reference layout: modules/residual.py)
from models.modules.vector_quantizer import VectorQuantizer, VectorQuantizerEMA  # noqa: F401


class Encoder(nn.Module):
    """reference vqvae.py:17-54"""

    def __init__(self, img_channels, embedding_dim, hidden_dim, num_residual_layers, num_residual_hiddens):
        super().__init__()
        h = hidden_dim
        self.layers = nn.Sequential(
            Conv2d(img_channels, h // 4, 4, 2, 1), nn.Identity(),
            Conv2d(h // 4, h // 2, 4, 2, 1), nn.Identity(),
            Conv2d(h // 2, h, 4, 2, 1), nn.Identity(),
            Conv2d(h, h, 3, 1, 1),
            ResidualStack(h, h, num_residual_layers, num_residual_hiddens),
            Conv2d(h, embedding_dim, 1))

    def fwd(self, x):
        L = self.layers
        acts = [x]
        cur = x
        for i in (0, 2, 4):
            cur = L[i].fwd(cur, act=ops.ACT_RELU)                       # Conv2d -> ReLU: one launch
            acts.append(cur)
        a3 = L[6].fwd(cur, act=ops.ACT_RELU)                            # the residual stack's first in-place ReLU
        s, st = L[7].fwd(a3, True)
        lat = L[8].fwd(s)
        return lat, (acts, st, s)

    def bwd(self, gc, saved, glat):
        acts, st, s = saved
        L = self.layers
        g = L[8].bwd(gc, s, glat, mask=s)                               # * relu'(stack output)
        g = L[7].bwd(gc, st, g)
        g = L[6].bwd(gc, acts[3], g, mask=acts[3])
        for k, i in ((3, 4), (2, 2), (1, 0)):
            g = L[i].bwd(gc, acts[k - 1], g, need_gx=(i != 0), mask=(acts[k - 1] if i != 0 else None))


class Decoder(nn.Module):
    """reference vqvae.py:57-88"""

    def __init__(self, img_channels, embedding_dim, hidden_dim, num_residual_layers, num_residual_hiddens):
        super().__init__()
        h = hidden_dim
        self.layers = nn.Sequential(
            Conv2d(embedding_dim, h, 3, 1, 1),
            ResidualStack(h, h, num_residual_layers, num_residual_hiddens),
            ConvTranspose2d(h, h // 2, 4, 2, 1), nn.Identity(),
            ConvTranspose2d(h // 2, h // 4, 4, 2, 1), nn.Identity(),
            ConvTranspose2d(h // 4, img_channels, 4, 2, 1), nn.Identity())

    def fwd(self, q):
        L = self.layers
        a0 = L[0].fwd(q, act=ops.ACT_RELU)                              # the residual stack's first in-place ReLU
        s, st = L[1].fwd(a0, True)
        u1 = L[2].fwd(s, act=ops.ACT_RELU)
        u2 = L[4].fwd(u1, act=ops.ACT_RELU)
        pre = L[6].fwd(u2)          # the Tanh behind it is applied by the caller together with the reconstruction loss
        return pre, (q, st, s, u1, u2)

    def bwd(self, gc, saved, gpre):
        """gpre: gradient w.r.t. the last layer's output BEFORE the Tanh"""
        q, st, s, u1, u2 = saved
        L = self.layers
        g = L[6].bwd(gc, u2, gpre, mask=u2)
        g = L[4].bwd(gc, u1, g, mask=u1)
        g = L[2].bwd(gc, s, g, mask=s)
        g = L[1].bwd(gc, st, g)
        return L[0].bwd(gc, q, g)


class VQVAE(LightningModule):
    def __init__(self, img_channels: int = 3, img_size: int = 64, embedding_dim: int = 64, num_embeddings: int = 512,
                 hidden_dim: int = 256, num_residual_layers: int = 2, num_residual_hiddens: int = 256,
                 commitment_cost: float = 0.25, use_ema: bool = True, decay: float = 0.99, epsilon: float = 1e-5,
                 lr: float = 1e-4, b1: float = 0.5, b2: float = 0.999, weight_decay: float = 1e-5,
                 loss_weights: Dict = {"recon_loss": 1.0, "vq_loss": 1.0}) -> None:
        super().__init__()
        self.save_hyperparameters()
        kw = dict(img_channels=img_channels, embedding_dim=embedding_dim, hidden_dim=hidden_dim,
                  num_residual_layers=num_residual_layers, num_residual_hiddens=num_residual_hiddens)
        self.encoder = Encoder(**kw)
        self.decoder = Decoder(**kw)
        if use_ema:
            self.vector_quantizer = VectorQuantizerEMA(num_embeddings, embedding_dim, commitment_cost, decay, epsilon)
        else:
            self.vector_quantizer = VectorQuantizer(num_embeddings, embedding_dim, commitment_cost)
        self._flat: Optional[FlatParams] = None
        self.last = {}

    # ---- flat storage ---------------------------------------------------------------------
    def prepare_hip(self, device) -> FlatParams:
        device = torch.device(device)
        if self._flat is not None and self._flat.device == device and self._flat.still_bound():
            return self._flat
        named = [(n, p, param_kind(n, p)) for n, p in self.named_parameters()]
        self._flat = FlatParams(named, device)
        return self._flat

    def _anchor(self, device):
        self.prepare_hip(device)
        a = getattr(self, "_anchor_t", None)
        if a is None or a.device != torch.device(device):
            a = torch.zeros(1, device=device, requires_grad=True)
            self._anchor_t = a
        return a

    # ---- engine ---------------------------------------------------------------------------
    def run(self, x: torch.Tensor, save: bool):
        """x NCHW.  Returns dict(loss terms, x_hat nhwc) and the tape."""
        B, C, H, W = x.shape
        Cp = _r4(C)
        x4 = ops.new((B, H, W, Cp), x)
        ops.nchw_to_nhwc(x.contiguous(), x4)
        lat, enc_saved = self.encoder.fwd(x4)
        q, out3, vq_saved = self.vector_quantizer.fwd(lat, self.training)
        pre, dec_saved = self.decoder.fwd(q)
        xh = ops.new(pre.shape, pre)
        per = ops.new((B,), x)      # per-sample reconstruction terms; their mean is taken by the loss kernel (_VQVAEStepFn)
        ops.lib().lgm_tanh_mse_fwd(pre.data_ptr(), x4.data_ptr(), Cp, B, C, H * W, Cp, xh.data_ptr(), per.data_ptr(),
                                   ops.stream())
        tape = (x4, enc_saved, vq_saved, dec_saved, xh) if save else None
        return dict(recon_samples=per, out3=out3, x_hat=xh, indices=vq_saved[2], latents=lat), tape

    def backward_hip(self, tape, gloss, w_recon: float, w_vq: float):
        """gloss: device scalar d L / d loss for loss = w_recon * recon + w_vq * vq"""
        x4, enc_saved, vq_saved, dec_saved, xh = tape
        B, H, W, Cp = x4.shape
        C = self.hparams.img_channels
        gc = GradCtx(self._flat, defer=True)     # weight-gradient slabs of all layers reduced by ONE launch (flush)
        gpre = ops.new(xh.shape, xh)
        g2 = ops.new((2,), xh)                   # (gloss w_recon, gloss w_vq)
        ops.lib().lgm_tanh_mse_bwd(xh.data_ptr(), x4.data_ptr(), Cp, gloss.data_ptr(), w_recon, w_vq, B, C, H * W, Cp,
                                   gpre.data_ptr(), g2.data_ptr(), ops.stream())
        gq = self.decoder.bwd(gc, dec_saved, gpre)
        glat = self.vector_quantizer.bwd(gc, vq_saved, gq, g2[1:2])
        self.encoder.bwd(gc, enc_saved, glat)
        gc.flush()
        self._flat.bind_grad_views()

    def forward(self, x: torch.Tensor):
        """(x_hat NCHW, vq_loss, perplexity) like the reference forward (inference / no-grad use)."""
        self.prepare_hip(x.device)
        r, _ = self.run(x.detach().float(), False)
        B, C, H, W = x.shape
        xh = ops.new((B, C, H, W), x)
        ops.nhwc_to_nchw(r["x_hat"], xh)
        return xh, r["out3"][0], r["out3"][1]

    def _common_step(self, batch, batch_idx: int, split: str):
        x, _ = batch
        w = self.hparams.loss_weights
        loss, recon, vq, ppl = _VQVAEStepFn.apply(self._anchor(x.device), self, x, float(w["recon_loss"]),
                                                  float(w["vq_loss"]))
        self.log_dict({f"{split}_loss": loss, f"{split}_recon_loss": recon, f"{split}_vq_loss": vq,
                       f"{split}_perplexity": ppl}, prog_bar=True, logger=True, sync_dist=multi_rank())
        return loss

    def training_step(self, batch, batch_idx):
        return self._common_step(batch, batch_idx, "train")

    def validation_step(self, batch, batch_idx):
        return self._common_step(batch, batch_idx, "val")

    def configure_optimizers(self):
        return FusedAdam(self.parameters(), lr=self.hparams.lr, betas=(self.hparams.b1, self.hparams.b2),
                         weight_decay=self.hparams.weight_decay)


    def make_fast_step(self, opt, world: int = 1, use_graph: bool = True):
        """The step object ``MiniTrainer.fit`` drives (and bench.py times): training_step + backward replayed
        from one HIP graph, eager fallback inside.  The EMA codebook's batch statistics are all-reduced inside the
        forward when N > 1: that step stays eager."""
        from lgm_hip.graph import ModuleFastStep
        self.prepare_hip(next(self.parameters()).device)
        return ModuleFastStep(self, opt, world, use_graph,
                              collective_inside=isinstance(self.vector_quantizer, VectorQuantizerEMA))


class _VQVAEStepFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, m: VQVAE, x, w_recon, w_vq):
        ctx.set_materialize_grads(False)            # no zero tensors (3 fill launches) for the logged, non-differentiable outputs
        save = bool(ctx.needs_input_grad[0])
        r, tape = m.run(x.detach().float(), save)
        per = r["recon_samples"]
        vals = ops.new((4,), per)                   # (loss, recon, vq, perplexity): one launch, no torch arithmetic
        ops.lib().lgm_vqvae_loss_samples(per.data_ptr(), per.numel(), r["out3"].data_ptr(), w_recon, w_vq, vals.data_ptr(),
                                         ops.stream())
        m.last = r
        ctx.stuff = (m, tape, w_recon, w_vq)
        loss, recon_o, vq_o, ppl_o = vals[0], vals[1], vals[2], vals[3]
        ctx.mark_non_differentiable(recon_o, vq_o, ppl_o)
        return loss, recon_o, vq_o, ppl_o

    @staticmethod
    def backward(ctx, gloss, *_unused):
        m, tape, w_recon, w_vq = ctx.stuff
        if tape is None:
            raise RuntimeError("VQVAE step ran without saving activations")
        gl = gloss.detach().float().reshape(1).contiguous()
        m.backward_hip(tape, gl, w_recon, w_vq)
        ctx.stuff = None
        return None, None, None, None, None
