"""Vector quantisers on the MI355X HIP engine - drop-in for the reference's models/modules/vector_quantizer.py (same class
names, constructor arguments, state_dict keys ``embedding.weight`` / ``_ema_cluster_size`` / ``_ema_embedding``).

Nearest-code search, segmented sums, EMA update and the straight-through backward run on lgm_vq_* (csrc/vq.hip): no [N, K]
distance or one-hot matrices, int64 indices equal to the reference's argmin, deterministic sums for the EMA statistics and
the codebook gradient.
"""
from __future__ import annotations

import torch
import torch.distributed as dist
from torch import nn

from lgm_hip import ops
from lgm_hip.nn import GradCtx


class _Embedding(nn.Module):
    def __init__(self, num_embeddings, embedding_dim):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(num_embeddings, embedding_dim).uniform_(
            -1 / num_embeddings, 1 / num_embeddings))          # vector_quantizer.py:39-43


def _world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


class VectorQuantizer(nn.Module):
    """reference vector_quantizer.py:8-93 (state: ``embedding.weight``)."""

    use_ema = False

    def __init__(self, num_embeddings, embedding_dim, commitment_cost=0.25):
        super().__init__()
        self.num_embeddings, self.embedding_dim, self.commitment_cost = num_embeddings, embedding_dim, commitment_cost
        self.embedding = _Embedding(num_embeddings, embedding_dim)

    def fwd(self, lat, training: bool):
        """lat: [B,H,W,D] dense NHWC.  Returns (q, scalars[3] = vq_loss, perplexity, mse, saved)."""
        B, H, W, D = lat.shape
        N, K = B * H * W, self.num_embeddings
        L = ops.lib()
        st = ops.stream()
        fp = self.embedding.weight._lgm_flat
        cb = fp.ptr(self.embedding.weight)
        idx = torch.empty(N, dtype=torch.long, device=lat.device)
        L.lgm_vq_assign(lat.data_ptr(), D, cb, N, K, D, idx.data_ptr(), None, st)
        dw = ops.new((K, D), lat)
        counts = ops.new((K,), lat)
        L.lgm_vq_segment_sum(lat.data_ptr(), D, idx.data_ptr(), N, K, D, dw.data_ptr(), counts.data_ptr(), st)
        if self.use_ema and training:      # codebook is replaced BEFORE the lookup (:168-177)
            cnt_u, dw_u = counts, dw
            if _world_size() > 1:
                # Deliberate deviation: upstream updates the codebook Parameter from per-rank
                # statistics while DDP only re-broadcasts the EMA buffers, so ranks drift.  Here the batch
                # statistics (count[K], dw[K,D]: 133 KB) are summed over ranks first; every rank then applies
                # the identical update.  Single-GPU arithmetic is unchanged.
                stat = torch.cat([counts.reshape(-1), dw.reshape(-1)])
                dist.all_reduce(stat)
                cnt_u, dw_u = stat[:K], stat[K:].view(K, D)
            L.lgm_vq_ema_update(self._ema_cluster_size.data_ptr(), self._ema_embedding.data_ptr(), cb,
                                cnt_u.data_ptr(), dw_u.data_ptr(), K, D, self.decay, self.epsilon, st)
        q = ops.new(lat.shape, lat)
        out3 = ops.new((3,), lat)
        ws = ops.workspace(L.lgm_vq_gather_workspace(N, D), lat.device)
        L.lgm_vq_gather_loss(lat.data_ptr(), D, cb, idx.data_ptr(), counts.data_ptr(), N, K, D,
                             self.commitment_cost, q.data_ptr(), D, out3.data_ptr(), ws.data_ptr(), st)
        return q, out3, (lat, q, idx, dw, counts)

    def bwd(self, gc: GradCtx, saved, gq, g_vq):
        lat, q, idx, dw, counts = saved
        B, H, W, D = lat.shape
        N, K = B * H * W, self.num_embeddings
        fp = gc.flat
        w = self.embedding.weight
        glat = ops.new(lat.shape, lat)
        ops.lib().lgm_vq_bwd(lat.data_ptr(), D, q.data_ptr(), D, gq.data_ptr(), D, fp.ptr(w), dw.data_ptr(),
                             counts.data_ptr(), g_vq.data_ptr(), self.commitment_cost, N, K, D, glat.data_ptr(), D,
                             fp.gptr(w), gc.beta(w), ops.stream())
        return glat


class VectorQuantizerEMA(VectorQuantizer):
    """reference vector_quantizer.py:96-179 (buffers ``_ema_cluster_size``, ``_ema_embedding``)."""

    use_ema = True

    def __init__(self, num_embeddings, embedding_dim, commitment_cost=0.25, decay=0.99, epsilon=1e-5):
        super().__init__(num_embeddings, embedding_dim, commitment_cost)
        self.register_buffer("_ema_cluster_size", torch.zeros(num_embeddings))
        self.register_buffer("_ema_embedding", self.embedding.weight.data.clone())
        self.decay, self.epsilon = decay, epsilon
