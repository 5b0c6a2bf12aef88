"""The VQ-VAE's residual stack on the MI355X HIP engine - drop-in for the reference's models/modules/residual.py
(same class names, constructor arguments and state_dict keys ``layers.<i>.block.<1|3>.weight``).

Every ReLU rides in a convolution epilogue (lgm_conv_xy_post) or in an input gradient's mask; on the 4 x 4 maps of the
32 x 32 configuration the whole stack is ONE launch (csrc/resstack.hip: an image pair per workgroup, the weights from L2).
"""
from __future__ import annotations

from torch import nn

from lgm_hip import ops
from lgm_hip.nn import Conv2d


class ResidualBlock(nn.Module):
    """reference residual.py:5-21 — note the in-place first ReLU: the block returns
    relu(x) + conv1x1(relu(conv3x3(relu(x))))."""

    def __init__(self, in_channels, hidden_dim, num_residual_hiddens):
        super().__init__()
        self.block = nn.Sequential(nn.Identity(), Conv2d(in_channels, num_residual_hiddens, 3, padding=1, bias=False),
                                   nn.Identity(), Conv2d(num_residual_hiddens, hidden_dim, 1, bias=False))


class ResidualStack(nn.Module):
    def __init__(self, in_channels, hidden_dim, num_residual_layers, num_residual_hiddens):
        super().__init__()
        self.layers = nn.ModuleList([ResidualBlock(in_channels, hidden_dim, num_residual_hiddens)
                                     for _ in range(num_residual_layers)])

    def fwd(self, cur, save):
        """``cur`` arrives with the first block's (in-place) ReLU already applied by the epilogue of the convolution
        that produced it; every ReLU in here rides in a convolution epilogue too (lgm_conv_xy_post):
        y = relu(conv3x3(cur)), cur' = relu(conv1x1(y) + cur) - the next block's in-place ReLU, or the stack's final one."""
        tape = []
        c3 = self.layers[0].block[1]
        if c3.weight.shape[1] == cur.shape[-1] and c3.weight.shape[1] == self.layers[0].block[3].weight.shape[0]:
            fp = c3.weight._lgm_flat
            r = ops.resstack_fwd(cur, [fp.ptr(b.block[1].weight) for b in self.layers],
                                 [fp.ptr(b.block[3].weight) for b in self.layers], c3.weight.shape[0])
            if r is not None:                     # the whole stack in one launch (4 x 4 maps of the 32 x 32 configuration)
                for y, z in zip(*r):
                    tape.append((cur, y))
                    cur = z
                return cur, (tape, cur)
        for blk in self.layers:
            y = blk.block[1].fwd(cur, act=ops.ACT_RELU)
            z = blk.block[3].fwd(y, res=cur, act=ops.ACT_RELU)
            tape.append((cur, y))
            cur = z
        return cur, (tape, cur)

    def bwd(self, gc, saved, g):
        """``g`` arrives already multiplied by the final ReLU's mask (the consumer's input gradient applied it in its
        epilogue, mask = the stack's output); returns the gradient w.r.t. the stack's (ReLU'd) input, again with that
        ReLU's mask applied - every activation backward is an epilogue mask of the input gradient before it."""
        tape, out = saved
        for blk, (r, y) in zip(reversed(self.layers), reversed(tape)):
            gy = blk.block[3].bwd(gc, y, g, mask=y)
            blk.block[1].bwd(gc, r, gy, g, True, mask=r)                # g = (g + dgrad(gy)) * relu'(r)
        return g
