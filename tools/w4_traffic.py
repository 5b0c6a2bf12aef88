"""Isolated launches of one 3x3 layer for counter passes (where does the F(4x4) kernel's fetch traffic
come from?).  Forward only, bias, no residual; every kernel variant five times behind an L2 / Infinity Cache flush.
usage (GPU box):  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/x --output-format csv -- python3 tools/w4_traffic.py B [ci co hw]
then tools/pmc_kernels.py /tmp/x_fetch /tmp/x_write"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "lightning-generative-models_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from lgm_hip import ops  # noqa: E402
from wino_bench import wino, wino_weights  # noqa: E402
from wino4_bench import wino4, wino4_weights  # noqa: E402
from wino4l_bench import wino4l  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ci, co, hw = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (64, 64, 32)
dev = torch.device("cuda", 0)
g = ops.make_geom(B, hw, hw, ci, co, 3, 3, 1, 1)
x = torch.randn(B, hw, hw, ci, device=dev)
y = torch.empty(B, hw, hw, co, device=dev)
w = torch.randn(co, 9, ci, device=dev) * (1.0 / (3 * ci ** 0.5))
bd = torch.randn(co, device=dev)
uf, _ = wino_weights(w)
uf4, _ = wino4_weights(w)
big = torch.empty(96 << 20, device=dev)
L = ops.lib()
L.lgm_wino4_set_light(0)
for rep in range(5):
    for fn in (lambda: wino(0, g, x, uf, bd, None, y), lambda: wino4(0, g, x, uf4, bd, None, y),
               lambda: wino4l(0, g, x, uf4, bd, None, y)):
        big.fill_(0.0)
        torch.cuda.synchronize()
        fn()
        torch.cuda.synchronize()
print("algorithmic bytes: input", x.numel() * 4, "output", y.numel() * 4, "U(F4)", uf4.numel() * 4)
