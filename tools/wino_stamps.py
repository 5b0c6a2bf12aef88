"""Per-phase cycle stamps of the Winograd convolution kernel (diagnostic build).  usage: python tools/wino_stamps.py ci co hw [B]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "lightning-generative-models_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from lgm_hip import ops  # noqa: E402
from wino_bench import wino, wino_weights  # noqa: E402

ci, co, hw = (int(v) for v in sys.argv[1:4])
B = int(sys.argv[4]) if len(sys.argv) > 4 else 128
yx = int(sys.argv[5]) if len(sys.argv) > 5 else 0
mode = int(sys.argv[6]) if len(sys.argv) > 6 else 0
dev = torch.device("cuda", 0)
x = torch.randn(B, hw, hw, ci if not yx else co, device=dev)
y = torch.empty(B, hw, hw, co if not yx else ci, device=dev)
w = torch.randn(co, 9, ci, device=dev) * 0.05
uf, ub = wino_weights(w)
g = ops.make_geom(B, hw, hw, ci, co, 3, 3, 1, 1)
for _ in range(50):
    wino(yx, g, x, ub if yx else uf, None, None, y)
dbg = torch.zeros(256 * 64, dtype=torch.int64, device=dev)
ops.lib().lgm_wino_set_debug_buffer(dbg.data_ptr(), mode)
for _ in range(20):
    wino(yx, g, x, ub if yx else uf, None, None, y)
torch.cuda.synchronize()
ops.lib().lgm_wino_set_debug_buffer(None, 0)
d = dbg.cpu().view(256, 64)
n = int(d[0, 0])
st = d[:, 2:2 + min(n, 62)].double()
dt = (st[:, 1:] - st[:, :-1])
med = dt.median(0).values
print(f"stamps per workgroup: {n};  total cycles median {float((st[:, -1] - st[:, 0]).median()):.0f}  max {float((st[:, -1] - st[:, 0]).max()):.0f}")
print("median cycles per interval (prologue, phases..., epilogues):")
print(" ".join(f"{float(v):.0f}" for v in med))
