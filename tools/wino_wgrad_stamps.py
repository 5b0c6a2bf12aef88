import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "lightning-generative-models_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import torch
from lgm_hip import ops
ci, co, hw = (int(v) for v in sys.argv[1:4])
B = int(sys.argv[4]) if len(sys.argv) > 4 else 128
dev = torch.device("cuda", 0)
x = torch.randn(B, hw, hw, ci, device=dev); y = torch.randn(B, hw, hw, co, device=dev)
gw = torch.zeros(co, 9, ci, device=dev); gb = torch.zeros(co, device=dev)
g = ops.make_geom(B, hw, hw, ci, co, 3, 3, 1, 1)
for _ in range(50):
    ops.conv_wgrad(g, y, x, gw.data_ptr(), 0.0, gb.data_ptr())
dbg = torch.zeros(1024 * 64, dtype=torch.int64, device=dev)
ops.lib().lgm_wino_set_debug_buffer(dbg.data_ptr(), 0)
for _ in range(10):
    ops.conv_wgrad(g, y, x, gw.data_ptr(), 0.0, gb.data_ptr())
torch.cuda.synchronize()
ops.lib().lgm_wino_set_debug_buffer(None, 0)
d = dbg.cpu().view(1024, 64)
n = int(d[0, 0]); nb = int((d[:, 0] > 0).sum())
st = d[:nb, 2:2 + min(n, 62)].double()
dt = st[:, 1:] - st[:, :-1]
print(f"blocks {nb} stamps {n} total median {float((st[:, -1] - st[:, 0]).median()):.0f}")
print(" ".join(f"{float(v):.0f}" for v in dt.median(0).values))
