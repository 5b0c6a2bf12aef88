"""3x3 weight gradient (kernel + its slab reduction) at the UNet's layer shapes; run once as is and once with
LGM_NO_WINO=1 to compare the Winograd and the direct kernels.  usage (GPU box): python tools/wgrad_layers.py [B]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "lightning-generative-models_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from lgm_hip import ops  # noqa: E402
from wino_bench import SHAPES, timeit  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda", 0)
tot = 0.0
print("mode", "direct" if os.environ.get("LGM_NO_WINO") else "winograd", "B", B)
for name, ci, co, hw, cnt in SHAPES:
    x = torch.randn(B, hw, hw, ci, device=dev)
    y = torch.randn(B, hw, hw, co, device=dev)
    gw = torch.zeros(co, 9, ci, device=dev)
    gb = torch.zeros(co, device=dev)
    g = ops.make_geom(B, hw, hw, ci, co, 3, 3, 1, 1)
    t = timeit(lambda: ops.conv_wgrad(g, y, x, gw.data_ptr(), 0.0, gb.data_ptr()), 50)
    fl = 2.0 * B * hw * hw * ci * co * 9
    tot += t * cnt
    print(f"{name:16s} x{cnt}  {t * 1e3:8.1f} us  {fl / t / 1e9:7.1f} TF(alg)", flush=True)
print(f"weighted total {tot:.3f} ms")
