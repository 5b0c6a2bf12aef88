import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "lightning-generative-models_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import torch
from lgm_hip import ops
from wino_bench import wino, wino_weights
dev = torch.device("cuda", 0)
B, hw, ci, co = 1, 16, 32, 64
g = torch.Generator().manual_seed(0)
x = torch.randn(B, hw, hw, ci, generator=g).to(dev)
w = (torch.randn(co, 9, ci, generator=g) * 0.1).to(dev)
uf, ub = wino_weights(w)
geom = ops.make_geom(B, hw, hw, ci, co, 3, 3, 1, 1)
ref = torch.empty(B, hw, hw, co, device=dev)
import torch.nn.functional as F
ref = F.conv2d(x.permute(0, 3, 1, 2).cpu().double(), w.reshape(co, 3, 3, ci).permute(0, 3, 1, 2).cpu().double(), padding=1).permute(0, 2, 3, 1).float()
out = torch.full((B, hw, hw, co), float("nan"), device=dev)
wino(0, geom, x, uf, None, None, out)
o = out.cpu()
print("nan count", int(torch.isnan(o).sum()), "of", o.numel())
err = (o - ref).abs()
print("max err", float(err.max()))
ok = err < 1e-4
print("correct fraction", float(ok.float().mean()))
# which channels / pixels are right?
print("correct per channel (first 64):", ok.float().mean((0, 1, 2))[:64].tolist())
print("correct per row:", ok.float().mean((0, 2, 3)).tolist())
print("correct per col:", ok.float().mean((0, 1, 3)).tolist())
# find where a given output value went: out[0,0,0,:8] vs ref values
flat_ref = ref.reshape(-1)
for (py, px, c) in [(0, 0, 0), (0, 0, 1), (0, 0, 4), (0, 0, 8), (0, 1, 0), (1, 0, 0), (0, 2, 0), (2, 0, 0), (5, 7, 33)]:
    v = o[0, py, px, c].item()
    d = (flat_ref - v).abs()
    i = int(d.argmin())
    rp = i // co
    print(f"out[{py},{px},{c}]={v:+.5f} matches ref at pix ({rp // hw},{rp % hw}) ch {i % co} (diff {float(d.min()):.1e})")
bad = (~ok).nonzero()
print("n bad", bad.shape[0])
for k in range(0, min(bad.shape[0], 400), 40):
    b, py, px, c = bad[k].tolist()
    v = o[b, py, px, c].item()
    d = (flat_ref - v).abs()
    i = int(d.argmin())
    rp = i // co
    print(f"BAD out[{py},{px},{c}]={v:+.5f} (ref {ref[b,py,px,c]:+.5f}) nearest ref at pix ({rp // hw},{rp % hw}) ch {i % co} (diff {float(d.min()):.1e})")
