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
x = torch.randn(B, hw, hw, ci, generator=g)
w = torch.randn(co, 9, ci, generator=g) * 0.1
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1.]]).double()
Bt = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1.]]).double()
g4 = w.reshape(co, 3, 3, ci).double()
U = torch.einsum("ia,nabc,jb->ijnc", G, g4, G)            # [4,4,co,ci]
xp = torch.nn.functional.pad(x.double(), (0, 0, 1, 1, 1, 1))   # [B,18,18,ci]
M = torch.zeros(16, B, hw // 2, hw // 2, co, dtype=torch.double)
for ty in range(hw // 2):
    for tx in range(hw // 2):
        d = xp[:, 2 * ty:2 * ty + 4, 2 * tx:2 * tx + 4, :]          # [B,4,4,ci]
        V = torch.einsum("ia,bacn,jc->bijn", Bt, d, Bt)             # [B,4,4,ci]
        M[:, :, ty, tx, :] = torch.einsum("bijc,ijnc->ijbn", V, U).reshape(16, B, co)
xd, wd = x.to(dev), w.to(dev)
uf, ub = wino_weights(wd)
geom = ops.make_geom(B, hw, hw, ci, co, 3, 3, 1, 1)
dbg = torch.zeros(256 * 64, dtype=torch.int64, device=dev)
for q in [0, 9]:
    out = torch.full((B, hw, hw, co), float("nan"), device=dev)
    ops.lib().lgm_wino_set_debug_buffer(dbg.data_ptr(), 3 + q)
    wino(0, geom, xd, uf, None, None, out)
    torch.cuda.synchronize()
    ops.lib().lgm_wino_set_debug_buffer(None, 0)
    o = out.cpu().double()
    for dy in range(2):
        for dx in range(2):
            xi = (4 * q + 2 * dy + dx) if q < 4 else 3
            got = o[:, dy::2, dx::2, :]
            ref = M[xi]
            bad = (got - ref).abs() > 1e-4
            print(f"xi {xi:2d}: bad {int(bad.sum()):5d} of {bad.numel()}  bad tiles tx: {sorted(set(bad.nonzero()[:,2].tolist()))[:10]} ty: {sorted(set(bad.nonzero()[:,1].tolist()))[:10]} ch%4: {sorted(set((bad.nonzero()[:,3] % 4).tolist()))}")
