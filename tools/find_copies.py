import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "lightning-generative-models_amd")):
    sys.path.insert(0, p)
import torch
from models.generative.diffusion.ddpm import DDPM
from lgm_hip.graph import DDPMFastStep
dev = torch.device("cuda", 0)
torch.manual_seed(0)
m = DDPM(img_channels=3, img_size=32, dim=64); m.sample_every = 0
m.to(dev); m.prepare_hip(dev); m.train()
opt = m.configure_optimizers()
fast = DDPMFastStep(m, opt, 1, use_graph=False)
x = torch.rand(16, 3, 32, 32, device=dev) * 2 - 1
for i in range(3):
    fast.step((x, None), i)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    fast.step((x, None), 3)
    torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::fill_", "aten::zero_", "aten::add_", "aten::_to_copy"):
        st = [s for s in (e.stack or []) if "lightning-generative-models_amd" in s or "bench" in s]
        cnt[(e.name, st[0] if st else "?")] += 1
for (n, s), c in cnt.most_common(30):
    print(c, n, s)
