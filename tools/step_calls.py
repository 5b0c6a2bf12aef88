"""Every convolution-family call of ONE eager DDPM training step with its geometry-derived work and its measured time:
where the step is far from its roofs, call by call.  ideal = max(executed FLOPs / 157.3 TFLOP/s, algorithmic bytes / 5 TB/s).
usage (GPU box): python tools/step_calls.py [B] > gpurun_out/step_calls_bB.csv"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "lightning-generative-models_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import bench  # noqa: E402
from lgm_hip import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda", 0)
torch.manual_seed(10)
args = type("A", (), {"no_graph": True})()
step, eager_step, info = bench.setup_ddpm(args, dev, 1, 0, 32, B)
for i in range(4):
    eager_step(i)
torch.cuda.synchronize()
busy = torch.randn(6144, 6144, device=dev)
for _ in range(16):
    busy = (busy @ busy) * 1e-4
ops.TIMER = ops.KernelTimer()
eager_step(10)
torch.cuda.synchronize()
recs = ops.TIMER.records
ops.TIMER = None
print("idx,family,kernel,gflop_alg,gflop_exec,mbytes_alg,us,ideal_us,gap_us")
tot = [0.0, 0.0]
for i, (fam, kern, flops, nbytes, s, e) in enumerate(recs):
    us = s.elapsed_time(e) * 1e3
    ex = flops / bench.mfma_factor(kern)
    ideal = max(ex / 157.3e12, nbytes / 5e12) * 1e6
    tot[0] += us
    tot[1] += ideal
    print(f"{i},{fam},\"{kern}\",{flops / 1e9:.3f},{ex / 1e9:.3f},{nbytes / 1e6:.2f},{us:.1f},{ideal:.1f},{us - ideal:.1f}")
print(f"# {len(recs)} calls, {tot[0] / 1e3:.3f} ms measured, {tot[1] / 1e3:.3f} ms ideal", file=sys.stderr)
