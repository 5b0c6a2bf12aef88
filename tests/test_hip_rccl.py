"""GPU: the ONE-rank RCCL rehearsal (reference: the implicit DDPStrategy -> NCCL/RCCL process group,
utils/lightning_utils.py:37-43).

Every N > 1 test of the gradient exchange runs on gloo (one GPU per box), so before this file the first RCCL call of the
whole project would have been the driver's 8-GPU run.  Here a child process - started BEFORE anything touches the GPU, as
lgm_hip/launch.py does for real jobs - comes up as rank 0 of a world of ONE on backend ``nccl`` (= RCCL on ROCm) with
``device_id``, and ``LGM_DDP_FORCE=1`` makes ``FlatGradSync`` issue its collectives although world == 1.  An all-reduce
over one rank is the identity, so every result must be ``torch.equal`` to the run without collectives on the same kernel
selection - while the code path is the real one: RCCL communicator set-up, asynchronous all-reduces on RCCL's stream
between the four graph replays, stream ordering around them, HIP-graph captures beside RCCL's watchdog thread
(``capture_error_mode="thread_local"``)."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
PKG = os.path.join(ROOT, "lightning-generative-models_amd")

_WORKER = r'''
import json, os, sys
sys.path.insert(0, sys.argv[1])
import torch
import torch.distributed as dist

dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
torch.cuda.set_device(dev)
import datetime
dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=120))     # RCCL, eager communicator
probe = torch.full((4,), 3.0, device=dev)
dist.all_reduce(probe)
torch.cuda.synchronize()
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1 and float(probe.sum()) == 12.0

from lgm_hip import ops
from lgm_hip.graph import DDPMFastStep
from models.generative.diffusion.ddpm import DDPM

B, STEPS = 16, 12


def run(force, overlap, selection):
    """12 steps of the fast step; -> (parameters, Adam exp_avg, EMA shadow, mode, collectives per step, selection record)"""
    os.environ["LGM_DDP_FORCE"] = "1" if force else "0"
    os.environ["LGM_DDP_OVERLAP"] = "1" if overlap else "0"
    ops.set_kernel_selection(cu_margin=selection[0], light=selection[1])
    torch.manual_seed(10)
    m = DDPM(img_channels=3, img_size=32, dim=64, diffusion_timesteps=1000, lr=2e-5, betas=(0.9, 0.99),
             ema_update_every=10, ema_decay=0.995).to(dev)
    m.sample_every = 0
    m.prepare_hip(dev)
    m.train()
    opt = m.configure_optimizers()
    fast = DDPMFastStep(m, opt, 1, use_graph=True)
    assert (fast.sync is not None) == force
    g = torch.Generator().manual_seed(5)
    torch.manual_seed(77)                                    # the device generator draws (t, noise) inside graph 1
    for i in range(STEPS):
        x = (torch.rand(B, 3, 32, 32, generator=g) * 2 - 1).to(dev)
        fast.step((x, None), i)
    torch.cuda.synchronize()
    net = m.ema.online_model.model
    mom = torch.cat([st["m"].reshape(-1).cpu() for st in opt._flat_state.values()])
    rec = (net._flat.data.clone().cpu(), mom, m.ema.ema_model.model._flat.data.clone().cpu(), fast.mode,
           len(fast.sync.last_buckets) if fast.sync is not None else 0,
           dict(fast.sync.selection) if fast.sync is not None else None,
           len(fast.graphed.graphs) if fast.graphed is not None else 0)
    return rec


out = {}
# (a) overlapped exchange on RCCL: FlatGradSync switches the rank's kernel selection on (margin 16, light workgroups)
a = run(True, True, (-1, -1))
sel = (int(ops.lib().lgm_cu_margin()), 1)
out["overlap_selection"] = a[5]
out["overlap_margin"] = sel[0]
# (c) no collectives, the SAME kernel selection
c = run(False, True, sel)
# (b) one all-reduce after the backward: nothing resident beside it -> one-GPU rules; (d) its twin without collectives
b = run(True, False, (-1, -1))
out["late_selection"] = b[5]
out["late_margin"] = int(ops.lib().lgm_cu_margin())
d = run(False, False, (-1, -1))
out.update(mode_overlap=a[3], mode_late=b[3], graphs_overlap=a[6], graphs_none=c[6],
           collectives_overlap=a[4], collectives_late=b[4],
           overlap_equal=[bool(torch.equal(x, y)) for x, y in zip(a[:3], c[:3])],
           late_equal=[bool(torch.equal(x, y)) for x, y in zip(b[:3], d[:3])],
           moved=float((a[0] - b[0]).abs().max()) >= 0.0, finite=bool(torch.isfinite(a[0]).all()))
print("RCCL_RESULT " + json.dumps(out), flush=True)
dist.destroy_process_group()
'''


def _launch(args, env_extra, timeout=900):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    for k in ("LGM_CU_MARGIN", "LGM_WINO4_LIGHT", "LGM_WINO4_LIGHT_BELOW"):
        env.pop(k, None)                                     # the rule under test must not be overridden from outside
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1",
                           "--master-addr", "127.0.0.1", "--master-port", env_extra.get("_PORT", "29571"), *args],
                          capture_output=True, text=True, timeout=timeout, env=env)


def test_one_rank_rccl_exchange_between_graph_replays_is_the_identity(tmp_path):
    """12 DDPMFastStep steps (B = 16, the per-rank batch of 8 GPUs) with the gradient exchange on RCCL: four graphs per
    step with five overlapped all-reduces, and ``LGM_DDP_OVERLAP=0`` (one all-reduce after the backward) - parameters, Adam
    moments and the EMA shadow ``torch.equal`` to the same steps without any collective, on the same kernel selection.
    Also pins the selection RULE: margin 16 + light workgroups exactly when the exchange overlaps the backward
    on RCCL, the one-GPU rules otherwise."""
    script = tmp_path / "rccl_worker.py"
    script.write_text(_WORKER)
    r = _launch([str(script), PKG], {"_PORT": "29571"})
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-6000:])
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RCCL_RESULT ")]
    assert line, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.loads(line[0][len("RCCL_RESULT "):])
    print("[rccl]", res)
    assert res["finite"]
    assert res["mode_overlap"].startswith("hipGraph replay (4 graphs/step)"), res      # captures survived the watchdog
    assert res["graphs_overlap"] == 4 and res["graphs_none"] == 1
    assert res["collectives_overlap"] == 5 and res["collectives_late"] == 1, res
    assert res["overlap_equal"] == [True, True, True], res
    assert res["late_equal"] == [True, True, True], res
    assert res["overlap_margin"] == 16 and "RCCL" in res["overlap_selection"]["rule"], res
    assert res["late_margin"] == 0 and "LGM_DDP_OVERLAP=0" in res["late_selection"]["rule"], res


def test_bench_line_of_a_one_rank_rccl_run_describes_its_exchange():
    """``bench.py --gpus 1 --only`` under a launcher with LGM_DDP_FORCE=1: the line carries backend nccl, rccl_ranks 1, the
    buckets FlatGradSync issued, the rule that chose the kernels and comm_exposed_ms measured with HIP events."""
    r = _launch([os.path.join(ROOT, "bench.py"), "--gpus", "1", "--only", "--steps", "6", "--warmup", "3",
                 "--no-cpu-baseline", "--batch", "64"], {"_PORT": "29573", "LGM_DDP_FORCE": "1"})
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-6000:])
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    cfg = line["config"]
    print("[rccl bench]", {k: cfg.get(k) for k in ("backend", "rccl_ranks", "bucket_bytes", "collectives_per_step",
                                                     "grad_exchange", "kernel_selection")}, line.get("comm_exposed_ms"))
    assert line["n_gpus"] == 1 and cfg["backend"] == "nccl" and cfg["rccl_ranks"] == 1
    assert cfg["collectives_per_step"] == 5
    assert sum(cfg["bucket_bytes"]) >= 4 * 35719555          # the whole flat buffer (it pads channel counts to 4)
    assert cfg["kernel_selection"]["cu_margin"] == 16 and "RCCL" in cfg["kernel_selection"]["rule"]
    assert line["comm_exposed_ms"] is not None and line["comm_exposed_ms"] >= 0.0
    assert line["value"] > 0
