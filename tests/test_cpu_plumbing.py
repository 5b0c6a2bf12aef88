"""CPU: config/registry/CLI plumbing (BASELINE config 1: VAE through train.py on the CPU accelerator),
loader error behaviour (reference utils/loader.py:47-86), flat parameter storage, FusedAdam surface,
and the N>1 data-parallel path with world_size-2 gloo processes."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "lightning-generative-models_amd")


def test_load_config_checks_and_registry(tmp_path):
    from utils.loader import load_config, load_model
    cfg = load_config(os.path.join(PKG, "configs", "diffusion", "ddpm.json"))
    assert cfg["model"]["name"] == "DDPM" and cfg["model"]["args"]["dim"] == 64
    bad = dict(cfg)
    bad["dataset"] = dict(cfg["dataset"], img_size=48)
    p = tmp_path / "bad.json"
    p.write_text(json.dumps(bad))
    with pytest.raises(ValueError, match="img_size"):
        load_config(str(p))
    p.write_text("{ not json")
    with pytest.raises(ValueError, match="not a valid JSON"):
        load_config(str(p))
    with pytest.raises(FileNotFoundError):
        load_config(str(tmp_path / "missing.json"))
    with pytest.raises(ValueError, match="Failed to import"):
        load_model({"name": "NoSuchModel", "args": {}})
    for rel in ("gan/wgan_gp.json", "gan/wgan_gp_celeba.json", "vae/vqvae.json", "vae/vqvae_ema.json", "vae/vae.json",
                "diffusion/ddim.json", "gan/dcgan.json", "gan/dcgan_mnist.json", "gan/lsgan.json", "gan/r1gan.json",
                "gan/wgan_cp.json"):
        c = load_config(os.path.join(PKG, "configs", rel))
        m = load_model(c["model"])          # constructs on CPU without touching the GPU
        assert type(m).__name__ == c["model"]["name"]


def test_train_entry_vae_cpu():
    """python train.py --config_path configs/vae/vae.json on the CPU accelerator (config 1)."""
    r = subprocess.run([sys.executable, os.path.join(PKG, "train.py"), "--config_path",
                        os.path.join(PKG, "configs", "vae", "vae.json"), "--max_steps", "20", "--accelerator", "cpu",
                        "--experiment_name", "pytest_cpu"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    ck = os.path.join(PKG, "experiments", "VAE", "pytest_cpu", "last.ckpt")
    sd = torch.load(ck, map_location="cpu")
    assert sd["global_step"] == 20 and "encoder.mu.weight" in sd["state_dict"]


def test_train_entry_with_pytorch_lightning_installed(tmp_path):
    """The reference's own environment HAS pytorch_lightning.  A fake package whose LightningModule breaks
    outside a pl.Trainer (log / optimizers / global_step go through self.trainer) is put on the path: the
    models must still derive from the in-repo module and train.py must still run, write periodic + final
    checkpoints atomically, run validation at the epoch end and keep the best val_loss checkpoint."""
    fake = tmp_path / "site" / "pytorch_lightning"
    fake.mkdir(parents=True)
    (fake / "__init__.py").write_text(
        "import torch.nn as nn\n"
        "class LightningModule(nn.Module):\n"
        "    def log(self, *a, **k):\n        raise RuntimeError('log() outside a pl.Trainer loop')\n"
        "    def log_dict(self, *a, **k):\n        raise RuntimeError('log_dict() outside a pl.Trainer loop')\n"
        "    def optimizers(self):\n        return self.trainer.optimizers\n"
        "    @property\n    def global_step(self):\n        return self.trainer.global_step\n"
        "class Trainer:\n    def __init__(self, *a, **k):\n        raise RuntimeError('pl.Trainer must not be used')\n")
    env = dict(os.environ, PYTHONPATH=str(tmp_path / "site") + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, os.path.join(PKG, "train.py"), "--config_path",
                        os.path.join(PKG, "configs", "vae", "vae.json"), "--max_epochs", "2", "--accelerator", "cpu",
                        "--check_val_every_n_epoch", "1", "--experiment_name", "pytest_cpu_pl"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "val_loss" in r.stdout
    d = os.path.join(PKG, "experiments", "VAE", "pytest_cpu_pl")
    sd = torch.load(os.path.join(d, "last.ckpt"), map_location="cpu")
    assert sd["epoch"] == 2 and sd["global_step"] > 0
    best = [f for f in os.listdir(d) if f.startswith("epoch=") and f.endswith(".ckpt")]
    assert len(best) == 1                                     # save_top_k = 1 on val_loss
    assert not [f for f in os.listdir(d) if ".tmp." in f]     # atomic writes leave nothing behind
    chk = subprocess.run([sys.executable, "-c",
                          "import sys; sys.path.insert(0, sys.argv[1]); import lgm_hip.lightning as L; "
                          "from models.generative.vae.vae import VAE; "
                          "assert L.HAVE_PL and issubclass(VAE, L.MiniLightningModule); print('ok')", PKG],
                         capture_output=True, text=True, env=env, timeout=120)
    assert chk.returncode == 0 and "ok" in chk.stdout, chk.stderr[-1500:]


def test_trainer_periodic_checkpoint_interrupt_and_accumulation_tail(tmp_path):
    """MiniTrainer: last.ckpt every ``ckpt_every_n_steps`` optimizer steps and on an exception raised inside
    the loop (max_steps = max_epochs = -1, the CLI defaults, never reach the final save); the leftover
    micro-batches of an epoch are stepped (Lightning semantics for accumulate_grad_batches)."""
    from lgm_hip.lightning import MiniTrainer
    from models.generative.vae.vae import VAE
    torch.manual_seed(0)

    class Boom(Exception):
        pass

    def batches(n, boom_at=None):
        g = torch.Generator().manual_seed(1)
        for i in range(n):
            if boom_at is not None and i == boom_at:
                raise Boom()
            yield torch.rand(4, 1, 8, 8, generator=g) * 2 - 1, torch.zeros(4, dtype=torch.long)

    m = VAE(img_channels=1, img_size=8, latent_dim=4)
    tr = MiniTrainer(default_root_dir=str(tmp_path), log_every=0, device="cpu", ckpt_every_n_steps=3)
    with pytest.raises(Boom):
        tr.fit(m, train_dataloader=batches(100, boom_at=8))
    sd = torch.load(os.path.join(str(tmp_path), "last.ckpt"), map_location="cpu")
    assert sd["global_step"] == 8                             # saved by the exception handler (periodic: 3, 6)
    m2 = VAE(img_channels=1, img_size=8, latent_dim=4)
    tr2 = MiniTrainer(max_epochs=1, log_every=0, device="cpu", accumulate_grad_batches=4)
    tr2.fit(m2, train_dataloader=list(batches(10)))
    assert m2.global_step == 3                                # 4 + 4 + the 2 leftover micro-batches


def test_flat_params_views_and_state_dict_roundtrip():
    from lgm_hip.flat import FlatParams
    from lgm_hip.nn import Conv2d, param_kind
    conv = Conv2d(3, 5, 3, padding=1)
    w0, b0 = conv.weight.detach().clone(), conv.bias.detach().clone()
    fp = FlatParams([(n, p, param_kind(n, p)) for n, p in conv.named_parameters()], "cpu")
    assert conv.weight.shape == (5, 3, 3, 3) and torch.equal(conv.weight, w0) and torch.equal(conv.bias, b0)
    # physical layout [Np=8][T=9][Cp=4], padding lanes zero
    phys = fp.data[: 8 * 9 * 4].view(8, 9, 4)
    assert torch.equal(phys[:5, :, :3], w0.permute(0, 2, 3, 1).reshape(5, 9, 3))
    assert float(phys[5:].abs().sum()) == 0 and float(phys[:, :, 3].abs().sum()) == 0
    sd = {k: v.clone() for k, v in conv.state_dict().items()}
    conv.load_state_dict({k: v * 2 for k, v in sd.items()})
    assert torch.equal(conv.weight, w0 * 2) and fp.still_bound()
    assert fp.begin_backward() == 0.0 and fp.begin_backward() == 1.0
    fp.zero_grad()
    assert fp.begin_backward() == 0.0


_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from lgm_hip.lightning import MiniTrainer
from models.generative.vae.vae import VAE
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
torch.manual_seed(0)
m = VAE(img_channels=1, img_size=8, latent_dim=4)
g = torch.Generator().manual_seed(1)
x = torch.rand(8, 1, 8, 8, generator=g) * 2 - 1
mu_noise = torch.randn(8, 4, generator=g)
def loss_of(model, xs, noise):
    mu, lv = model.encoder(xs)
    xh = model.decoder(mu + noise * torch.exp(lv / 2))
    return torch.nn.functional.l1_loss(xh, xs)
# every rank: its shard of the global batch (8 / world)
n = 8 // world
loss = loss_of(m, x[rank * n:(rank + 1) * n], mu_noise[rank * n:(rank + 1) * n])
loss.backward()
MiniTrainer().allreduce_grads(m)
ref = VAE(img_channels=1, img_size=8, latent_dim=4)
ref.load_state_dict(m.state_dict())
loss_of(ref, x, mu_noise).backward()
err = max(float((p.grad - q.grad).abs().max()) for p, q in zip(m.parameters(), ref.parameters()))
# flat-storage path: averaged in ONE all-reduce per network
from lgm_hip.flat import FlatParams
from lgm_hip.nn import Conv2d, param_kind
conv = Conv2d(4, 4, 1)
fp = FlatParams([(k, p, param_kind(k, p)) for k, p in conv.named_parameters()], "cpu")
fp.grad.fill_(float(rank + 1))
MiniTrainer().allreduce_grads(conv)
ok = bool(torch.allclose(fp.grad, torch.full_like(fp.grad, (world + 1) / 2)))
# bucketed async exchange used by bench.py (buckets launched while "backward" is still producing others)
from lgm_hip.lightning import FlatGradSync
fp.grad.copy_(torch.arange(fp.total, dtype=torch.float32) * (rank + 1))
sync = FlatGradSync(fp)
third = fp.total // 3
sync.ready(2 * third, fp.total)
sync.ready(third, 2 * third)
sync.ready(0, third)
sync.finish()
expect = torch.arange(fp.total, dtype=torch.float32) * sum(range(1, world + 1))
ok = ok and bool(torch.allclose(fp.grad * sync.grad_scale, expect / world))
# LGM_DDP_OVERLAP=0: the same slices noted during the "backward", ONE all-reduce of the whole buffer in finish()
fp.grad.copy_(torch.arange(fp.total, dtype=torch.float32) * (rank + 1))
late = FlatGradSync(fp, overlap=False)
calls = []
real = dist.all_reduce
dist.all_reduce = lambda t, **kw: (calls.append(t.numel()), real(t, **kw))[1]
assert late.ready(2 * third, fp.total) is None and late.ready(third, 2 * third) is None and late.ready(0, third) is None
assert not calls
late.finish()
dist.all_reduce = real
ok = ok and calls == [fp.total] and bool(torch.allclose(fp.grad * late.grad_scale, expect / world))
print(f"RANK{rank} err={err:.3e} flat_ok={ok}", flush=True)
dist.destroy_process_group()
'''


def test_downsample_weight_is_exchanged_in_the_reference_shape():
    """ddpm._DownConv holds Downsample's weight (reference ddpm.py:100-104) as [N, C, 2, 2]; its row-major flattening is the
    reference's [N, 4 C, 1, 1] tensor (input channel 'c p1 p2').  state_dict / load_state_dict use the reference's shape, on
    plain parameters and on parameters bound to flat storage; the arithmetic is the reference's (Rearrange + 1x1)."""
    import torch.nn.functional as F
    from models.generative.diffusion.ddpm import Unet, _Down
    torch.manual_seed(0)
    d = _Down(6, 10)
    conv = d._modules["1"]
    assert conv.weight.shape == (10, 6, 2, 2) and conv.ref_shape == (10, 24, 1, 1)
    sd = d.state_dict()
    assert sd["1.weight"].shape == (10, 24, 1, 1) and sd["1.bias"].shape == (10,)
    assert torch.equal(sd["1.weight"].reshape(10, 6, 2, 2), conv.weight)
    x = torch.randn(2, 6, 8, 8)
    b, c, hh, ww = x.shape                                      # the reference's Rearrange 'b c (h p1) (w p2) -> b (c p1 p2) h w'
    lo = x.view(b, c, hh // 2, 2, ww // 2, 2).permute(0, 1, 3, 5, 2, 4).reshape(b, 4 * c, hh // 2, ww // 2)
    want = F.conv2d(lo, sd["1.weight"], sd["1.bias"])
    got = F.conv2d(x, conv.weight, conv.bias, stride=2)
    assert torch.allclose(got, want, rtol=1e-5, atol=1e-6)
    w_new = torch.randn(10, 24, 1, 1)
    d.load_state_dict({"1.weight": w_new, "1.bias": sd["1.bias"]}, strict=True)
    assert torch.equal(conv.weight, w_new.reshape(10, 6, 2, 2))
    d.load_state_dict({"1.weight": w_new.reshape(10, 6, 2, 2) * 2, "1.bias": sd["1.bias"]}, strict=True)   # its own shape too
    assert torch.equal(conv.weight, w_new.reshape(10, 6, 2, 2) * 2)
    # a whole network: reference-keyed, reference-shaped dictionary in and out, also once the parameters are flat-bound
    from oracle import diffusion as OD
    P = OD.unet_init(dim=8, channels=3, seed=3)
    net = Unet(dim=8, channels=3)
    net.load_state_dict(P, strict=True)
    fp = net.prepare_hip("cpu")
    out = net.state_dict()
    assert set(out) == set(P) and all(out[k].shape == P[k].shape and torch.equal(out[k], P[k]) for k in P)
    changed = [n for n, q in net.named_parameters() if q.shape != P[n].shape]
    assert changed == ["downs.0.3.1.weight", "downs.1.3.1.weight", "downs.2.3.1.weight"]
    for n in changed:
        assert fp.slot(dict(net.named_parameters())[n]).ref_shape == tuple(P[n].shape)
    net.load_state_dict({k: v * 3 for k, v in P.items()}, strict=True)
    assert fp.still_bound() and torch.equal(net.state_dict()[changed[0]], P[changed[0]] * 3)


def test_data_parallel_gloo_world2(tmp_path):
    """N-rank averaged gradients == 1-rank gradients on the concatenated batch (gloo, 2 ranks)."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29541", str(script), PKG],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    # the two ranks share one stdout pipe: their lines can arrive glued together
    import re
    recs = re.findall(r"RANK(\d) err=([0-9.eE+-]+) flat_ok=(True|False)", r.stdout)
    assert sorted(rk for rk, _, _ in recs) == ["0", "1"], r.stdout
    for rk, err, ok in recs:
        assert float(err) < 1e-6 and ok == "True", (rk, err, ok)


_WORKER_SEMANTICS = r'''
import os, sys, torch, torch.distributed as dist
from torch import nn
sys.path.insert(0, sys.argv[1])
from lgm_hip.lightning import BufferSync, MiniLightningModule, MiniTrainer, multi_rank
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()

class Net(MiniLightningModule):
    """two BatchNorm layers (per-rank running statistics, the DCGAN situation) + one constant buffer"""
    def __init__(self):
        super().__init__()
        self.lin = nn.Linear(4, 4)
        self.bn1, self.bn2 = nn.BatchNorm1d(4), nn.BatchNorm1d(4)
        self.register_buffer("table", torch.arange(5, dtype=torch.float32))
    def training_step(self, batch, batch_idx):
        x, = batch
        loss = self.bn2(self.lin(self.bn1(x))).square().mean()
        self.log("train_loss", loss, sync_dist=multi_rank())
        self.log("rank_plain", torch.tensor(float(rank)))            # not synced: stays per-rank
        return loss
    def configure_optimizers(self):
        return torch.optim.SGD(self.parameters(), lr=0.1)

torch.manual_seed(0)
m = Net()
# --- C2: buffers of rank 0 re-broadcast before every training forward, as ONE flat tensor ------------------------
bs = BufferSync(m)
assert bs.flat is not None and bs.still_packed()
names = dict(m.named_buffers())
lo = bs.flat.data_ptr()
assert all(lo <= b.data_ptr() < lo + 4 * bs.flat.numel() for k, b in names.items() if b.dtype.is_floating_point)   # views of ONE block
sd_keys = sorted(m.state_dict().keys())
m.bn1.running_mean.fill_(float(rank + 1)); m.bn2.running_var.fill_(10.0 * (rank + 1))
bs.broadcast()
c2 = bool(torch.all(m.bn1.running_mean == 1.0)) and bool(torch.all(m.bn2.running_var == 10.0))
c2 = c2 and sorted(m.state_dict().keys()) == sd_keys and torch.equal(m.table, torch.arange(5.0))
# --- through the trainer: ranks see different data, running statistics still agree after fit -------------------------
g = torch.Generator().manual_seed(100 + rank)
data = [(torch.randn(8, 4, generator=g) * (rank + 1),) for _ in range(6)]
tr = MiniTrainer(max_epochs=1, log_every=3, device="cpu")
tr.fit(m, train_dataloader=data)
stat = torch.cat([m.bn1.running_mean, m.bn1.running_var, m.bn2.running_mean, m.bn2.running_var])
gathered = [torch.zeros_like(stat) for _ in range(world)]
dist.all_gather(gathered, stat)
# rank r != 0 holds: rank 0's statistics before the last forward, updated with its OWN last batch (DDP semantics);
# one more broadcast makes them identical
tr.buffer_sync.broadcast()
stat2 = torch.cat([m.bn1.running_mean, m.bn1.running_var, m.bn2.running_mean, m.bn2.running_var])
g2 = [torch.zeros_like(stat2) for _ in range(world)]
dist.all_gather(g2, stat2)
c2 = c2 and all(torch.equal(t, g2[0]) for t in g2) and torch.equal(g2[0], gathered[0])
# parameters identical on every rank (gradient averaging), i.e. the exchange ran
pv = torch.cat([p.detach().reshape(-1) for p in m.parameters()])
gp = [torch.zeros_like(pv) for _ in range(world)]
dist.all_gather(gp, pv)
c1 = all(torch.allclose(t, gp[0], atol=0, rtol=0) for t in gp)
# --- C3: sync_dist scalars = mean over ranks, ONE collective per interval -------------------------------------------
m.logged.clear()
m.log("a", torch.tensor(float(rank)), sync_dist=True)
m.log_dict({"b": torch.tensor(2.0 * rank), "c": 7.0}, sync_dist=True)
m.log("mine", torch.tensor(100.0 + rank))
calls = []
orig = dist.all_reduce
def counting(*a, **k):
    calls.append(1)
    return orig(*a, **k)
dist.all_reduce = counting
out = m.synced_logs()
dist.all_reduce = orig
mean_r = sum(range(world)) / world
c3 = (abs(out["a"] - mean_r) < 1e-6 and abs(out["b"] - 2 * mean_r) < 1e-6 and out["c"] == 7.0
      and out["mine"] == 100.0 + rank and len(calls) == 1)
print(f"RANK{rank} c1={c1} c2={c2} c3={c3}", flush=True)
dist.destroy_process_group()
'''


def test_ddp_side_semantics_gloo_world2(tmp_path):
    """SURVEY §2.1 C2 / C3 (reference DDPStrategy, utils/lightning_utils.py:37-43; sync_dist ddpm.py:1017-1023):
    rank 0's buffers are re-broadcast before every training forward as one flat block; scalars logged with
    sync_dist=True are reported as the mean over ranks from one batched all-reduce per logging interval."""
    script = tmp_path / "worker_sem.py"
    script.write_text(_WORKER_SEMANTICS)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29547", str(script), PKG],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    import re
    recs = re.findall(r"RANK(\d) c1=(True|False) c2=(True|False) c3=(True|False)", r.stdout)
    assert sorted(rk for rk, *_ in recs) == ["0", "1"], r.stdout
    for rec in recs:
        assert rec[1:] == ("True", "True", "True"), rec


def test_max_steps_mid_epoch_is_not_a_completed_epoch(tmp_path):
    """A run stopped by max_steps inside an epoch records that epoch as NOT completed: a resume from its last.ckpt
    runs the epoch again instead of skipping the rest of it (and max_epochs accounting stays right)."""
    from lgm_hip.lightning import MiniTrainer
    from models.generative.vae.vae import VAE
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(1)
    data = [(torch.rand(4, 1, 8, 8, generator=g) * 2 - 1, torch.zeros(4, dtype=torch.long)) for _ in range(5)]
    m = VAE(img_channels=1, img_size=8, latent_dim=4)
    MiniTrainer(max_steps=7, default_root_dir=str(tmp_path), log_every=0, device="cpu").fit(m, train_dataloader=data)
    sd = torch.load(os.path.join(str(tmp_path), "last.ckpt"), map_location="cpu")
    assert sd["global_step"] == 7 and sd["epoch"] == 1        # epoch 0 complete (5 steps), epoch 1 cut after 2
    assert not [f for f in os.listdir(str(tmp_path)) if ".tmp." in f]


def _no_launcher_env(**extra):
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "LOCAL_WORLD_SIZE")}
    env.update(extra)
    return env


def test_train_entry_starts_its_own_ranks_cpu():
    """``python train.py --strategy ddp --devices 2`` with NO launcher in the environment becomes the parent of two ranks
    (gloo on the CPU accelerator) - what Lightning does for the reference's ``python train.py`` on a multi-GPU node
    (reference train.py:38,124-141, utils/lightning_utils.py:37-43).  Both ranks train, rank 0 writes last.ckpt."""
    r = subprocess.run([sys.executable, os.path.join(PKG, "train.py"), "--config_path",
                        os.path.join(PKG, "configs", "vae", "vae.json"), "--max_steps", "6", "--accelerator", "cpu",
                        "--strategy", "ddp", "--devices", "2", "--experiment_name", "pytest_cpu_spawn2"],
                       capture_output=True, text=True, timeout=600, env=_no_launcher_env())
    assert r.returncode == 0, r.stderr[-3000:]
    assert "[launch] starting 2 ranks" in r.stderr
    sd = torch.load(os.path.join(PKG, "experiments", "VAE", "pytest_cpu_spawn2", "last.ckpt"), map_location="cpu")
    assert sd["global_step"] == 6
    # one rank by default on the CPU accelerator (the reference's SingleDeviceStrategy branch), no launcher involved
    r1 = subprocess.run([sys.executable, os.path.join(PKG, "train.py"), "--config_path",
                         os.path.join(PKG, "configs", "vae", "vae.json"), "--max_steps", "2", "--accelerator", "cpu",
                         "--experiment_name", "pytest_cpu_spawn1"], capture_output=True, text=True, timeout=300,
                        env=_no_launcher_env())
    assert r1.returncode == 0 and "[launch]" not in r1.stderr, r1.stderr[-2000:]


_WORKER_DIES = r'''
import os, sys, time
rank = int(os.environ["RANK"])
assert int(os.environ["WORLD_SIZE"]) == 2
if rank == 1:
    sys.exit(7)              # a rank that dies ...
time.sleep(600)              # ... while the other would wait for ever
'''

_WORKER_LOG_MISMATCH = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from lgm_hip.lightning import MiniLightningModule
dist.init_process_group("gloo")
rank = dist.get_rank()
m = MiniLightningModule()
m.log("loss", torch.tensor(1.0 + rank), sync_dist=True)
out = m.synced_logs()
assert abs(out["loss"] - 1.5) < 1e-9, out
m.log("loss", torch.tensor(3.0), sync_dist=True)
if rank == 0:
    m.log("only_rank0", torch.tensor(1.0), sync_dist=True)      # rank-dependent name: must fail on BOTH ranks, not hang
try:
    m.synced_logs()
    print(f"RANK{rank} no-error", flush=True)
except RuntimeError as e:
    print(f"RANK{rank} raised: {str(e)[:60]}", flush=True)
# a name registered everywhere but not logged this interval on one rank: mean over the ranks that logged it
m2 = MiniLightningModule()
m2._sync_dist_names.update(["a", "b"])
m2.logged["a"] = float(rank)
if rank == 1:
    m2.logged["b"] = 10.0
o2 = m2.synced_logs()
print(f"RANK{rank} a={o2['a']} b={o2['b']}", flush=True)
dist.destroy_process_group()
'''


def test_launcher_counts_devices_without_touching_them_and_reports_a_dead_rank(tmp_path, monkeypatch):
    from lgm_hip import launch
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1,2")
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "0,1,2,3")
    assert launch.visible_gpu_count() == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")
    assert launch.visible_gpu_count() == 0
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    monkeypatch.delenv("ROCR_VISIBLE_DEVICES")
    assert launch.ranks_wanted("auto", "auto", use_gpu=False) == 1
    assert launch.ranks_wanted("ddp", "4", use_gpu=True) == 4
    assert launch.ranks_wanted("single_device", "auto", use_gpu=True) == 1
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    script = tmp_path / "dies.py"
    script.write_text(_WORKER_DIES)
    import time
    t0 = time.time()
    rc = launch.spawn_ranks(str(script), [], 2, timeout=120)
    assert rc != 0 and time.time() - t0 < 100          # the surviving rank is stopped by the launcher, not waited for
    slow = tmp_path / "slow.py"
    slow.write_text("import time\ntime.sleep(600)\n")
    t0 = time.time()
    assert launch.spawn_ranks(str(slow), [], 2, timeout=8) == 124 and time.time() - t0 < 60


def test_precision_flag_is_checked_not_ignored():
    """reference train.py:40,132 passes --precision to pl.Trainer; here only the fp32 spellings exist and anything else
    must say so."""
    import train
    assert train.check_precision(None) is None
    assert train.check_precision("32") == "32" and train.check_precision("32-true") == "32-true"
    for bad in ("16-mixed", "bf16-mixed", "16", "64", "bf16-true"):
        with pytest.raises(SystemExit) as e:
            train.check_precision(bad)
        assert "fp32 only" in str(e.value)
    r = subprocess.run([sys.executable, os.path.join(PKG, "train.py"), "--config_path",
                        os.path.join(PKG, "configs", "vae", "vae.json"), "--precision", "16-mixed", "--accelerator", "cpu"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "fp32 only" in (r.stderr + r.stdout)


def test_a_signal_to_the_launching_parent_ends_its_ranks(tmp_path):
    """the ranks run in their own session; SIGTERM to the parent (``timeout -k``, a scheduler) must end them
    too.  The parent here is a real child process of the test so that the signal goes to exactly one PID."""
    import signal
    import subprocess
    import time
    slow = tmp_path / "slow_rank.py"
    slow.write_text("import os, time\nopen(os.environ['PIDDIR'] + '/' + str(os.getpid()), 'w').close()\ntime.sleep(600)\n")
    piddir = tmp_path / "pids"
    piddir.mkdir()
    parent = tmp_path / "parent.py"
    parent.write_text(
        "import sys\nsys.path.insert(0, %r)\nfrom lgm_hip import launch\n"
        "sys.exit(launch.spawn_ranks(%r, [], 2, extra_env={'PIDDIR': %r}))\n" % (PKG, str(slow), str(piddir)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    proc = subprocess.Popen([sys.executable, str(parent)], env=env)
    try:
        t0 = time.time()
        while len(os.listdir(piddir)) < 2 and time.time() - t0 < 90:
            time.sleep(0.2)
        pids = [int(x) for x in os.listdir(piddir)]
        assert len(pids) == 2, "the two ranks never started"
        proc.send_signal(signal.SIGTERM)
        rc = proc.wait(timeout=40)
        assert rc == 128 + signal.SIGTERM
        t0 = time.time()
        alive = pids
        while alive and time.time() - t0 < 20:
            alive = [p for p in alive if os.path.exists(f"/proc/{p}") and
                     open(f"/proc/{p}/stat").read().split(")")[-1].split()[0] != "Z"]
            time.sleep(0.2)
        assert not alive, f"ranks {alive} outlived their launcher"
    finally:
        if proc.poll() is None:
            proc.kill()
        for x in os.listdir(piddir):
            try:
                os.kill(int(x), signal.SIGKILL)
            except (ProcessLookupError, ValueError):
                pass


def test_sync_dist_logging_is_shape_safe_gloo_world2(tmp_path):
    """a scalar logged with sync_dist=True under a rank-dependent condition must not leave the ranks in
    mismatched collectives: every rank raises the same error; names registered but not logged on a rank average over
    the ranks that did log them."""
    script = tmp_path / "worker_log.py"
    script.write_text(_WORKER_LOG_MISMATCH)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29549", str(script), PKG],
                       capture_output=True, text=True, timeout=300, env=dict(os.environ, MASTER_ADDR="127.0.0.1"))
    assert r.returncode == 0, r.stderr[-3000:]
    assert "RANK0 raised" in r.stdout and "RANK1 raised" in r.stdout, r.stdout
    assert "RANK0 a=0.5 b=10.0" in r.stdout and "RANK1 a=0.5 b=10.0" in r.stdout, r.stdout


def test_device_errors_are_told_from_argument_errors():
    """the trainer keeps progress (last.ckpt) after a HOST-side rejection (rc < 0: nothing was launched) and
    must not touch the GPU after a launch / runtime failure (rc > 0, torch.AcceleratorError, RCCL)."""
    from lgm_hip._lib import LgmArgumentError, LgmDeviceError, LgmError
    from lgm_hip.lightning import is_device_error
    assert issubclass(LgmArgumentError, LgmError) and issubclass(LgmDeviceError, LgmError)
    assert not is_device_error(LgmArgumentError("lgm_conv_xy failed (rc=-1): bad pitch"))
    assert is_device_error(LgmDeviceError("lgm_conv_xy failed (rc=719): hipErrorLaunchFailure"))
    assert is_device_error(RuntimeError("HIP error: an illegal memory access was encountered"))
    assert is_device_error(RuntimeError("NCCL error in: ... unhandled system error"))
    assert not is_device_error(ValueError("img_size mismatch")) and not is_device_error(KeyboardInterrupt())
    assert not is_device_error(RuntimeError("shape '[2, 3]' is invalid for input of size 5"))


def test_prescale_follows_the_optimizers_not_a_sticky_flag():
    """1/world is folded into the fused optimizers' kernels (grad_scale); the gradient exchange divides unless
    the optimizers the module holds NOW do that - fresh optimizers on a module that once trained prescaled get averages."""
    from lgm_hip.lightning import MiniLightningModule, _CountingOptimizer, _prescaled

    class Opt:
        def __init__(self, gs):
            self.grad_scale = gs

        def step(self):
            pass

    m = MiniLightningModule()
    assert not _prescaled(m, 2)                                  # no optimizers: divide
    m._optimizers = [_CountingOptimizer(Opt(0.5), m)]
    m._grads_prescaled = False
    assert _prescaled(m, 2) and not _prescaled(m, 4)             # decided by grad_scale == 1 / world
    m._optimizers = [_CountingOptimizer(Opt(0.5), m), _CountingOptimizer(Opt(1.0), m)]
    m._grads_prescaled = True                                    # the old sticky flag is ignored
    assert not _prescaled(m, 2)
    m._optimizers = [_CountingOptimizer(torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=0.1), m)]
    assert not _prescaled(m, 2)                                  # an optimizer without grad_scale


def test_workspaces_of_live_flat_buffers_survive_cleanup():
    """ops.forget_dead_flats() drops the slab workspaces / reducer tables of flat buffers that are gone and
    keeps those of live ones (a captured graph of a live model has their addresses baked in)."""
    import gc
    from lgm_hip import ops
    from lgm_hip.flat import FlatParams
    from lgm_hip.nn import Conv2d, param_kind

    def flat():
        conv = Conv2d(4, 8, 3, padding=1)
        return conv, FlatParams([(n, p, param_kind(n, p)) for n, p in conv.named_parameters()], "cpu")
    c1, f1 = flat()
    c2, f2 = flat()
    k1, k2 = (f1.grad.data_ptr(), 64), (f2.grad.data_ptr() + 16, 64)
    ops._WGRAD_WS[k1] = torch.zeros(16)
    ops._WGRAD_WS[k2] = torch.zeros(16)
    t1 = ((123, 1, f1.grad.data_ptr(), 4, 0, 0, 2, 0),)
    t2 = ((456, 1, f2.grad.data_ptr(), 4, 0, 0, 2, 0),)
    ops._WGRAD_TABLES[t1] = (torch.zeros(1), 1)
    ops._WGRAD_TABLES[t2] = (torch.zeros(1), 1)
    del c2, f2
    gc.collect()
    ops.forget_dead_flats()
    assert k1 in ops._WGRAD_WS and t1 in ops._WGRAD_TABLES          # the live model keeps its workspaces
    assert k2 not in ops._WGRAD_WS and t2 not in ops._WGRAD_TABLES  # the dead one's are gone
    del ops._WGRAD_WS[k1], ops._WGRAD_TABLES[t1]


def test_counter_summary_counts_steady_state_steps_only(tmp_path):
    """tools/pmc_kernels.py (the source of bench.py's `roofline.traffic` and of the per-step HBM bytes): only the dispatches
    between the first and the last `adam_kernel` count - model construction and the first step's one-time fills must not be
    averaged into the step - FETCH_SIZE is doubled (gfx950) and both counters are KB."""
    import csv
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def write(d, counter, rows):
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "x_counter_collection.csv"), "w", newline="") as fh:
            w = csv.writer(fh)
            w.writerow(["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"])
            for did, name, val in rows:
                w.writerow([did, name, counter, val])

    def run(conv_kb, fill_kb):
        rows, did = [], 1
        for _ in range(50):                                   # set-up: parameter initialisation, fills
            rows.append((did, "void at::native::fill_kernel(float*)", fill_kb)); did += 1
        for step in range(4):                                 # a step: two convolution launches, then Adam
            for _ in range(2):
                rows.append((did, "void lgmwino4::wino4_conv_kernel<0, false, 0, false>(lgmwino4::Args)", conv_kb)); did += 1
            rows.append((did, "(anonymous namespace)::adam_kernel(float*, float const*)", 100.0)); did += 1
        rows.append((did, "void at::native::fill_kernel(float*)", fill_kb))        # the tail of the run
        return rows

    f, w = str(tmp_path / "fetch"), str(tmp_path / "write")
    write(f, "FETCH_SIZE", run(1000.0, 5000.0))
    write(w, "WRITE_SIZE", run(400.0, 7000.0))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "pmc_kernels.py"), f, w], capture_output=True, text=True,
                         check=True).stdout
    d = json.loads(out)
    assert d["step_total"]["steps_profiled"] == 3            # four Adam launches bound three whole steps
    k = d["kernels"]["lgmwino4::wino4_conv_kernel<0, false, 0, false>"]
    assert k["launches_per_step"] == 2.0
    assert k["fetch_bytes_per_launch"] == 1000 * 2048 and k["write_bytes_per_launch"] == 400 * 1024
    assert not any("fill_kernel" in n for n in d["kernels"])
    assert d["step_total"]["fetch_bytes"] == (2 * 1000 + 100) * 2048 and d["step_total"]["write_bytes"] == (2 * 400 + 100) * 1024
