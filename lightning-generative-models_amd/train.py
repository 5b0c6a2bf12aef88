"""Training entry point with the reference's CLI (train.py:31-66):

    python train.py --config_path configs/diffusion/ddpm.json [--max_steps N] [--max_epochs N]
                    [--accumulate_grad_batches K] [--ckpt_path last.ckpt] [--strategy ddp|auto]

One process per GPU.  Like the reference (``--strategy`` defaults to DDP when the node shows more than one
GPU, and Lightning starts the ranks), a plain ``python train.py ...`` on a multi-GPU node becomes the parent of one rank
per visible GPU (``--devices N`` overrides the count; ``maybe_spawn`` below, before any GPU call); under an external
``python -m torch.distributed.run --nproc-per-node N train.py ...`` it is one of the ranks.  Every
rank binds to its LOCAL_RANK device, joins a torch.distributed group (backend "nccl" = RCCL over
xGMI on ROCm, "gloo" on CPU) and averages the flat gradient buffers once per optimizer step.
The loop is ALWAYS the in-repo MiniTrainer (lgm_hip/lightning.py) — also where pytorch_lightning is
installed: the HIP engine's flat gradient buffers, overlapped bucketed all-reduce and graph-replayed step
are driven by it (for DDPM: the same DDPMFastStep bench.py times).  It keeps the reference trainer's
observable behaviour: ``last.ckpt`` (+ the best-``val_loss`` checkpoint) in the experiment directory,
validation at the end of every epoch, ``--max_steps`` / ``--max_epochs`` / ``--accumulate_grad_batches``.
"""
import argparse
import os
import sys
from datetime import datetime
from pathlib import Path
from pprint import pprint

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this stack

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import yaml  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

from data.datamodule import DataModule  # noqa: E402
from lgm_hip.lightning import HAVE_PL, MiniTrainer  # noqa: E402
from utils.loader import load_config, load_model  # noqa: E402
from utils.path import EXPERIMENT_DIR  # noqa: E402
from utils.seed import seed_everything  # noqa: E402

seed_everything(seed=10, workers=True)
EXPERIMENT_TIME = datetime.now().strftime("%Y-%m-%d_%H:%M")


def check_precision(value):
    """``--precision`` is handed to ``pl.Trainer(precision=...)`` by the reference (train.py:40,132); its default ``None``
    is Lightning's "32-true".  This engine computes in fp32 only (exact fp32 MFMA, parity at 1e-4): the fp32 spellings are
    accepted, anything else is an error instead of a silent fp32 run."""
    if value is None or str(value) in ("32", "32-true"):
        return None if value is None else str(value)
    raise SystemExit(f"--precision {value!r}: this engine trains in fp32 only (accepted: omitted, 32, 32-true); "
                     "mixed / half / double precision of the reference's Trainer is not implemented")


def setup_arguments(argv=None, print_args=True, save_args=True):
    p = argparse.ArgumentParser("Train script")
    p.add_argument("--config_path", type=str, required=True, help="Path to configs")
    p.add_argument("--num_workers", type=int, default=0)
    p.add_argument("--check_val_every_n_epoch", type=int, default=5)
    p.add_argument("--max_epochs", type=int, default=-1)
    p.add_argument("--max_steps", type=int, default=-1)
    p.add_argument("--strategy", type=str, default="auto")
    p.add_argument("--accumulate_grad_batches", type=int, default=1)
    p.add_argument("--precision", type=str, default=None)
    p.add_argument("--ckpt_path", type=str, default=None)
    p.add_argument("--project", type=str, default="Lightning generative models")
    p.add_argument("--experiment_name", type=str, default=EXPERIMENT_TIME)
    p.add_argument("--resume", action="store_true")
    p.add_argument("--id", type=str, default=None)
    p.add_argument("--accelerator", type=str, default="auto", help="auto | cpu | gpu")
    p.add_argument("--devices", type=str, default="auto",
                   help="ranks to run as (Lightning's Trainer(devices=...)): auto = every visible GPU, like the reference")
    args = p.parse_args(argv)
    args.precision = check_precision(args.precision)
    args.config = load_config(args.config_path)
    args.experiment_dir = os.path.join(EXPERIMENT_DIR, args.config["model"]["name"], args.experiment_name)
    os.makedirs(args.experiment_dir, exist_ok=True)
    if print_args:
        pprint(vars(args))
    if save_args:
        with open(os.path.join(args.experiment_dir, Path(args.config_path).name), "w") as f:
            yaml.safe_dump({k: (v if isinstance(v, (int, float, str, bool, dict, list, type(None))) else str(v))
                            for k, v in vars(args).items()}, f)
    return args


def init_distributed(use_gpu: bool):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 1
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if use_gpu:
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group("nccl" if use_gpu else "gloo")
    return dist.get_rank(), world


def maybe_spawn(argv):
    """The reference's ``python train.py`` on a multi-GPU node trains on ALL its GPUs (``--strategy`` defaults to
    ``configure_strategy()`` = DDP when ``device_count() > 1``, reference train.py:38, utils/lightning_utils.py:37-43, and
    Lightning starts the ranks).  Same here: with no launcher in the environment, decide from the command line and the
    visible devices - WITHOUT touching the GPU - whether this process is the parent of N ranks; if so start them
    (lgm_hip/launch.py) and return their exit code, else return None and run as the one rank."""
    from lgm_hip import launch
    if launch.launched():
        return None
    pre = argparse.ArgumentParser(add_help=False)
    pre.add_argument("--strategy", type=str, default="auto")
    pre.add_argument("--accelerator", type=str, default="auto")
    pre.add_argument("--devices", type=str, default="auto")
    a, _ = pre.parse_known_args(argv)
    use_gpu = a.accelerator != "cpu" and launch.visible_gpu_count() > 0
    n = launch.ranks_wanted(a.strategy, a.devices, use_gpu)
    if n <= 1:
        return None
    if use_gpu:
        assert n <= launch.visible_gpu_count(), f"--devices {n} but {launch.visible_gpu_count()} GPUs are visible"
    child_argv = list(sys.argv[1:] if argv is None else argv)
    if not any(x == "--experiment_name" or x.startswith("--experiment_name=") for x in child_argv):
        child_argv += ["--experiment_name", EXPERIMENT_TIME]     # one experiment directory for all ranks
    return launch.spawn_ranks(os.path.abspath(__file__), child_argv, n)


def main(argv=None):
    rc = maybe_spawn(argv)
    if rc is not None:
        if rc != 0:
            sys.exit(rc)
        return None
    args = setup_arguments(argv, print_args=int(os.environ.get("RANK", "0")) == 0)
    use_gpu = args.accelerator != "cpu" and torch.cuda.is_available()
    rank, world = init_distributed(use_gpu)
    model = load_model(args.config["model"])
    datamodule = DataModule(**args.config["dataset"], num_workers=args.num_workers, pin_memory=True)
    if use_gpu:
        device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    else:
        device = torch.device("cpu")
    trainer = MiniTrainer(max_epochs=args.max_epochs, max_steps=args.max_steps, default_root_dir=args.experiment_dir,
                          accumulate_grad_batches=args.accumulate_grad_batches, device=device,
                          check_val_every_n_epoch=args.check_val_every_n_epoch)   # reference train.py: Trainer(...)
    trainer.fit(model, datamodule=datamodule, ckpt_path=args.ckpt_path)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return model


if __name__ == "__main__":
    main()
