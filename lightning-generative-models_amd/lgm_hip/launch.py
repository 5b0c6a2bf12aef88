"""One process per GPU, started by the entry point itself.

The reference's ``train.py:38`` defaults ``--strategy`` to ``configure_strategy()``
(``utils/lightning_utils.py:37-43``): ``DDPStrategy`` whenever ``torch.cuda.device_count() > 1``, and Lightning then
starts one process per GPU from a plain ``python train.py``.  This module is that behaviour for ``train.py`` and
``bench.py``: when no launcher set ``WORLD_SIZE`` and more than one rank is wanted, the PARENT — before it has made any
GPU call — starts ``python -m torch.distributed.run --nproc-per-node N <script> <argv>`` as a child process, relays
its output and exits with its return code.  Never ``exec`` (the pool forbids replacing a process that may have touched
the GPU), never a pattern kill: the one child is stopped by its PID / process group.

Nothing here imports the HIP library or initialises a device: ``visible_gpu_count`` reads the environment and sysfs.
"""
import os
import signal
import socket
import subprocess
import sys
import threading
from typing import List, Optional, Sequence

_VISIBLE_VARS = ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES")


def _kfd_gpu_nodes() -> Optional[int]:
    """GPU agents the kernel driver exposes (topology nodes with SIMDs; the CPU nodes have simd_count 0)."""
    root = "/sys/class/kfd/kfd/topology/nodes"
    try:
        nodes = os.listdir(root)
    except OSError:
        return None
    n = 0
    for d in nodes:
        try:
            with open(os.path.join(root, d, "properties")) as f:
                for ln in f:
                    k, _, v = ln.partition(" ")
                    if k == "simd_count":
                        n += int(v) > 0
                        break
        except (OSError, ValueError):
            continue
    return n


def visible_gpu_count() -> int:
    """Devices a rank could bind to, WITHOUT initialising HIP: the visibility variables when set (the narrowest one wins),
    else the driver's topology, else ``torch.cuda.device_count()`` (which does not create a context on this image)."""
    counts = []
    for var in _VISIBLE_VARS:
        v = os.environ.get(var)
        if v is None:
            continue
        ids = [s for s in v.split(",") if s.strip() != ""]
        # an index of -1 (or a first invalid id) hides every device behind it
        keep = 0
        for s in ids:
            if s.strip() == "-1":
                break
            keep += 1
        counts.append(keep)
    if counts:
        return min(counts)
    n = _kfd_gpu_nodes()
    if n is not None:
        return n
    import torch
    return int(torch.cuda.device_count())


def launched() -> bool:
    """True when a launcher (torch.distributed.run, or this module's parent) already set the rank environment."""
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])


def spawn_ranks(script: str, argv: Sequence[str], nproc: int, *, extra_env: Optional[dict] = None,
                timeout: Optional[float] = None) -> int:
    """Run ``script argv`` as ``nproc`` ranks under torch.distributed.run (rendezvous on 127.0.0.1) and return the
    launcher's exit code (non-zero as soon as any rank fails: the elastic agent stops the others).  ``timeout``
    (seconds, default ``LGM_LAUNCH_TIMEOUT`` or none) bounds the whole child: on expiry its process group is ended
    and 124 is returned.  stdout / stderr are inherited, so rank 0's JSON line reaches the caller's stdout as is."""
    assert nproc >= 1
    assert not launched(), "spawn_ranks() from inside a launched rank"
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // nproc)))
    if extra_env:
        env.update({k: str(v) for k, v in extra_env.items()})
    cmd: List[str] = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
                      "--master-addr", "127.0.0.1", "--master-port", str(free_port()), script, *argv]
    if timeout is None and os.environ.get("LGM_LAUNCH_TIMEOUT"):
        timeout = float(os.environ["LGM_LAUNCH_TIMEOUT"])
    print(f"[launch] starting {nproc} ranks: {' '.join(cmd[1:])}", file=sys.stderr, flush=True)
    # The ranks live in their own session: a SIGTERM / SIGHUP / SIGINT that ends THIS process (``timeout -k``, a
    # scheduler, a closed terminal) would otherwise leave them running and holding the GPUs.  The handlers are installed
    # BEFORE the child is started (a signal that arrived between Popen and signal.signal() killed the parent by
    # default action and orphaned the rank group); while the child runs the signals are turned into an exception, so the
    # ``finally``-like paths below end exactly the group started here - or nothing, when the child does not exist yet.
    caught: List[int] = []
    proc: Optional[subprocess.Popen] = None

    def _on_signal(signum, _frame):
        caught.append(signum)
        raise _Interrupted(signum)

    old = {}
    if threading.current_thread() is threading.main_thread():
        for sig in (signal.SIGTERM, signal.SIGHUP, signal.SIGINT):
            try:
                old[sig] = signal.signal(sig, _on_signal)
            except (ValueError, OSError):
                pass
    try:
        proc = subprocess.Popen(cmd, env=env, start_new_session=True)
        return int(proc.wait(timeout=timeout))
    except subprocess.TimeoutExpired:
        print(f"[launch] {nproc}-rank run exceeded {timeout:.0f} s: ending its process group", file=sys.stderr, flush=True)
        _end_group(proc)
        return 124
    except _Interrupted as e:
        print(f"[launch] signal {e.signum}: ending the {nproc}-rank process group", file=sys.stderr, flush=True)
        for sig in old:                      # a second signal while the group is being ended must not re-enter
            signal.signal(sig, signal.SIG_IGN)
        if proc is not None:                 # the signal may have arrived before the child existed
            _end_group(proc, first=e.signum if e.signum != signal.SIGHUP else signal.SIGTERM)
        return 128 + int(e.signum)
    except BaseException:
        if proc is not None:
            _end_group(proc)
        raise
    finally:
        for sig, h in old.items():
            signal.signal(sig, h)


class _Interrupted(Exception):
    def __init__(self, signum: int):
        super().__init__(f"signal {signum}")
        self.signum = signum


def _end_group(proc: subprocess.Popen, first: int = signal.SIGTERM) -> None:
    """``first`` (SIGTERM unless a caught signal is being forwarded), then SIGKILL, to exactly the process group this
    module started."""
    for sig, wait in ((first, 10.0), (signal.SIGKILL, 5.0)):
        try:
            os.killpg(proc.pid, sig)
        except (ProcessLookupError, PermissionError):
            return
        try:
            proc.wait(timeout=wait)
            return
        except subprocess.TimeoutExpired:
            continue


def ranks_wanted(strategy: str, devices: str, use_gpu: bool) -> int:
    """How many ranks ``train.py`` should run as, following the reference's defaults: ``--strategy auto`` (its
    ``configure_strategy()``) or ``ddp`` on a node that shows more than one GPU means one rank per GPU; an explicit
    ``--devices N`` overrides the count (Lightning's ``Trainer(devices=N)``); CPU runs are one rank unless asked."""
    if devices not in (None, "auto", "-1"):
        n = int(devices)
        assert n >= 1, "--devices must be >= 1"
        return n
    if strategy not in ("auto", "ddp", "ddp_find_unused_parameters_true"):
        return 1
    if not use_gpu:
        return 1
    return max(1, visible_gpu_count())
