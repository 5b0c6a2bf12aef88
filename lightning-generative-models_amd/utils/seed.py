"""seed_everything(seed, workers) — reference utils/seed.py (pl.seed_everything + PYTHONHASHSEED)."""
import os
import random

import numpy as np
import torch


def seed_everything(seed: int = 10, workers: bool = True) -> int:
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    os.environ["PL_GLOBAL_SEED"] = str(seed)
    os.environ["PL_SEED_WORKERS"] = str(int(workers))
    return seed
