"""Config + model registry — same contract as the reference's utils/loader.py:
``load_config(path) -> dict`` (JSON with an img_size / img_channels cross-check between the model
and dataset sections) and ``load_model(cfg["model"])`` (import ``models.generative.<family>.<name
lower>`` and instantiate ``<name>(**args)``)."""
import json
from importlib import import_module
from typing import Dict

GENERATIVE_MODELS = ["autoencoder", "autoregressive", "diffusion", "flow", "gan", "vae"]


def load_model(model_config: Dict):
    name = model_config["name"]
    errors = []
    for family in GENERATIVE_MODELS:
        try:
            module = import_module(f"models.generative.{family}.{name.lower()}")
        except ImportError as e:  # family does not provide this model: try the next one
            errors.append(f"{family}: {e}")
            continue
        return getattr(module, name)(**model_config["args"])
    raise ValueError(f"Failed to import {name}. Errors encountered: \n " + "\n".join(errors))


def load_config(config_path: str) -> Dict:
    try:
        with open(config_path, "r") as f:
            config = json.load(f)
    except FileNotFoundError:
        raise FileNotFoundError(f"Configuration file not found at '{config_path}'.")
    except json.JSONDecodeError:
        raise ValueError(f"The file at '{config_path}' is not a valid JSON.")
    margs = config.get("model", {}).get("args", {})
    dset = config.get("dataset", {})
    for key in ("img_channels", "img_size"):
        if margs.get(key) != dset.get(key):
            raise ValueError(f"Mismatch in '{key}' between model and dataset configurations.")
    return config
