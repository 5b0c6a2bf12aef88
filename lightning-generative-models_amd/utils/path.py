from pathlib import Path

PROJECT_ROOT = Path(__file__).parents[1]
DATASET_PATH = PROJECT_ROOT / "data" / "dataset"
EXPERIMENT_DIR = PROJECT_ROOT / "experiments"
