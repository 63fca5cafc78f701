"""Stand-in for torchvision: only `utils.save_image` (train.py:112-113, eval.py:70-71)."""
from . import utils  # noqa: F401
