"""/root/reference/utils/general_utils.py — the helpers the training / evaluation scripts and the model use."""
import random
import sys
from datetime import datetime

import numpy as np
import torch

from gaussianavatar_amd.dataset import load_masks, to_cuda, uv_index_map  # noqa: F401
from gaussianavatar_amd.losses import adjust_loss_weights  # noqa: F401


def worker_init_fn(worker_id):
    """utils/general_utils.py:9-11: loader workers get distinct numpy seeds."""
    np.random.seed(np.random.get_state()[1][0] + worker_id)


def getIdxMap_torch(img, offset=False):
    """utils/general_utils.py:165-176 for a [C,H,W] image: (row, col) of every texel, normalised."""
    _, H, W = img.shape
    r, c = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    idx = torch.stack([r.reshape(-1), c.reshape(-1)], dim=1).float()
    return (idx + 0.5) / H if offset else idx / (H - 1)


def inverse_sigmoid(x):
    return torch.log(x / (1 - x))


class _Stamped:
    def __init__(self, stream, silent):
        self.stream, self.silent = stream, silent

    def write(self, x):
        if self.silent:
            return
        if x.endswith("\n"):
            x = x.replace("\n", " [{}]\n".format(datetime.now().strftime("%d/%m %H:%M:%S")))
        self.stream.write(x)

    def flush(self):
        self.stream.flush()


def safe_state(silent):
    """utils/general_utils.py:108-129: time-stamped (or silenced) stdout, seeds 0, device 0."""
    sys.stdout = _Stamped(sys.stdout, silent)
    random.seed(0)
    np.random.seed(0)
    torch.manual_seed(0)
    torch.cuda.set_device(torch.device("cuda:0"))
