"""/root/reference/model/network.py:8 POP_no_unet -> gaussianavatar_amd.network (state-dict compatible)."""
from gaussianavatar_amd.network import POP_no_unet  # noqa: F401
