"""/root/reference/utils/loss_utils.py:7-53 -> gaussianavatar_amd.losses (L1 + SSIM share one HIP pass)."""
import torch

from gaussianavatar_amd.losses import l1_loss_w, ssim  # noqa: F401


def l2_loss(network_output, gt):
    """utils/loss_utils.py:10-11"""
    return ((network_output - gt) ** 2).mean()
