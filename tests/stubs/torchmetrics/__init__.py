"""Stand-in for the three torchmetrics classes of eval.py:12-13,20-22."""
import torch


class PeakSignalNoiseRatio(torch.nn.Module):
    def __init__(self, data_range=1.0):
        super().__init__()
        self.data_range = data_range

    def forward(self, a, b):
        return 10.0 * torch.log10(self.data_range ** 2 / torch.mean((a - b) ** 2).clamp_min(1e-12))


class StructuralSimilarityIndexMeasure(torch.nn.Module):
    def __init__(self, data_range=1.0):
        super().__init__()

    def forward(self, a, b):
        from gaussianavatar_amd.losses import ssim
        return ssim(a, b)
