"""Stand-in for the `lpips` package (train.py:27,84): a perceptual term that is identically zero
(but differentiable), so the loop's control flow past `lpips_start_iter` still runs."""
import torch


class LPIPS(torch.nn.Module):
    def __init__(self, net="alex", **_):
        super().__init__()
        self.net = net

    def forward(self, a, b):
        return ((a - b) * 0.0).mean(dim=(1, 2, 3), keepdim=True)
