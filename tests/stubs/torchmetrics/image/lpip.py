import torch


class LearnedPerceptualImagePatchSimilarity(torch.nn.Module):
    def __init__(self, net_type="alex"):
        super().__init__()

    def forward(self, a, b):
        return torch.zeros((), device=a.device)
