"""/root/reference/utils/graphics_utils.py -> gaussianavatar_amd.camera (same conventions)."""
import numpy as np
import torch

from gaussianavatar_amd.camera import focal2fov, fov2focal, projection_matrix, world_to_view  # noqa: F401


def geom_transform_points(points, transf_matrix):
    """utils/graphics_utils.py:12-19: row-vector projective transform with the 1e-7 guard on w."""
    hom = torch.cat([points, torch.ones_like(points[:, :1])], dim=1)
    out = hom @ transf_matrix
    return out[:, :3] / (out[:, 3:] + 0.0000001)


def getWorld2View2(R, t, translate=np.array([.0, .0, .0]), scale=1.0):
    return world_to_view(R, t, translate, scale)


def getProjectionMatrix(znear, zfar, fovX, fovY, K, h, w):
    return torch.from_numpy(projection_matrix(znear, zfar, fovX, fovY, K, h, w))
