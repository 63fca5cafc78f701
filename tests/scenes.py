"""Seeded synthetic scenes shared by the oracle and GPU parity tests (numpy only)."""
from __future__ import annotations

import math

import numpy as np

from gaussianavatar_amd.camera import make_camera, TEST_POSE_EXTRINSIC


def camera(W: int, H: int, focal_scale: float = 1.0):
    """The reference's fixture camera (assets/test_pose/cam_parms.npz) rescaled to W x H."""
    s = W / 1024.0
    K = np.array([[1100.0 * s * focal_scale, 0, W / 2.0], [0, 1100.0 * s * focal_scale, H / 2.0], [0, 0, 1.0]])
    return make_camera(K, TEST_POSE_EXTRINSIC, W, H)


def random_scene(P: int, W: int, H: int, seed: int = 0, kind: str = "general", spread: float = 0.6,
                 scale_med: float = 0.03):
    """`general`: anisotropic scales, random un-normalised quaternions, opacity U(0.05,1).
    `avatar`:  isotropic scales, identity rotation, opacity 1 (the reference's configuration,
               /root/reference/model/avatar_model.py:79-83,323-324)."""
    rng = np.random.default_rng(seed)
    cam = camera(W, H)
    # points around the world origin + (0, 0.3, 0): the camera centre is (0, 0.3, 2.5)
    means = rng.normal(0, spread, (P, 3)).astype(np.float32) * np.array([0.5, 1.0, 0.3], np.float32)
    means[:, 1] += 0.3
    colors = rng.uniform(0, 1, (P, 3)).astype(np.float32)
    if kind == "avatar":
        s = np.exp(rng.normal(math.log(scale_med), 0.4, (P, 1))).astype(np.float32)
        scales = np.repeat(s, 3, 1)
        rots = np.zeros((P, 4), np.float32)
        rots[:, 0] = 1
        opac = np.ones(P, np.float32)
    else:
        scales = np.exp(rng.normal(math.log(scale_med), 0.7, (P, 3))).astype(np.float32)
        rots = rng.normal(0, 1, (P, 4)).astype(np.float32)
        rots /= np.linalg.norm(rots, axis=1, keepdims=True) * rng.uniform(0.8, 1.25, (P, 1)).astype(np.float32)
        opac = rng.uniform(0.05, 1.0, P).astype(np.float32)
    return dict(P=P, W=W, H=H, means3D=means, colors=colors, opacities=opac, scales=scales,
                rotations=rots, bg=np.array([1.0, 1.0, 1.0], np.float32),
                viewmatrix=cam["world_view_transform"], projmatrix=cam["full_proj_transform"],
                campos=cam["camera_center"],
                tanfovx=math.tan(cam["FovX"] * 0.5), tanfovy=math.tan(cam["FovY"] * 0.5))


def cam_kwargs(sc):
    return dict(viewmatrix=sc["viewmatrix"], projmatrix=sc["projmatrix"], bg=sc["bg"], W=sc["W"],
                H=sc["H"], tanfovx=sc["tanfovx"], tanfovy=sc["tanfovy"])


def dataset_fixture(tmp, smpl_type):
    """A small dataset in the reference's on-disk layout (synthetic.write_dataset): 700 points on
    a 32x32 UV map, 4 frames of 64x48 with random images and masks. oracle/make_golden.py reads it
    with the reference's MonoDataset_* classes, tests/test_dataset.py with ours."""
    import os

    import torch
    from gaussianavatar_amd.synthetic import make_assets, make_frames, write_dataset
    assets = make_assets(700, 32, smpl_type)
    frames = make_frames(assets, 4, 64, 48)
    g = torch.Generator().manual_seed(11)
    images = torch.rand(4, 3, 48, 64, generator=g)
    masks = torch.rand(4, 48, 64, generator=g) > 0.4
    paths = write_dataset(os.path.join(tmp, "data"), os.path.join(tmp, "proj"), assets, frames,
                          images=images, masks=masks, inp_posmap_size=16)
    return assets, frames, paths
