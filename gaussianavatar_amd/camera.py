"""Camera matrix conventions of the reference, restated (numpy, no torch dependency).

These are part of the drop-in boundary: the rasterizer consumes `world_view_transform`,
`full_proj_transform` and `camera_center` exactly as the reference's datasets build them
(/root/reference/scene/dataset_mono.py:204-255 via /root/reference/utils/graphics_utils.py:28-72,99-100):

    world_view_transform = W2C^T            (row-vector convention: p_view = [p,1] @ wvt)
    full_proj_transform  = wvt @ P^T
    camera_center        = inverse(wvt)[3,:3]
"""
from __future__ import annotations

import math

import numpy as np


def focal2fov(focal: float, pixels: float) -> float:
    """utils/graphics_utils.py:99-100"""
    return 2.0 * math.atan(pixels / (2.0 * focal))


def fov2focal(fov: float, pixels: float) -> float:
    return pixels / (2.0 * math.tan(fov / 2.0))


def world_to_view(R: np.ndarray, t: np.ndarray, translate=(0.0, 0.0, 0.0), scale: float = 1.0) -> np.ndarray:
    """getWorld2View2 (utils/graphics_utils.py:28-39). `R` is the camera-to-world rotation
    (the transpose of the extrinsic's rotation block), `t` the extrinsic translation."""
    Rt = np.eye(4, dtype=np.float64)
    Rt[:3, :3] = np.asarray(R, np.float64).T
    Rt[:3, 3] = np.asarray(t, np.float64)
    C2W = np.linalg.inv(Rt)
    C2W[:3, 3] = (C2W[:3, 3] + np.asarray(translate, np.float64)) * scale
    return np.linalg.inv(C2W).astype(np.float32)


def projection_matrix(znear: float, zfar: float, fovx: float, fovy: float, K=None, h=None, w=None) -> np.ndarray:
    """getProjectionMatrix (utils/graphics_utils.py:41-72): pinhole with the principal point
    taken from K when given. Returned in column-vector form (4x4 float32)."""
    if K is None:
        top = math.tan(fovy / 2) * znear
        right = math.tan(fovx / 2) * znear
        bottom, left = -top, -right
    else:
        near_fx = znear / float(K[0][0])
        near_fy = znear / float(K[1][1])
        left = -(w - float(K[0][2])) * near_fx
        right = float(K[0][2]) * near_fx
        bottom = (float(K[1][2]) - h) * near_fy
        top = float(K[1][2]) * near_fy
    P = np.zeros((4, 4), dtype=np.float32)
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def make_camera(intrinsic: np.ndarray, extrinsic: np.ndarray, width: int, height: int,
                znear: float = 0.01, zfar: float = 100.0) -> dict:
    """Everything a dataset item carries for one view (scene/dataset_mono.py:204-255),
    as float32 numpy arrays / python scalars."""
    K = np.asarray(intrinsic, np.float64)
    E = np.asarray(extrinsic, np.float64)
    fovx = focal2fov(K[0, 0], width)
    fovy = focal2fov(K[1, 1], height)
    R = E[:3, :3].T
    T = E[:3, 3]
    wvt = world_to_view(R, T).T                      # [4,4] transposed (row-vector form)
    P = projection_matrix(znear, zfar, fovx, fovy, K, height, width).T
    full = (wvt.astype(np.float32) @ P.astype(np.float32)).astype(np.float32)
    center = np.linalg.inv(wvt.astype(np.float64))[3, :3].astype(np.float32)
    return dict(FovX=float(fovx), FovY=float(fovy), width=int(width), height=int(height),
                world_view_transform=np.ascontiguousarray(wvt, dtype=np.float32),
                full_proj_transform=np.ascontiguousarray(full, dtype=np.float32),
                camera_center=center, intrinsic=K.astype(np.float32), extrinsic=E.astype(np.float32))


# The single camera shipped with the reference (assets/test_pose/cam_parms.npz):
# fx = fy = 1100, cx = cy = 512 for a 1024x1024 image.
TEST_POSE_INTRINSIC = np.array([[1100.0, 0.0, 512.0], [0.0, 1100.0, 512.0], [0.0, 0.0, 1.0]])
TEST_POSE_EXTRINSIC = np.array([
    [0.99970485, 0.0, 0.02429441, -0.06073601],
    [-0.00589733, -0.97009033, 0.24267256, -0.3156543],
    [0.02356777, -0.24274421, -0.96980401, 2.49733328],
    [0.0, 0.0, 0.0, 1.0]])


def test_pose_camera(size: int = 1024) -> dict:
    """The reference's novel-pose camera, intrinsics scaled to a size x size image."""
    s = size / 1024.0
    K = TEST_POSE_INTRINSIC.copy()
    K[:2] *= s
    return make_camera(K, TEST_POSE_EXTRINSIC, size, size)
