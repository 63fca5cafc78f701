"""/root/reference/scene/dataset_mono.py:98-672 -> gaussianavatar_amd.dataset (same class names, item keys)."""
from gaussianavatar_amd.dataset import (  # noqa: F401
    MonoDataset_novel_pose,
    MonoDataset_novel_view,
    MonoDataset_test,
    MonoDataset_train,
    rotate_camera_by_frame_idx,
)
