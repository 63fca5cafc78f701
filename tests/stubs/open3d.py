"""Stand-in for the three open3d calls of train.py:106-110 (point-cloud dump of the first log step)."""
import numpy as np


class _PointCloud:
    points = None


class geometry:
    PointCloud = _PointCloud


class utility:
    @staticmethod
    def Vector3dVector(a):
        return np.asarray(a, dtype=np.float64).reshape(-1, 3)


class io:
    @staticmethod
    def write_point_cloud(path, pcd):
        pts = np.asarray(pcd.points)
        with open(path, "w") as f:
            f.write("ply\nformat ascii 1.0\nelement vertex %d\nproperty float x\nproperty float y\n"
                    "property float z\nend_header\n" % len(pts))
            np.savetxt(f, pts, fmt="%.6f")
        return True
