"""Image point -> CATER 6x6 floor-grid class (reference baselines/proj_utils.py).

The reference fits a homography between the image plane and the floor plane z = 0.342 with
`cv2.findHomography` on four projected points (proj_utils.py:37-48) and applies it with
`cv2.perspectiveTransform` (:57-58).  Four exact correspondences determine the homography uniquely, and the
floor -> image map is available in closed form from the camera matrix (the columns of CATER_CAM with z folded
in), so here H is simply its inverse - same map, no cv2.  Vectorised over points.
"""
from __future__ import annotations

import numpy as np

# proj_utils.py:12-16 (the fixed CATER camera)
CATER_CAM = np.array([
    (1.4503, 1.6376, 0.0000, -0.0251),
    (-1.0346, 0.9163, 2.5685, 0.0095),
    (-0.6606, 0.5850, -0.4748, 10.5666),
    (-0.6592, 0.5839, -0.4738, 10.7452)], dtype=np.float64)
Z = 0.3421497941017151                       # :40, the plane the objects sit on


def project_3d_point(pts: np.ndarray) -> np.ndarray:
    """:19-33: Nx3 world points -> Nx2 image points in [-1, 1], top-left = (-1, -1)"""
    pts = np.asarray(pts, dtype=np.float64)
    p = (CATER_CAM @ np.hstack((pts, np.ones((pts.shape[0], 1)))).T).T
    return np.stack([p[:, 0] / p[:, -1], p[:, 1] / -p[:, -1]], axis=1)


def _floor_to_image() -> np.ndarray:
    c = CATER_CAM
    return np.array([[c[0, 0], c[0, 1], c[0, 2] * Z + c[0, 3]],
                     [-c[1, 0], -c[1, 1], -(c[1, 2] * Z + c[1, 3])],
                     [c[3, 0], c[3, 1], c[3, 2] * Z + c[3, 3]]])


H = np.linalg.inv(_floor_to_image())
H = H / H[2, 2]                              # cv2.findHomography's normalisation


def get_class_predictions(cx, cy, nrows: int = 3, ncols: int = 3) -> np.ndarray:
    """:50-75 for arrays of points (cx, cy in [-1, 1] as project_3d_point returns them) -> class ids"""
    cx, cy = np.asarray(cx, dtype=np.float64), np.asarray(cy, dtype=np.float64)
    q = H @ np.stack([cx, cy, np.ones_like(cx)])
    x, y = q[0] / q[2], q[1] / q[2]
    x = np.minimum(np.maximum(-3, x), 3 - 0.00001) * (ncols / 3.0)
    y = np.minimum(np.maximum(-3, y), 3 - 0.00001) * (nrows / 3.0)
    x1 = np.floor(x).astype(np.int64) + ncols
    y1 = np.floor(y).astype(np.int64) + nrows
    cls_id = y1 * (2 * ncols) + x1
    assert np.all((cls_id >= 0) & (cls_id < 4 * nrows * ncols))
    return cls_id


def get_class_prediction(cx: float, cy: float, nrows: int = 3, ncols: int = 3) -> int:
    return int(get_class_predictions(np.array([cx]), np.array([cy]), nrows, ncols)[0])
