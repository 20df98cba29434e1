"""cv2.findHomography / cv2.perspectiveTransform restated in numpy - TEST INFRASTRUCTURE ONLY (never imported by the product).

The reference's 6x6-grid localisation (baselines/proj_utils.py:37-48, :57-58) calls these two OpenCV functions; cv2 is not
installed here.  With exactly four correspondences and method 0, findHomography returns THE homography through them
(direct linear transform, normalised so that H[2,2] = 1); perspectiveTransform is the projective map x' = H [x y 1]^T
divided by its third component.  oracle/gen_golden.py installs these as the `cv2` stub under which it runs the
reference's own proj_utils module, and tests/golden/grid_classes.npz holds what that run produced."""
import numpy as np


def find_homography_dlt(src: np.ndarray, dst: np.ndarray):
    """4+ point DLT (least squares for more than four): -> (H [3,3] with H[2,2] = 1, status)"""
    src, dst = np.asarray(src, dtype=np.float64).reshape(-1, 2), np.asarray(dst, dtype=np.float64).reshape(-1, 2)
    rows = []
    for (x, y), (u, v) in zip(src, dst):
        rows.append([-x, -y, -1, 0, 0, 0, u * x, u * y, u])
        rows.append([0, 0, 0, -x, -y, -1, v * x, v * y, v])
    _, _, vt = np.linalg.svd(np.asarray(rows))
    H = vt[-1].reshape(3, 3)
    return H / H[2, 2], np.ones((len(src), 1), dtype=np.uint8)


def perspective_transform(pts: np.ndarray, H: np.ndarray) -> np.ndarray:
    """pts [N,1,2] -> [N,1,2]"""
    p = np.asarray(pts, dtype=np.float64).reshape(-1, 2)
    q = (H @ np.concatenate([p, np.ones((len(p), 1))], axis=1).T).T
    return (q[:, :2] / q[:, 2:3]).reshape(-1, 1, 2)
