"""Host-side quaternion helpers (numpy), the subset of quadjax/dynamics/geom.py the MPC plumbing uses.

Quaternions are (x, y, z, w).  qtoQ is the closed form of geom.py:68-77 (H^T T L T L H equals the
standard rotation matrix of a unit quaternion).
"""
import numpy as np


def hat(v):
    """geom.py:35-39."""
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]], dtype=np.asarray(v).dtype)


def qtoQ(q):
    """geom.py:68-77 (closed form)."""
    x, y, z, w = q
    return np.array([
        [w * w + x * x - y * y - z * z, 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), w * w - x * x + y * y - z * z, 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), w * w - x * x - y * y + z * z],
    ], dtype=np.asarray(q).dtype)


def Qtoq(Q):
    """geom.py:79-87."""
    tr = 1 + Q[0, 0] + Q[1, 1] + Q[2, 2]
    s = np.sqrt(tr)
    return np.array([0.5 / s * (Q[2, 1] - Q[1, 2]), 0.5 / s * (Q[0, 2] - Q[2, 0]), 0.5 / s * (Q[1, 0] - Q[0, 1]),
                     0.5 * s], dtype=Q.dtype)


def axisangletoR(axis, angle):
    """geom.py:106-112."""
    axis = axis / np.linalg.norm(axis)
    K = hat(axis)
    return np.eye(3, dtype=axis.dtype) + np.sin(angle) * K + (1 - np.cos(angle)) * (K @ K)


def vee(R):
    """geom.py:114-120."""
    return np.array([R[2, 1], R[0, 2], R[1, 0]], dtype=R.dtype)
