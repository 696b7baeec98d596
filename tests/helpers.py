import numpy as np


def pose_error(T_a, T_b):
    """(translation distance [m], rotation angle [rad]) between two 4x4 transforms."""
    dt = float(np.linalg.norm(T_a[:3, 3] - T_b[:3, 3]))
    R = T_a[:3, :3].T @ T_b[:3, :3]
    # |R - I|_F = 2 sqrt(2) |sin(theta / 2)|: well conditioned near theta = 0 (arccos is not)
    ang = float(2.0 * np.arcsin(min(1.0, np.linalg.norm(R - np.eye(3)) / (2.0 * np.sqrt(2.0)))))
    return dt, ang


TOL_T = 1e-4    # metres   (north_star: <= 1e-4 m vs the reference CPU path)
TOL_R = 1e-4    # radians  (north_star: <= 1e-4 rad)


def svd_stats_numpy(p, q, d2):
    """The 17 Umeyama statistics in the public 32-slot layout (float64)."""
    st = np.zeros(32)
    p = p.astype(np.float64)
    q = q.astype(np.float64)
    st[0] = len(p)
    st[1:4] = p.sum(0)
    st[4:7] = q.sum(0)
    st[7:16] = (q.T @ p).reshape(-1)
    st[16] = d2.astype(np.float64).sum()
    return st
