"""SE(3) helpers used by tests and the golden-vector generator.

TEST INFRASTRUCTURE ONLY (see oracle/README.md): nothing under ``oracle/`` may be
imported by the product package ``dicp_amd``.

The reference's tests build their ground truth with the third-party package
``asrl-pylgmath`` (unpinned, /root/reference/environment.yml:11), which is absent
from this image.  Only two of its functions are used
(/root/reference/tests/test_ICP.py:45-47,65):

* ``Transformation(xi_ab=xi).matrix()``  ->  :func:`vec2tran`
* ``se3op.tran2vec(T)``                  ->  :func:`tran2vec`

Both follow the published convention of Barfoot, *State Estimation for Robotics*,
eq. (7.83)-(7.86): ``xi = [rho; phi]``, ``C = exp(phi^)``, ``r = J(phi) rho`` with
``J`` the left Jacobian of SO(3).  The convention is pinned indirectly: three
independent reference ICP runs reach ``inv(vec2tran([1,1,0,0,0,.1]))`` to < 1e-10
(tests/test_oracle_golden.py).
"""
import numpy as np


def hat(v):
    v = np.asarray(v, dtype=np.float64).reshape(3)
    return np.array([[0.0, -v[2], v[1]],
                     [v[2], 0.0, -v[0]],
                     [-v[1], v[0], 0.0]])


def so3_exp(phi):
    phi = np.asarray(phi, dtype=np.float64).reshape(3)
    th = np.linalg.norm(phi)
    K = hat(phi)
    if th < 1e-12:
        return np.eye(3) + K + 0.5 * K @ K
    return np.eye(3) + (np.sin(th) / th) * K + ((1.0 - np.cos(th)) / th ** 2) * (K @ K)


def so3_left_jacobian(phi):
    phi = np.asarray(phi, dtype=np.float64).reshape(3)
    th = np.linalg.norm(phi)
    K = hat(phi)
    if th < 1e-12:
        return np.eye(3) + 0.5 * K + (K @ K) / 6.0
    return (np.eye(3) + ((1.0 - np.cos(th)) / th ** 2) * K
            + ((th - np.sin(th)) / th ** 3) * (K @ K))


def so3_log(C):
    C = np.asarray(C, dtype=np.float64)
    c = np.clip((np.trace(C) - 1.0) * 0.5, -1.0, 1.0)
    th = np.arccos(c)
    w = np.array([C[2, 1] - C[1, 2], C[0, 2] - C[2, 0], C[1, 0] - C[0, 1]])
    if th < 1e-12:
        return 0.5 * w
    return (th / (2.0 * np.sin(th))) * w


def vec2tran(xi):
    """xi = [rho(3); phi(3)] -> 4x4 transform (pylgmath ``Transformation(xi_ab=xi).matrix()``)."""
    xi = np.asarray(xi, dtype=np.float64).reshape(6)
    T = np.eye(4)
    T[:3, :3] = so3_exp(xi[3:])
    T[:3, 3] = so3_left_jacobian(xi[3:]) @ xi[:3]
    return T


def tran2vec(T):
    """Inverse of :func:`vec2tran`; accepts (4,4) or (N,4,4), returns (6,1) or (N,6,1)."""
    T = np.asarray(T, dtype=np.float64)
    if T.ndim == 3:
        return np.stack([tran2vec(t) for t in T], axis=0)
    phi = so3_log(T[:3, :3])
    rho = np.linalg.solve(so3_left_jacobian(phi), T[:3, 3])
    return np.concatenate([rho, phi]).reshape(6, 1)
