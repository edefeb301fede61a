"""TEST INFRASTRUCTURE ONLY -- float64 closed-form oracle for the Lie-group kernels.

The reference's dependency (princeton-vl/lietorch 0.2) is absent from /root/reference (empty submodule) and the
reference holds no test or golden vector at this boundary => PARITY UNPINNED against lietorch itself.  The oracle is
the matrix exponential/logarithm of the 4x4 twist (scipy.linalg.expm/logm, fp64) with the conventions visible at the
reference call sites: data [t, q_xyzw(, s)], tangent [tau, phi(, sigma)], retr(a) = exp(a) * X."""
import numpy as np
from scipy.linalg import expm, logm
from scipy.spatial.transform import Rotation


def hat(phi):
    x, y, z = phi
    return np.array([[0, -z, y], [z, 0, -x], [-y, x, 0.0]])


def twist_matrix(group, a):
    a = np.asarray(a, np.float64)
    M = np.zeros((4, 4))
    if group == 0:
        M[:3, :3] = hat(a[:3])
    else:
        M[:3, :3] = hat(a[3:6]) + (a[6] if group == 2 else 0.0) * np.eye(3)
        M[:3, 3] = a[:3]
    return M


def exp_matrix(group, a):
    return expm(twist_matrix(group, a))


def data_to_matrix(group, d):
    d = np.asarray(d, np.float64)
    M = np.eye(4)
    if group == 0:
        M[:3, :3] = Rotation.from_quat(d[:4]).as_matrix()
    else:
        s = d[7] if group == 2 else 1.0
        M[:3, :3] = s * Rotation.from_quat(d[3:7]).as_matrix()
        M[:3, 3] = d[:3]
    return M


def matrix_to_data(group, M):
    M = np.asarray(M, np.float64)
    if group == 0:
        return Rotation.from_matrix(M[:3, :3]).as_quat()
    s = np.cbrt(np.linalg.det(M[:3, :3])) if group == 2 else 1.0
    q = Rotation.from_matrix(M[:3, :3] / s).as_quat()
    d = np.concatenate([M[:3, 3], q])
    return np.concatenate([d, [s]]) if group == 2 else d


def log_tangent(group, M):
    L = np.real(logm(M))
    phi = np.array([L[2, 1], L[0, 2], L[1, 0]])
    if group == 0:
        return phi
    tau = L[:3, 3]
    if group == 1:
        return np.concatenate([tau, phi])
    return np.concatenate([tau, phi, [np.trace(L[:3, :3]) / 3.0]])


def adjoint_matrix(group, M, eps=1e-6):
    """numerical Ad: X exp(a) X^-1 = exp(Ad a)  =>  column j = d/da_j log(X exp(a) X^-1) at 0 (central differences)"""
    n = {0: 3, 1: 6, 2: 7}[group]
    Ad = np.zeros((n, n))
    Mi = np.linalg.inv(M)
    for j in range(n):
        e = np.zeros(n)
        e[j] = eps
        p = log_tangent(group, M @ exp_matrix(group, e) @ Mi)
        m = log_tangent(group, M @ exp_matrix(group, -e) @ Mi)
        Ad[:, j] = (p - m) / (2 * eps)
    return Ad
