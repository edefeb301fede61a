"""GPU: lietorch-compatible SE3/SO3/Sim3 kernels vs the fp64 matrix-exponential oracle (oracle/lie_oracle.py).
Parity vs lietorch itself is UNPINNED (the dependency is absent from the reference tree); conventions come from the
reference call sites.  Tolerances: fp32 kernels vs fp64 oracle, 2e-5 absolute on O(1) quantities."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from cut3r_slam_amd.lietorch import SE3, SO3, Sim3  # noqa: E402
from oracle import lie_oracle as LO  # noqa: E402

DEV = "cuda:0"
GROUPS = [(SO3, 0), (SE3, 1), (Sim3, 2)]


def _tangents(gid, n, seed, scale=1.0):
    g = np.random.default_rng(seed)
    a = g.normal(0, scale, (n, {0: 3, 1: 6, 2: 7}[gid]))
    if gid == 2:
        a[:, 6] *= 0.3
    a[0] = 0.0                       # identity
    a[1] *= 1e-4                     # small-angle branch
    if gid:
        a[2, 3:6] *= 1e-5            # tiny rotation, finite translation
    if gid == 2:
        a[3, 6] = 1e-6               # tiny log-scale branch
    return a


@pytest.mark.parametrize("cls,gid", GROUPS)
def test_exp_matrix_log_roundtrip(cls, gid):
    a = _tangents(gid, 64, gid)
    X = cls.exp(torch.from_numpy(a).float().to(DEV))
    M = X.matrix().cpu().numpy()
    for i in range(len(a)):
        np.testing.assert_allclose(M[i], LO.exp_matrix(gid, a[i]), atol=3e-5, err_msg=f"exp->matrix #{i}")
        np.testing.assert_allclose(LO.data_to_matrix(gid, X.data[i].cpu().numpy()), M[i], atol=3e-5)
    back = X.log().cpu().numpy()
    # rotation angle < pi for these samples => log(exp(a)) == a
    ok = np.linalg.norm(a[:, (0 if gid == 0 else 3):(3 if gid == 0 else 6)], axis=1) < 3.0
    np.testing.assert_allclose(back[ok], a[ok], atol=5e-5)


@pytest.mark.parametrize("cls,gid", GROUPS)
def test_group_axioms_and_actions(cls, gid):
    a, b = _tangents(gid, 32, 10 + gid), _tangents(gid, 32, 20 + gid)
    X, Y = cls.exp(torch.from_numpy(a).float().to(DEV)), cls.exp(torch.from_numpy(b).float().to(DEV))
    MX, MY = X.matrix().cpu().numpy().astype(np.float64), Y.matrix().cpu().numpy().astype(np.float64)
    np.testing.assert_allclose((X * Y).matrix().cpu().numpy(), MX @ MY, atol=1e-4)
    np.testing.assert_allclose((X * X.inv()).matrix().cpu().numpy(), np.tile(np.eye(4), (32, 1, 1)), atol=1e-4)
    g = np.random.default_rng(5)
    p3 = g.normal(0, 1, (32, 5, 3))
    out3 = X[:, None].act(torch.from_numpy(p3).float().to(DEV)).cpu().numpy()
    ref3 = np.einsum("nij,npj->npi", MX[:, :3, :3], p3) + MX[:, None, :3, 3]
    np.testing.assert_allclose(out3, ref3, atol=1e-4)
    p4 = g.normal(0, 1, (32, 5, 4))
    out4 = (X[:, None] * torch.from_numpy(p4).float().to(DEV)).cpu().numpy()
    ref4 = np.einsum("nij,npj->npi", MX, p4)
    np.testing.assert_allclose(out4, ref4, atol=1e-4)
    # retr(a) = exp(a) * X
    R = X.retr(torch.from_numpy(b).float().to(DEV)).matrix().cpu().numpy()
    np.testing.assert_allclose(R, MY @ MX, atol=1e-4)


@pytest.mark.parametrize("cls,gid", GROUPS)
def test_adjoint_and_transpose(cls, gid):
    a = _tangents(gid, 8, 30 + gid, 0.7)[4:]
    X = cls.exp(torch.from_numpy(a).float().to(DEV))
    n = cls.manifold_dim
    eye = torch.eye(n, device=DEV)
    for i in range(len(a)):
        Ad = torch.stack([X[i].adj(eye[j]) for j in range(n)], 1).cpu().numpy()         # columns = Ad e_j
        AdT = torch.stack([X[i].adjT(eye[j]) for j in range(n)], 1).cpu().numpy()
        ref = LO.adjoint_matrix(gid, LO.exp_matrix(gid, a[i]))
        np.testing.assert_allclose(Ad, ref, atol=2e-4)
        np.testing.assert_allclose(AdT, ref.T, atol=2e-4)


@pytest.mark.parametrize("cls,gid", GROUPS)
def test_autograd_exp_matrix_act_mul_against_fp64_finite_differences(cls, gid):
    n = cls.manifold_dim
    g = np.random.default_rng(40 + gid)
    a0, b0 = g.normal(0, 0.4, n), g.normal(0, 0.4, n)
    Wm, pts = g.normal(0, 1, (4, 4)), g.normal(0, 1, (6, 3))

    def loss64(a, b):
        M = LO.exp_matrix(gid, a) @ LO.exp_matrix(gid, b)
        Mi = np.linalg.inv(LO.exp_matrix(gid, a))
        act = pts @ M[:3, :3].T + M[:3, 3]
        return float((Wm * M).sum() + (act ** 2).sum() * 0.1 + (Wm.T * Mi).sum() * 0.3)

    a = torch.tensor(a0, dtype=torch.float32, device=DEV, requires_grad=True)
    b = torch.tensor(b0, dtype=torch.float32, device=DEV, requires_grad=True)
    X, Y = cls.exp(a[None]), cls.exp(b[None])
    Z = X * Y
    M = Z.matrix()[0]
    act = Z.act(torch.from_numpy(pts).float().to(DEV)[None])[0]
    loss = (torch.from_numpy(Wm).float().to(DEV) * M).sum() + (act ** 2).sum() * 0.1 + \
           (torch.from_numpy(Wm.T).float().to(DEV) * X.inv().matrix()[0]).sum() * 0.3
    loss.backward()
    assert abs(loss.item() - loss64(a0, b0)) < 2e-4 * max(1, abs(loss64(a0, b0)))
    eps = 1e-6
    for var, grad, which in ((a0, a.grad, 0), (b0, b.grad, 1)):
        fd = np.zeros(n)
        for j in range(n):
            e = np.zeros(n); e[j] = eps
            args_p = (a0 + e, b0) if which == 0 else (a0, b0 + e)
            args_m = (a0 - e, b0) if which == 0 else (a0, b0 - e)
            fd[j] = (loss64(*args_p) - loss64(*args_m)) / (2 * eps)
        np.testing.assert_allclose(grad.cpu().numpy(), fd, rtol=2e-3, atol=2e-3)


def test_log_autograd_and_reference_call_shapes():
    # track_backend.py:269-270: T = SE3.exp(cat([zeros(1,6), xi])).matrix()  ->  [B,4,4] with gradient to xi
    xi = torch.zeros(5, 6, device=DEV, requires_grad=True)
    lie = torch.cat([torch.zeros(1, 6, device=DEV), xi], 0)
    T = SE3.exp(lie).matrix()
    assert T.shape == (6, 4, 4)
    np.testing.assert_allclose(T.detach().cpu().numpy(), np.tile(np.eye(4), (6, 1, 1)), atol=1e-7)
    (T[:, :3, 3].sum() + T[:, 0, 1].sum()).backward()
    gr = xi.grad.cpu().numpy()
    np.testing.assert_allclose(gr[:, :3], 1.0, atol=1e-6)          # d t / d tau = I at identity
    np.testing.assert_allclose(gr[:, 5], -1.0, atol=1e-6)          # d R01 / d phi_z = -1
    # SE3(data).matrix() with data = [t, q_xyzw] (gs_backend_per_frame.py:721-725)
    d = torch.tensor([[1.0, 2.0, 3.0, 0.0, 0.0, np.sin(0.25), np.cos(0.25)]], device=DEV)
    M = SE3(d).matrix()[0].cpu().numpy()
    np.testing.assert_allclose(M[:3, 3], [1, 2, 3], atol=1e-6)
    np.testing.assert_allclose(M[:2, :2], [[np.cos(0.5), -np.sin(0.5)], [np.sin(0.5), np.cos(0.5)]], atol=1e-6)
    # log backward: d/dX of sum(log(X)) matches finite differences through exp
    a = torch.tensor([[0.2, -0.1, 0.3, 0.3, 0.2, -0.4]], device=DEV, requires_grad=True)
    SE3.exp(a).log().sum().backward()
    np.testing.assert_allclose(a.grad.cpu().numpy(), np.ones((1, 6)), atol=2e-4)   # log(exp(a)) = a => gradient 1
