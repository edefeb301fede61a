"""GPU: legacy dense-BA operators (row A13).  PARITY UNPINNED vs the reference (its BA code cannot run and
droid_backends is absent); checked against oracle/ba_oracle.py = numerical Jacobians + dense solve of the UN-REDUCED
system in fp64, and F.grid_sample for the correlation lookup."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from cut3r_slam_amd import droid_backends as db  # noqa: E402
from cut3r_slam_amd.ba import BA  # noqa: E402
from cut3r_slam_amd.lietorch import SE3  # noqa: E402
from oracle import ba_oracle as BO  # noqa: E402
from oracle import lie_oracle as LO  # noqa: E402

DEV = "cuda:0"


def test_corr_index_forward_backward_vs_grid_sample():
    g = torch.Generator().manual_seed(0)
    BN, h1, w1, h2, w2, r = 3, 5, 6, 7, 9, 3
    vol = torch.randn(BN, h1, w1, h2, w2, generator=g)
    coords = torch.stack([torch.rand(BN, h1, w1, generator=g) * (w2 + 4) - 2, torch.rand(BN, h1, w1, generator=g) * (h2 + 4) - 2], 1)
    (out,) = db.corr_index_forward(vol.to(DEV), coords.to(DEV), r)
    ref = BO.corr_lookup(vol.double(), coords.double(), r)
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), atol=2e-5)
    # backward = transpose of the (linear) forward: <fwd(V), G> == <V, bwd(G)>
    G = torch.randn_like(out)
    (gv,) = db.corr_index_backward(vol.to(DEV), coords.to(DEV), G, r)
    lhs = float((out.double() * G.double()).sum())
    rhs = float((vol.to(DEV).double() * gv.double()).sum())
    assert abs(lhs - rhs) < 1e-4 * max(1.0, abs(lhs))


def _scene(P, ht, wd, seed):
    g = np.random.default_rng(seed)
    fx = fy = 0.8 * wd
    intr = np.tile(np.array([fx, fy, wd / 2 - 0.5, ht / 2 - 0.5]), (P, 1))
    # world->camera poses near identity, moving sideways; inverse depths ~ 0.4
    tang = np.zeros((P, 6))
    tang[:, 0] = np.linspace(0, 0.6, P) + g.normal(0, 0.02, P)
    tang[:, 1:3] = g.normal(0, 0.03, (P, 2))
    tang[:, 3:] = g.normal(0, 0.03, (P, 3))
    poses = np.stack([LO.matrix_to_data(1, LO.exp_matrix(1, a)) for a in tang])
    disps = g.uniform(0.3, 0.6, (P, ht, wd))
    ii, jj = [], []
    for i in range(P):
        for j in range(P):
            if i != j and abs(i - j) <= 2:
                ii.append(i); jj.append(j)
    ii, jj = np.array(ii), np.array(jj)
    N = len(ii)
    # targets: the true projections under perturbed geometry (so the residual is small but non-zero)
    G = BO.se3_matrix(poses)
    tgt = np.zeros((N, ht, wd, 2))
    for e in range(N):
        c, _ = BO.project(G[jj[e]] @ torch.linalg.inv(G[ii[e]]), torch.from_numpy(disps[ii[e]] * 1.05), intr[ii[e]], intr[jj[e]], ht, wd)
        tgt[e] = c.reshape(ht, wd, 2).numpy() + g.normal(0, 0.3, (ht, wd, 2))
    wgt = g.uniform(0.2, 1.0, (N, ht, wd, 2))
    eta = g.uniform(1e-3, 1e-2, (len(np.unique(ii)), ht, wd))
    return poses, disps, intr, ii, jj, tgt, wgt, eta


@pytest.mark.parametrize("P,ht,wd,fixedp", [(4, 6, 8, 1), (6, 12, 16, 2)])
def test_ba_step_matches_unreduced_dense_solve(P, ht, wd, fixedp):
    poses, disps, intr, ii, jj, tgt, wgt, eta = _scene(P, ht, wd, P)
    dx_ref, dz_ref, kx = BO.ba_dense(torch.from_numpy(tgt), torch.from_numpy(wgt), torch.from_numpy(eta), poses,
                                     torch.from_numpy(disps), intr, ii, jj, fixedp)
    f = lambda a: torch.from_numpy(np.asarray(a)).float().to(DEV)
    new_poses, new_disps, info = BA(f(tgt)[None], f(wgt)[None], f(eta), SE3(f(poses)[None]), f(disps)[None], f(intr)[None],
                                    torch.from_numpy(ii), torch.from_numpy(jj), fixedp=fixedp)
    torch.cuda.synchronize()
    assert int(info["failed"].item()) == 0
    dx, dz = info["dx"].cpu().double(), info["dz"].cpu().double()
    sx, sz = dx_ref.abs().max().item(), dz_ref.abs().max().item()
    assert (dx - dx_ref).abs().max().item() <= 5e-3 * sx + 1e-6, (dx, dx_ref)
    assert (dz - dz_ref).abs().max().item() <= 5e-3 * sz + 1e-6
    # retraction: poses <- exp(dx) * poses for the free poses, first `fixedp` untouched (ba.py:100-101)
    Mnew = new_poses.matrix()[0].cpu().double()
    G = BO.se3_matrix(poses)
    for p in range(P):
        ref = G[p] if p < fixedp else torch.from_numpy(LO.exp_matrix(1, dx_ref[p - fixedp].numpy())) @ G[p]
        np.testing.assert_allclose(Mnew[p].numpy(), ref.numpy(), atol=5e-4)
    nd = new_disps[0].cpu().double()
    ref_d = torch.from_numpy(disps).clone()
    ref_d[torch.from_numpy(kx)] += dz_ref.reshape(-1, ht, wd)
    ref_d = torch.where(ref_d > 10, torch.zeros_like(ref_d), ref_d).clamp(min=0.001)
    np.testing.assert_allclose(nd.numpy(), ref_d.numpy(), atol=5e-3 * sz + 1e-5)
