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


# ------------------------------------------------------------------------------------------------ round 2: the rest of row A13
def _f(a):
    return torch.from_numpy(np.asarray(a)).float().to(DEV)


def test_moba_matches_dense_pose_only_solve():
    from cut3r_slam_amd.ba import MoBA
    P, ht, wd, fixedp = 5, 8, 10, 1
    poses, disps, intr, ii, jj, tgt, wgt, eta = _scene(P, ht, wd, 3)
    dx_ref = BO.moba_dense(torch.from_numpy(tgt), torch.from_numpy(wgt), poses, torch.from_numpy(disps), intr, ii, jj, fixedp)
    new_poses = MoBA(_f(tgt)[None], _f(wgt)[None], _f(eta), SE3(_f(poses)[None]), _f(disps)[None], _f(intr)[None], torch.from_numpy(ii),
                     torch.from_numpy(jj), fixedp=fixedp)
    torch.cuda.synchronize()
    Mnew = new_poses.matrix()[0].cpu().double()
    G = BO.se3_matrix(poses)
    for p in range(P):
        ref = G[p] if p < fixedp else torch.from_numpy(LO.exp_matrix(1, dx_ref[p - fixedp].numpy())) @ G[p]
        np.testing.assert_allclose(Mnew[p].numpy(), ref.numpy(), atol=5e-4)
    assert dx_ref.abs().max() > 1e-3                                # the step is not trivially zero


def test_proj_trans_and_bi_inter_match_their_restatements():
    P, ht, wd = 4, 6, 8
    poses, disps, intr, ii, jj, tgt, wgt, eta = _scene(P, ht, wd, 9)
    C_ref, w_ref, kx = BO.proj_trans_dense(torch.from_numpy(tgt), torch.from_numpy(wgt), poses, torch.from_numpy(disps), intr, ii, jj)
    C, w = db.proj_trans(_f(poses), _f(disps), _f(intr[0]), _f(tgt)[None], _f(wgt)[None], torch.from_numpy(ii), torch.from_numpy(jj))
    torch.cuda.synchronize()
    np.testing.assert_allclose(C.cpu().numpy(), C_ref.numpy(), rtol=2e-3, atol=1e-6 * float(C_ref.abs().max()) + 1e-9)
    np.testing.assert_allclose(w.cpu().numpy(), w_ref.numpy(), rtol=2e-3, atol=2e-3 * float(w_ref.abs().max()))
    g = torch.Generator().manual_seed(1)
    M, hs, ws = 3, 3, 4
    scales = torch.rand(M, hs, ws, generator=g) + 0.5
    grid = torch.stack([torch.rand(M, ht, wd, generator=g) * (ws - 1), torch.rand(M, ht, wd, generator=g) * (hs - 1)], -1)
    v, J = db.bi_inter(scales.to(DEV), grid.to(DEV))
    v_ref, J_ref = BO.bi_inter_ref(scales, grid)
    np.testing.assert_allclose(v.cpu().numpy(), v_ref.numpy(), atol=1e-6)
    np.testing.assert_allclose(J.cpu().numpy(), J_ref.numpy(), atol=1e-6)
    np.testing.assert_allclose(J.sum(-1).cpu().numpy(), 1.0, atol=1e-6)                  # interior points: weights sum to one


def test_jdsa_matches_unreduced_dense_solve():
    from cut3r_slam_amd.ba import JDSA
    P, ht, wd, hs, ws, alpha = 4, 6, 8, 2, 3, 0.05
    poses, disps, intr, ii, jj, tgt, wgt, eta = _scene(P, ht, wd, 5)
    g = np.random.default_rng(2)
    prior = disps * g.uniform(0.8, 1.2, disps.shape)
    prior[:, :2, :3] = 0.0                                          # pixels without a prior fall back to eta (ba.py:216-217)
    scales = g.uniform(0.9, 1.1, (P, hs, ws))
    C_ref, w_ref, kx = BO.proj_trans_dense(torch.from_numpy(tgt), torch.from_numpy(wgt), poses, torch.from_numpy(disps), intr, ii, jj)
    dz_ref, dso_ref = BO.jdsa_dense(C_ref, w_ref, torch.from_numpy(eta), torch.from_numpy(disps[kx]), torch.from_numpy(prior[kx]),
                                    torch.from_numpy(scales[kx]), alpha)
    dsc = _f(scales).clone()
    new_disps, new_scales, dzcov = JDSA(_f(tgt)[None], _f(wgt)[None], _f(eta), SE3(_f(poses)[None]), _f(disps)[None], _f(intr)[None],
                                        _f(prior), dsc, torch.from_numpy(ii), torch.from_numpy(jj), alpha)
    torch.cuda.synchronize()
    ref_d = torch.from_numpy(disps).clone()
    ref_d[torch.from_numpy(kx)] += dz_ref.reshape(-1, ht, wd)
    ref_d = torch.where(ref_d > 10, torch.zeros_like(ref_d), ref_d).clamp(min=0.001)
    sz, ss = float(dz_ref.abs().max()), float(dso_ref.abs().max())
    assert sz > 1e-4 and ss > 1e-4
    np.testing.assert_allclose(new_disps[0].cpu().numpy(), ref_d.numpy(), atol=1e-2 * sz + 1e-6)
    np.testing.assert_allclose((new_scales.cpu().double() - torch.from_numpy(scales))[torch.from_numpy(kx)].reshape(len(kx), -1).numpy(),
                               dso_ref.numpy(), atol=1e-2 * ss + 1e-6)
    assert dzcov.shape == (len(kx), ht * wd) and bool((dzcov > 0).all())


def test_schur_solve_mono_prior_matches_fp64_restatement_including_the_covariance():
    """geom/chol.py:80-107 on the HIP kernels (cut3r_schur_mono_prior: reduction, damping, in-LDS Cholesky, back-substitution, column-wise
    covariance) against the same formulas in fp64 torch on the CPU, with GENERAL (not block-diagonal) Hs / Es; a non-SPD system gives
    the reference's swallowed failure: dso = 0, dz = w / C (CholeskySolver, chol.py:9-18).  Parity unpinned (dead code in the reference)."""
    from cut3r_slam_amd.ba import schur_solve_mono_prior
    g = torch.Generator().manual_seed(4)
    M, D, HW = 3, 6, 40
    Es = torch.randn(1, M, M, D, HW, generator=g, dtype=torch.float64) * 0.3
    A = torch.randn(M * D, M * D, generator=g, dtype=torch.float64)
    Hfull = A @ A.T + 4.0 * torch.eye(M * D, dtype=torch.float64)
    Hs = Hfull.reshape(M, D, M, D).permute(0, 2, 1, 3)[None].contiguous()
    vs = torch.randn(1, M, D, generator=g, dtype=torch.float64)
    C = torch.rand(1, M, HW, generator=g, dtype=torch.float64) * 2 + 6.0
    w = torch.randn(1, M, HW, generator=g, dtype=torch.float64)

    def ref(C, w, Hs, Es, vs, ep=0.1, lm=1e-4):
        Q = (1.0 / C).view(1, M * HW, 1)
        wv = w.reshape(1, M * HW, 1)
        H = Hs.permute(0, 1, 3, 2, 4).reshape(1, M * D, M * D)
        E = Es.permute(0, 1, 3, 2, 4).reshape(1, M * D, M * HW)
        v = vs.reshape(1, M * D, 1)
        H = H + (ep + lm * H) * torch.eye(M * D, dtype=torch.float64)
        Et = E.transpose(1, 2)
        S = H - E @ (Q * Et)
        v = v - E @ (Q * wv)
        L = torch.linalg.cholesky(S)
        dso = torch.cholesky_solve(v, L)
        dz = Q * (wv - Et @ dso)
        Fm = torch.linalg.solve_triangular(L, E * Q[..., 0][:, None], upper=False)
        cov = (Fm ** 2).sum(1) + Q[..., 0]
        return dso.reshape(1, M, D), dz.reshape(1, M, HW), cov.reshape(M, HW)

    dso_r, dz_r, cov_r = ref(C, w, Hs, Es, vs)
    f = lambda t: t.float().to(DEV)
    dso, dz, cov = schur_solve_mono_prior(f(C), f(w), f(Hs), f(Es), f(vs), dzcov=True)
    torch.cuda.synchronize()
    for got, want, name in ((dso, dso_r, "dso"), (dz, dz_r, "dz"), (cov, cov_r, "dzcov")):
        err = float((got.cpu().double() - want).abs().max() / want.abs().max())
        assert err < 2e-4, (name, err)
    _, _, none = schur_solve_mono_prior(f(C), f(w), f(Hs), f(Es), f(vs), dzcov=False)
    assert none is None
    # failure path: an indefinite reduced system
    dso_b, dz_b, cov_b = schur_solve_mono_prior(f(C), f(w), f(-Hs), f(Es), f(vs), dzcov=True)
    assert float(dso_b.abs().max()) == 0.0
    np.testing.assert_allclose(dz_b.cpu().numpy(), (w / C).float().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(cov_b.cpu().numpy(), (1.0 / C)[0].float().numpy(), rtol=1e-5)


def test_altcorr_forward_and_backward_vs_all_pairs_volume():
    g = torch.Generator().manual_seed(4)
    BN, H, W, H2, W2, Cc, S, r = 2, 5, 6, 7, 8, 24, 2, 2
    f1 = torch.randn(BN, H, W, Cc, generator=g, dtype=torch.float64)
    f2 = torch.randn(BN, H2, W2, Cc, generator=g, dtype=torch.float64)
    coords = torch.stack([torch.rand(BN, S, H, W, generator=g, dtype=torch.float64) * (W2 + 3) - 1.5,
                          torch.rand(BN, S, H, W, generator=g, dtype=torch.float64) * (H2 + 3) - 1.5], -1)
    f1r, f2r = f1.clone().requires_grad_(True), f2.clone().requires_grad_(True)
    ref = BO.altcorr_ref(f1r, f2r, coords, r)
    G = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    (ref * G).sum().backward()
    a1, a2 = f1.float().to(DEV).requires_grad_(True), f2.float().to(DEV).requires_grad_(True)
    out = db.CorrLayer.apply(a1, a2, coords.float().to(DEV), r)
    (out * G.float().to(DEV)).sum().backward()
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), atol=2e-4)
    np.testing.assert_allclose(a1.grad.cpu().numpy(), f1r.grad.numpy(), atol=5e-4)
    np.testing.assert_allclose(a2.grad.cpu().numpy(), f2r.grad.numpy(), atol=5e-4)
    # AltCorrBlock == CorrBlock lookup semantics on level 0 (features / 4 on both sides)
    fm = torch.randn(1, 3, Cc, 8, 8, generator=g)
    blk = db.AltCorrBlock(fm.to(DEV), num_levels=2, radius=1)
    cc = torch.rand(1, 2, 8, 8, 2, generator=g) * 7
    res = blk(cc.to(DEV), torch.tensor([0, 1]), torch.tensor([1, 2]))
    assert res.shape == (1, 2, 2 * 9, 8, 8)
    v0 = BO.altcorr_ref((fm[0, [0, 1]] / 4).permute(0, 2, 3, 1).double(), (fm[0, [1, 2]] / 4).permute(0, 2, 3, 1).double(), cc[0][:, None].double(), 1)
    np.testing.assert_allclose(res[0, :, :9].cpu().numpy(), v0[:, 0].numpy(), atol=2e-4)


def test_edge_sharded_ba_sums_to_the_full_system():
    """north_star's second split: edges sharded by source frame, normal-equation blocks summed.  Two 'ranks' in one process: the
    undamped reduced systems of the two edge subsets add up to the full one, and BA over a fake two-rank all-reduce reproduces
    the single-rank step (poses, disparities)."""
    from cut3r_slam_amd import ba as B
    P, ht, wd, fixedp = 6, 8, 10, 1
    poses, disps, intr, ii, jj, tgt, wgt, eta = _scene(P, ht, wd, 7)
    args = (_f(tgt)[None], _f(wgt)[None], _f(eta), SE3(_f(poses)[None]), _f(disps)[None], _f(intr)[None], torch.from_numpy(ii), torch.from_numpy(jj))
    p_full, d_full, info = B.BA(*args, fixedp=fixedp)
    masks = [B.shard_edges_by_source(torch.from_numpy(ii), 2, r) for r in range(2)]
    assert bool((masks[0] ^ masks[1]).all()) and set(ii[masks[0].numpy()]).isdisjoint(set(ii[masks[1].numpy()]))
    parts = [B.BA(*args, fixedp=fixedp, edge_mask=m) for m in masks]           # each alone: only its S / vS are used below
    S = parts[0][2]["S"] + parts[1][2]["S"]
    vS = parts[0][2]["vS"] + parts[1][2]["vS"]
    torch.cuda.synchronize()
    sc = float(info["S"].abs().max())
    np.testing.assert_allclose(S.cpu().numpy(), info["S"].cpu().numpy(), atol=2e-5 * sc)
    np.testing.assert_allclose(vS.cpu().numpy(), info["vS"].cpu().numpy(), atol=2e-5 * float(info["vS"].abs().max()))


def test_depth_filter_counts_match_restatement():
    """droid_backends.depth_filter (droid_visualization.py:100): per-pixel count of confirming neighbour frames, whole numbers; pixels
    whose decision sits on an fp32 rounding of the threshold test (margin < 5e-5) are left out of the comparison"""
    g = torch.Generator().manual_seed(11)
    n, ht, wd = 9, 40, 56
    intr = torch.tensor([45.0, 46.0, 27.5, 19.5])
    y, x = torch.meshgrid(torch.arange(ht).float(), torch.arange(wd).float(), indexing="ij")
    depth0 = 2.5 + 0.4 * torch.sin(x / 9.0) + 0.3 * torch.cos(y / 7.0)
    poses = torch.zeros(n, 7)
    poses[:, 6] = 1.0
    tw = torch.randn(n, 6, generator=g) * torch.tensor([0.03, 0.03, 0.03, 0.01, 0.01, 0.01])
    poses = SE3.exp(tw.to(DEV)).data.cpu()                      # world->camera, small baselines around one scene
    # every frame sees the same smooth surface, rendered through its own pose (first-order: depth0 + noise), so most pixels agree
    disps = (1.0 / (depth0[None] + 0.02 * torch.randn(n, ht, wd, generator=g))).contiguous()
    disps[3, 5:12, 8:20] = 2.0                                   # an inconsistent blob: must lose its confirmations
    inds = torch.tensor([0, 3, 4, 8])
    thresh = torch.tensor([0.02, 0.05, 0.02, 0.1])
    got = db.depth_filter(poses.to(DEV), disps.to(DEV), intr.to(DEV), inds.to(DEV), thresh.to(DEV)).cpu().numpy()
    ref, margin = BO.depth_filter_ref(poses.numpy(), disps.numpy(), intr.numpy(), inds.numpy(), thresh.numpy())
    assert got.shape == (4, ht, wd) and np.array_equal(got, np.round(got))
    safe = margin > 5e-5
    assert safe.mean() > 0.97
    np.testing.assert_array_equal(got[safe], ref[safe])
    assert ref.max() >= 3 and (ref == 0).any()                   # both confirmed and rejected pixels are present
    assert ref[1, 5:12, 8:20].mean() < 0.5 * ref[1].mean()       # the blob is rejected
    with pytest.raises(IndexError):
        db.depth_filter(poses.to(DEV), disps.to(DEV), intr.to(DEV), torch.tensor([9]).to(DEV), thresh[:1].to(DEV))


def _chol_fixture():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "chol.npz"))


def test_schur_solve_mono_prior_equals_the_reference_chol_py():
    """cut3r_schur_mono_prior against the REFERENCE'S OWN geom/chol.py:80-107, run on the CPU by tests/golden/make_fixtures.py (`chol.npz`;
    general blocks): dso, dz and the covariance.  (The reference's fp32 run is 3e-7 from its fp64 run; the fp64 outputs are the target.)"""
    from cut3r_slam_amd.ba import schur_solve_mono_prior
    f = _chol_fixture()
    t = lambda k: torch.from_numpy(f[k]).float().to(DEV)
    dso, dz, cov = schur_solve_mono_prior(t("mp_C"), t("mp_w"), t("mp_Hs"), t("mp_Es"), t("mp_vs"), dzcov=True)
    torch.cuda.synchronize()
    for got, name in ((dso, "mp_dso"), (dz, "mp_dz"), (cov, "mp_cov")):
        ref = f[name + "_f64"]
        err = float(np.abs(got.cpu().double().numpy() - ref).max() / np.abs(ref).max())
        print(f"[schur_solve_mono_prior vs reference chol.py] {name}: {err:.1e}")
        assert err < 5e-5, (name, err)


def test_pose_system_solve_equals_the_reference_schur_solve():
    """cut3r_ba_solve (damping with diag H, in-LDS Cholesky) on the reduced system of the fixture against the reference's schur_solve
    (geom/chol.py:45-78, `chol.npz`): S = H - E C^-1 E^T and vS = v - E C^-1 w are formed on the host in fp64 exactly as :60-61 do, the
    kernel adds (ep + lm H_ii) and solves; dx equals the reference's (full and `sless` call), and the back-substituted dz = (w - E^T dx) / C."""
    from cut3r_slam_amd.ba import _solve
    f = _chol_fixture()
    H, E, C, v, w = (torch.from_numpy(f[k]) for k in ("ss_H", "ss_E", "ss_C", "ss_v", "ss_w"))
    _, P, M, D, HW = E.shape
    Hm = H.permute(0, 1, 3, 2, 4).reshape(P * D, P * D)
    Em = E.permute(0, 1, 3, 2, 4).reshape(P * D, M * HW)
    Q = (1.0 / C).reshape(M * HW)
    S = Hm - (Em * Q) @ Em.T
    vS = v.reshape(-1) - Em @ (Q * w.reshape(-1))
    dx, flag = _solve(S.float().to(DEV).contiguous(), vS.float().to(DEV).contiguous(), torch.diagonal(Hm).float().to(DEV).contiguous(), 0.1, 1e-4)
    torch.cuda.synchronize()
    assert int(flag.item()) == 0
    got = dx.cpu().double().numpy().reshape(1, P, D)
    for name in ("ss_dx_f64", "ss_dx_sless_f64"):
        err = float(np.abs(got - f[name]).max() / np.abs(f[name]).max())
        print(f"[cut3r_ba_solve vs reference schur_solve] {name}: {err:.1e}")
        assert err < 5e-5, (name, err)
    dz = (Q * (w.reshape(-1) - Em.T @ torch.from_numpy(got).reshape(-1))).reshape(1, M, HW).numpy()
    assert float(np.abs(dz - f["ss_dz_f64"]).max() / np.abs(f["ss_dz_f64"]).max()) < 5e-5


def test_reduced_systems_larger_than_the_in_lds_cholesky():
    """`M * D > 192` used to be refused.  JDSA's system is block diagonal (frame k's H / E blocks sit at (kx, kx): geom/ba.py
    :213-228), so it is solved as M independent D x D systems -- which must equal the dense solve wherever both run; the general
    `schur_solve_mono_prior` falls back to dense tensor ops (geom/chol.py:80-107 term by term) and must equal the kernel path on a
    system both can take."""
    from cut3r_slam_amd import ba as BA
    # (1) JDSA: the per-frame solves == the dense path (forced by lowering the limit) on the scene of test_jdsa_matches_unreduced_dense_solve
    P, ht, wd, hs, ws, alpha = 4, 6, 8, 2, 3, 0.05
    poses, disps, intr, ii, jj, tgt, wgt, eta = _scene(P, ht, wd, 5)
    g = np.random.default_rng(2)
    prior = disps * g.uniform(0.8, 1.2, disps.shape)
    prior[:, :2, :3] = 0.0
    scales = g.uniform(0.9, 1.1, (P, hs, ws))
    outs = []
    for limit in (BA.CHOL_MAXN, 4):                       # 4 < M * D: every frame on its own
        was = BA.CHOL_MAXN
        BA.CHOL_MAXN = limit
        try:
            dsc = _f(scales).clone()
            nd, ns, cov = BA.JDSA(_f(tgt)[None], _f(wgt)[None], _f(eta), SE3(_f(poses)[None]), _f(disps)[None], _f(intr)[None], _f(prior), dsc,
                                  torch.from_numpy(ii), torch.from_numpy(jj), alpha)
            torch.cuda.synchronize()
            outs.append((nd.cpu(), ns.cpu(), cov.cpu()))
        finally:
            BA.CHOL_MAXN = was
    for a, b, name in zip(outs[0], outs[1], ("disps", "scales", "dzcov")):
        np.testing.assert_allclose(b.numpy(), a.numpy(), rtol=2e-5, atol=1e-7, err_msg=name)
    # (2) a JDSA window with more than 192 / D source frames runs (50 frames x D = 4 = 200 > 192) and moves the disparities
    P2, hs2, ws2 = 52, 2, 2
    poses2, disps2, intr2, ii2, jj2, tgt2, wgt2, eta2 = _scene(P2, ht, wd, 7)
    assert len(np.unique(ii2)) * hs2 * ws2 > 192
    prior2 = disps2 * g.uniform(0.8, 1.2, disps2.shape)
    dsc2 = _f(g.uniform(0.9, 1.1, (P2, hs2, ws2))).clone()
    nd2, ns2, cov2 = BA.JDSA(_f(tgt2)[None], _f(wgt2)[None], _f(eta2), SE3(_f(poses2)[None]), _f(disps2)[None], _f(intr2)[None], _f(prior2), dsc2,
                             torch.from_numpy(ii2), torch.from_numpy(jj2), alpha)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(nd2).all()) and bool(torch.isfinite(cov2).all()) and float((nd2[0].cpu() - torch.from_numpy(disps2).float()).abs().max()) > 1e-5
    # (3) the general solve: kernel path == dense-tensor fallback on a general (not block diagonal) system
    gt = torch.Generator().manual_seed(4)
    M, D, HW = 3, 6, 40
    Es = (torch.randn(1, M, M, D, HW, generator=gt) * 0.3).to(DEV)
    A = torch.randn(M * D, M * D, generator=gt)
    Hs = (A @ A.T + 4.0 * torch.eye(M * D)).reshape(M, D, M, D).permute(0, 2, 1, 3)[None].contiguous().to(DEV)
    vs, C, w = torch.randn(1, M, D, generator=gt).to(DEV), (torch.rand(1, M, HW, generator=gt) * 2 + 6.0).to(DEV), torch.randn(1, M, HW, generator=gt).to(DEV)
    k = BA.schur_solve_mono_prior(C, w, Hs, Es, vs, dzcov=True)
    f = BA._schur_mono_prior_large(C, w, Hs, Es, vs, 0.1, 1e-4, True)
    for a, b in zip(k, f):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=2e-4, atol=2e-5)
