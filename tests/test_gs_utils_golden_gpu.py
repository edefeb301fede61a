"""GPU: kernels of the Gaussian mapper against the pure-torch pieces of the reference's GS backend run on the CPU (tests/golden/gs_utils.npz,
made by tests/golden/make_fixtures.py gs_utils): fused SSIM forward / backward, exp of se(3) on the lie kernels, get_pose / update_pose
(tensor-op form and the tape-free trainer's fold kernel), project2world."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def fixture():
    return np.load(os.path.join(GOLD, "gs_utils.npz"))


def test_fused_ssim_equals_the_reference_ssim_and_its_gradient():
    """cut3r_ssim_forward / backward vs loss_utils.ssim (:129-170) and its autograd gradient (fp64 on the reference side)"""
    from cut3r_slam_amd.gaussian_rasterizer import fused_ssim
    f = fixture()
    a = torch.from_numpy(f["ssim_a"]).float().to(DEV).requires_grad_(True)
    b = torch.from_numpy(f["ssim_b"]).float().to(DEV)
    v = fused_ssim(a, b)
    v.backward()
    torch.cuda.synchronize()
    err_v = abs(float(v.detach()) - float(f["ssim_value"]))
    g = f["ssim_grad_a"]
    err_g = float(np.abs(a.grad.cpu().double().numpy() - g).max() / np.abs(g).max())
    print(f"[fused ssim vs reference] value {err_v:.1e}, gradient {err_g:.1e} of its largest entry")
    assert err_v < 2e-6 and err_g < 2e-5


def test_lie_kernels_and_pose_update_equal_the_reference_slam_utils():
    """SE3.exp on the lie kernels vs slam_utils.SE3_exp (:26-75, incl. angles below its 1e-5 switch and a pure rotation of 2.8 rad);
    get_pose / update_pose (:77-102) in the tensor-op form and through cut3r_gs_pose_step's fold; project2world (:108-140)"""
    from cut3r_slam_amd import gs_mapper as GM
    from cut3r_slam_amd import _lib
    from cut3r_slam_amd.lietorch import SE3
    f = fixture()
    M = SE3.exp(torch.from_numpy(f["tau"]).float().to(DEV)).matrix().cpu().double().numpy()
    assert float(np.abs(M - f["se3_exp"]).max()) < 2e-6
    w2c = torch.eye(4)
    w2c[:3, :3], w2c[:3, 3] = torch.from_numpy(f["cam_R"]), torch.from_numpy(f["cam_T"])

    def camera():
        c = GM.Camera(0, torch.zeros(3, 8, 8), torch.ones(8, 8), w2c, 10.0, 10.0, 4.0, 4.0, device=DEV)
        c.cam_trans_delta.data.copy_(torch.from_numpy(f["cam_trans_delta"]))
        c.cam_rot_delta.data.copy_(torch.from_numpy(f["cam_rot_delta"]))
        return c
    cam = camera()
    assert float((GM.get_pose(cam).detach().cpu().double() - torch.from_numpy(f["get_pose"]).double()).abs().max()) < 2e-6
    GM.update_pose(cam)
    assert float(np.abs(cam.R.cpu().numpy() - f["updated_R"]).max()) < 2e-6 and float(np.abs(cam.T.cpu().numpy() - f["updated_T"]).max()) < 2e-6
    assert float(cam.cam_rot_delta.detach().abs().max()) == 0.0 and float(cam.cam_trans_delta.detach().abs().max()) == 0.0
    # the tape-free trainer's fold (cut3r_gs_pose_step with fold = 2: T <- exp(delta) T, delta <- 0, no gradient step)
    cam = camera()
    ps = torch.zeros(32, device=DEV)
    ps[0:7] = cam.w2c_data
    ps[7:10], ps[10:13] = cam.cam_trans_delta.detach(), cam.cam_rot_delta.detach()
    lib = _lib.load()
    sums = torch.zeros(16, device=DEV)
    assert lib.cut3r_gs_pose_step(C.c_void_p(ps.data_ptr()), C.c_void_p(sums.data_ptr()), 0.0, None, 0.0, 0.0, 2,
                                  C.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
    Mf = SE3(ps[None, 0:7].contiguous()).matrix()[0].cpu().numpy()
    assert float(np.abs(Mf[:3, :3] - f["updated_R"]).max()) < 2e-6 and float(np.abs(Mf[:3, 3] - f["updated_T"]).max()) < 2e-6
    assert float(ps[7:13].abs().max()) == 0.0
    pw = GM.project2world(torch.from_numpy(f["p2w_c2w"]).to(DEV), torch.from_numpy(f["p2w_depth"]).to(DEV), 20.0, 21.0, 7.5, 5.5)
    assert float(np.abs(pw.cpu().numpy() - f["p2w"]).max()) < 5e-6


def test_gaussian_map_equals_the_reference_gaussian_model_through_adam_densify_prune_reset():
    """GaussianMap (one [P,14] block, one fused Adam, row operations) against the reference's GaussianModel + torch.optim.Adam run on the CPU
    (tests/golden/gaussian_model.npz): 4 Adam steps on seeded gradients with the densification statistics of a rendered view each ->
    densify_and_prune (gradient OR absolute-gradient-quantile rule, clone / split with the reference's own normal draws replayed, prune incl.
    the degenerate-scale rule; the screen-size rule is dead after densification_postfix zeroes max_radii2D, as in the reference) -> 3 steps ->
    reset_opacity -> step -> second densification -> step -> an iteration that densifies BEFORE its optimiser step -> one that resets the opacities before it.  After every phase: every parameter, both Adam moments (zero-padded for new rows,
    cut for pruned ones, zeroed for the reset opacities), the shared step count and the decayed position learning rate."""
    from cut3r_slam_amd import gs_mapper as GM
    f = np.load(os.path.join(GOLD, "gaussian_model.npz"))
    o = f["opt"]
    op = {"position_lr_init": o[0], "position_lr_final": o[1], "position_lr_max_steps": int(o[2]), "feature_lr": o[3], "opacity_lr": o[4], "scaling_lr": o[5],
          "rotation_lr": o[6], "percent_dense": o[7]}
    extent, max_grad, min_opacity, size_threshold = float(o[8]), float(o[9]), float(o[10]), float(o[11])
    gm = GM.GaussianMap(op, DEV)
    t = lambda k: torch.from_numpy(f[k]).to(DEV)
    P = f["init_xyz"].shape[0]
    gm._append({"xyz": t("init_xyz"), "f_dc": t("init_f_dc").reshape(P, 3), "opacity": t("init_opacity"), "scaling": t("init_scaling"), "rotation": t("init_rotation")},
               torch.zeros(P))
    draws = [t(k) for k in sorted(k for k in f.files if k.startswith("split_draws_"))]
    gm._split_noise = lambda n: draws.pop(0)
    worst = {}

    def step(it, between=None, densified=False, reset=False):
        gm.theta.grad = t(f"grad_{it}").clone()
        gm.add_densification_stats(t(f"vs_{it}"), t(f"vis_{it}"))
        gm.max_radii2D = t(f"radii_{it}").clone()
        if between is not None:                  # where the reference's training loops densify / reset: between the statistics and optimizer.step()
            between()
        gm.step_like_reference(densified=densified, reset=reset)
        gm.zero_grad()
        gm.lr[0, 0:3] = GM.position_lr(op, it)

    def check(tag):
        torch.cuda.synchronize()
        th, m, v = gm.theta.detach().cpu().numpy(), gm.m.cpu().numpy(), gm.v.cpu().numpy()
        assert th.shape[0] == f[f"{tag}_xyz"].shape[0], (tag, th.shape[0], f[f"{tag}_xyz"].shape[0])
        for name, (a, b) in GM.GaussianMap.COLS.items():
            for what, got, key in (("value", th[:, a:b], f"{tag}_{name}"), ("m", m[:, a:b], f"{tag}_{name}_m"), ("v", v[:, a:b], f"{tag}_{name}_v")):
                ref = f[key]
                err = float(np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30))
                worst[what] = max(worst.get(what, 0.0), err)
                assert err < 5e-6, (tag, name, what, err)
        assert gm.steps == int(f[f"{tag}_step"]) and abs(float(gm.lr[0, 0]) - float(f[f"{tag}_lr_xyz"])) < 1e-10, (tag, gm.steps, float(gm.lr[0, 0]))
    for it in range(4):
        step(it)
    check("a")
    gm.densify_and_prune(max_grad, min_opacity, extent, size_threshold)
    check("b")
    for it in range(4, 7):
        step(it)
    check("c")
    gm.reset_opacity()
    check("d")
    step(7)
    gm.densify_and_prune(max_grad, min_opacity, extent, None)
    step(8)
    check("e")
    # the ORDER inside the reference's loops (gs_backend_per_frame.py:1025-1041): statistics -> densify_and_prune -> optimizer.step()
    # with no gradients on the re-created parameters (no update, no moment update, the step count stays at 9 in every group), then
    # statistics -> reset_opacity -> optimizer.step() (the opacity group is skipped, the others step)
    step(9, between=lambda: gm.densify_and_prune(max_grad, min_opacity, extent, None), densified=True)
    check("f")
    assert set(f["f_steps_by_group"].tolist()) == {9.0} and gm.steps == 9
    step(10, between=gm.reset_opacity, reset=True)
    check("g")
    assert dict(zip(f["group_names"].tolist(), f["g_steps_by_group"].tolist()))["opacity"] == 9.0       # (one count per block here: declared deviation)
    assert not draws
    print(f"[GaussianMap vs reference GaussianModel] {P} -> {int(f['n_after_densify'])} -> {len(gm)} Gaussians, worst relative errors", {k: f"{v:.1e}" for k, v in worst.items()})
