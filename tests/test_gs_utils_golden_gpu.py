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
