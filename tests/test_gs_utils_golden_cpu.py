"""CPU: the pure-torch pieces of the reference's Gaussian-splatting backend (run on the CPU by tests/golden/make_fixtures.py, `gs_utils.npz`)
against the oracle's restatements (oracle/gs_loss_oracle.py) and the product's host-side mirrors (gs_mapper.py): SSIM with its gradient,
exp of se(3), the projection matrix of getProjectionMatrix2, the position learning-rate schedule, inverse sigmoid."""
import os

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def fixture():
    return np.load(os.path.join(GOLD, "gs_utils.npz"))


def test_oracle_ssim_and_se3_exp_equal_the_reference():
    from oracle import gs_loss_oracle as LO
    f = fixture()
    a = torch.from_numpy(f["ssim_a"]).requires_grad_(True)
    v = LO.ssim(a, torch.from_numpy(f["ssim_b"]))
    v.backward()
    assert abs(float(v.detach()) - float(f["ssim_value"])) < 1e-12
    np.testing.assert_allclose(a.grad.numpy(), f["ssim_grad_a"], atol=1e-12 * np.abs(f["ssim_grad_a"]).max() + 1e-15, rtol=1e-9)
    for tau, ref in zip(f["tau"], f["se3_exp"]):
        got = LO.se3_exp_matrix(torch.from_numpy(tau[:3]), torch.from_numpy(tau[3:])).numpy()
        np.testing.assert_allclose(got, ref, atol=1e-12)


def test_product_host_mirrors_equal_the_reference():
    from cut3r_slam_amd import gs_mapper as GM
    f = fixture()
    # getProjectionMatrix2 (graphics_utils.py:72-93): the form with P[0,2] = 2 cx / W - 1, stored transposed by the cameras
    for args, ref in zip(f["proj_args"], f["proj"]):
        zn, zf, cx, cy, fx, fy, W, H = args
        cam = GM.Camera(0, torch.zeros(3, int(H), int(W)), torch.ones(int(H), int(W)), torch.eye(4), fx, fy, cx, cy, device="cpu")
        np.testing.assert_allclose(cam.projection_matrix_host.T.numpy(), ref, atol=2e-7)
    # the position learning-rate schedule (general_utils.py:41-56)
    op = {"position_lr_init": 0.00016, "position_lr_final": 0.0000016, "position_lr_max_steps": 29000}
    got = np.asarray([GM.position_lr(op, int(s)) for s in f["lr_steps"]])
    np.testing.assert_allclose(got, f["lr"], rtol=1e-12)
    np.testing.assert_allclose(GM.inverse_sigmoid(torch.from_numpy(f["isig_x"])).numpy(), f["isig"], rtol=1e-6)
