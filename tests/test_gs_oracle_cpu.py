"""oracle/gs_oracle.py (the fp64 restatement of the reference's Gaussian rasteriser) has no reference-run fixture to be pinned to
(the reference's rasteriser is a CUDA extension): PARITY UNPINNED.  These CPU tests check what can be checked without it -- closed-form
values of a single Gaussian, invariance under a rigid motion of scene and camera, and the tile-rectangle / culling rules as cited."""
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import gs_oracle as GO  # noqa: E402

D = torch.float64


def test_single_gaussian_closed_form():
    """one isotropic Gaussian on the optical axis: centre pixel alpha = min(0.99, opacity * coef), colour = alpha c + (1 - alpha) bg, depth =
    ray distance / ln = z at the centre, radius = ceil(3 sigma_px); forward.cu:377-410,535-560"""
    H = W = 33
    z, s, op = 2.0, 0.05, 0.6
    st = GO.camera_settings(H, W, 1.0, 1.0, torch.eye(4, dtype=D), bg=(0.1, 0.2, 0.3))
    fx = W / (2 * math.tan(0.5))
    # the pixel whose centre the mean projects to: ndc2Pix(0) = (W - 1) / 2 = 16
    out = GO.rasterize(torch.tensor([[0.0, 0.0, z]], dtype=D), torch.tensor([[op]], dtype=D), torch.full((1, 3), s, dtype=D),
                       torch.tensor([[1.0, 0, 0, 0]], dtype=D), st, colors_precomp=torch.tensor([[0.9, 0.5, 0.2]], dtype=D))
    sigma_px = fx * s / z
    det = sigma_px ** 4
    coef = math.sqrt(det / (det + 1e-6) + 1e-6)                           # forward.cu:122-124 with kernel_size 0
    a = float(out["alpha"][0, 16, 16])
    assert abs(a - op * coef) < 1e-12
    np.testing.assert_allclose(out["color"][:, 16, 16].numpy(), a * np.array([0.9, 0.5, 0.2]) + (1 - a) * np.array([0.1, 0.2, 0.3]), atol=1e-6)
    # depth image = ray distance / ln; ln is taken about (W/2, H/2) (forward.cu:466-467) while the mean lands on (W-1)/2: half a pixel off
    ln = math.sqrt(2 * (0.5 / fx) ** 2 + 1)
    assert abs(float(out["depth"][0, 16, 16]) - z / ln) < 1e-9 and abs(float(out["mdepth"][0, 16, 16]) - z / ln) < 1e-9
    assert int(out["radii"][0]) == math.ceil(3 * math.sqrt(sigma_px ** 2 + math.sqrt(0.1)))    # lambda1 = mid + sqrt(max(0.1, mid^2 - det)), :394-397
    # falloff along a row: alpha(dx) = op * exp(-dx^2 / (2 sigma_px^2)) while >= 1/255
    for dx in (1, 2, 3):
        expect = op * coef * math.exp(-dx * dx / (2 * sigma_px ** 2))
        got = float(out["alpha"][0, 16, 16 + dx])
        assert abs(got - (expect if expect >= 1 / 255 else 0.0)) < 1e-9
    assert float(out["normal"][2, 16, 16]) < -0.99                        # the normal of a fronto-parallel blob faces the camera


def test_rigid_motion_of_scene_and_camera_changes_nothing():
    g = torch.Generator().manual_seed(2)
    P = 40
    means = torch.randn(P, 3, generator=g, dtype=D) * 0.5 + torch.tensor([0, 0, 3.0], dtype=D)
    scales = torch.rand(P, 3, generator=g, dtype=D) * 0.15 + 0.03
    q = torch.randn(P, 4, generator=g, dtype=D)
    q = q / q.norm(dim=-1, keepdim=True)
    op = torch.rand(P, 1, generator=g, dtype=D) * 0.8 + 0.1
    col = torch.rand(P, 3, generator=g, dtype=D)
    st0 = GO.camera_settings(24, 32, 0.9, 0.7, torch.eye(4, dtype=D))
    a = GO.rasterize(means, op, scales, q, st0, colors_precomp=col)
    tw = torch.tensor([0.3, -0.2, 0.4, 0.5, -0.3, 0.2], dtype=D)
    Wm = torch.zeros(4, 4, dtype=D)
    Wm[0, 1], Wm[0, 2], Wm[1, 2] = -tw[5], tw[4], -tw[3]
    Wm = Wm - Wm.T
    Wm[:3, 3] = tw[:3]
    T = torch.matrix_exp(Wm)                                              # world motion
    R = T[:3, :3]
    # quaternion of R (r, x, y, z), then q' = q_R * q
    r = math.sqrt(max(0.0, 1 + float(R.trace()))) / 2
    qR = torch.tensor([r, float(R[2, 1] - R[1, 2]) / (4 * r), float(R[0, 2] - R[2, 0]) / (4 * r), float(R[1, 0] - R[0, 1]) / (4 * r)], dtype=D)
    r1, x1, y1, z1 = qR
    r2, x2, y2, z2 = q.unbind(-1)
    q2 = torch.stack([r1 * r2 - x1 * x2 - y1 * y2 - z1 * z2, r1 * x2 + x1 * r2 + y1 * z2 - z1 * y2, r1 * y2 - x1 * z2 + y1 * r2 + z1 * x2,
                      r1 * z2 + x1 * y2 - y1 * x2 + z1 * r2], -1)
    st1 = GO.camera_settings(24, 32, 0.9, 0.7, torch.inverse(T))         # w2c' = w2c T^-1
    b = GO.rasterize(means @ R.T + T[:3, 3], op, scales, q2, st1, colors_precomp=col)
    for k in ("color", "alpha", "depth", "mdepth", "coord", "mcoord", "normal"):
        np.testing.assert_allclose(b[k].numpy(), a[k].numpy(), atol=1e-9, err_msg=k)
    assert torch.equal(a["radii"], b["radii"])


def test_culling_and_tile_rectangle_rules():
    """auxiliary.h:170 (view z <= 0.2 is culled), :62-72 (rectangle of 16x16 tiles from the integer radius), forward.cu:405-407"""
    st = GO.camera_settings(40, 56, 1.0, 0.8, torch.eye(4, dtype=D))
    means = torch.tensor([[0.0, 0.0, 0.2], [0.0, 0.0, 0.21], [50.0, 0.0, 3.0], [0.0, 0.0, 3.0]], dtype=D)
    one = torch.ones(4, 1, dtype=D)
    g = GO.preprocess(means, 0.5 * one, 0.05 * one.repeat(1, 3), torch.tensor([[1.0, 0, 0, 0]] * 4, dtype=D), None, torch.rand(4, 3, dtype=D), st)
    assert g["visible"].tolist() == [False, True, False, True]            # at the near limit / far outside the image: no tiles
    r = g["rect"][3].tolist()
    rad, xy = int(g["radii"][3]), g["xy"][3].tolist()
    assert r == [max(0, int((xy[0] - rad) / 16)), max(0, int((xy[1] - rad) / 16)), min(4, int((xy[0] + rad + 15) / 16)), min(3, int((xy[1] + rad + 15) / 16))]
