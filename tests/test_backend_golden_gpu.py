"""GPU: loop-candidate scoring and the host pose helpers against fixtures produced by the REFERENCE itself on the CPU
(tests/golden/make_fixtures.py: FactorGraph.NMS / compute_feature_overlap_batch / cal_overlap_bi, hislam2/factor_graph.py:561-582,
328-341, 284-315; pose_encoding_to_camera / geotrf / pose_vec_to_matrix)."""
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from cut3r_slam_amd import geom_host as gh  # noqa: E402
from cut3r_slam_amd import ops  # noqa: E402
from cut3r_slam_amd.keyframe import KeyFrame  # noqa: E402
from cut3r_slam_amd.track_backend import TrackBackend  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda:0"


def test_nms_scores_and_choice_equal_the_reference():
    f = np.load(os.path.join(GOLD, "nms.npz"))
    pm, c2w, feats = f["pointmaps"], f["c2w"], f["feats"]
    n, h, w, _ = pm.shape
    kf = KeyFrame({}, (2 * h, 2 * w), buffer=n + 6, downsample_ratio=2, device=DEV, feat_dim=feats.shape[2], patch=8)
    assert kf.featI.shape[1] == feats.shape[1]
    for j in range(n):
        kf.submap_ds[j // 5, j % 5] = torch.from_numpy(pm[j]).to(DEV)
    kf.featI[:n] = torch.from_numpy(feats).to(DEV)
    kf.w2c[:n] = torch.from_numpy(gh.w2c_rows(c2w)).to(DEV)
    kf.counter.value = n
    be = TrackBackend(types.SimpleNamespace(model=None, graph=None, downsample_ratio=2), kf, {"iteration": 0}, DEV)
    K = f["K"]
    K4 = [float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2])]
    cur = int(f["idx_current"])
    for name in "abc":
        ids = f[f"{name}_ids"]
        scores = be.nms_scores(ids, cur, K4).numpy()
        # overlap terms are integer counts / (h*w) and the feature term a count / (N-1): exact up to fp32 rounding of the sum
        np.testing.assert_allclose(scores, f[f"{name}_scores"], rtol=0, atol=2e-6, err_msg=name)
        for th in (0.4, 0.95):
            k = be.nms(ids, cur, K4, th=th)
            assert (-1 if k is None else k) == int(f[f"{name}_k_th{int(th * 100)}"]), (name, th)
        feat = be._feat_overlap(kf.featI[cur], [kf.featI[int(i)] for i in ids]).cpu().numpy()
        np.testing.assert_allclose(feat, f[f"{name}_feat_sim"], rtol=0, atol=1e-6)
        # the batched form (round 4: all candidates through the motion filter's look-ahead chain, nobody taken) counts what the per-pair
        # launches count
        pair = torch.cat([be._feat_overlap(kf.featI[cur], [kf.featI[int(i)]]) for i in ids]).cpu().numpy()
        assert np.array_equal(feat, pair), name


def test_pose_helpers_and_aligned_pointmap_equal_the_reference():
    f = np.load(os.path.join(GOLD, "camera.npz"))
    c2w = gh.pose_encoding_to_camera(f["enc"])
    np.testing.assert_allclose(c2w, f["c2w"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(gh.pose_vec_to_matrix(f["pose_vec"]), f["pose_vec_c2w"], rtol=0, atol=1e-6)
    # geotrf(c2w, pts) == the HIP window-alignment kernel with scale 1, stride 1 (track_frontend.py:199,234)
    pts = torch.from_numpy(f["pts"]).to(DEV)
    B, H, W, _ = pts.shape
    for b in range(B):
        pm = torch.empty(H, W, 3, device=DEV)
        cf = torch.empty(H, W, device=DEV)
        dp = torch.empty(H, W, device=DEV)
        ops.align_view(pts[b].contiguous(), torch.full((H, W), 2.0, device=DEV), f["c2w"][b][:3, :4].reshape(-1), 1.0, 1, pm, cf, dp)
        np.testing.assert_allclose(pm.cpu().numpy(), f["geotrf"][b], rtol=1e-6, atol=2e-6)
