"""CPU: the oracle (oracle/) pinned against golden vectors produced by the REFERENCE (tests/golden/*.npz)."""
import json
import os

import numpy as np
import pytest
import torch

from cut3r_slam_amd.config import Cut3rConfig
from cut3r_slam_amd.weights import synth_state_dict
from oracle import cut3r_oracle as O
from oracle import geom as G

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _load_model_fixture(name):
    f = np.load(os.path.join(GOLD, name + ".npz"))
    cfg = Cut3rConfig.from_dict(json.loads(bytes(f["config_json"]).decode()))
    return f, cfg, synth_state_dict(cfg, int(f["seed"]))


@pytest.mark.parametrize("name", ["model_tiny_dpt", "model_tiny_linear", "model_medium_dpt"])
def test_model_oracle_matches_reference(name):
    f, cfg, sd = _load_model_fixture(name)
    imgs = O.normalize(torch.from_numpy(f["imgs"]))
    feat, pos = O.encode_image(cfg, sd, imgs[:1])
    assert torch.equal(pos, torch.from_numpy(f["enc_pos0"]))
    # fp32 re-association only (the medium fixture's 256-wide encoder: 8e-6 of the feature scale; the tiny ones: 1e-6)
    np.testing.assert_allclose(feat.numpy(), f["enc_feat0"], rtol=0, atol=2e-6 if "tiny" in name else 1e-5 * np.abs(f["enc_feat0"]).max())
    preds, states = O.forward_views(cfg, sd, imgs, return_states=True)
    for i, p in enumerate(preds):
        for k, v in p.items():
            g = f[f"pred{i}_{k}"]
            # fp32 re-association only: relative to the map's scale
            assert np.abs(v.numpy() - g).max() <= 2e-5 * max(1.0, np.abs(g).max()), (i, k)
    for i, (s, m) in enumerate(states):
        np.testing.assert_allclose(s.numpy(), f[f"state{i}_feat"], atol=1e-5)
        np.testing.assert_allclose(m.numpy(), f[f"state{i}_mem"], atol=2e-5 if "tiny" in name else 1e-5 * np.abs(f[f"state{i}_mem"]).max())
    np.testing.assert_array_equal(O.state_positions(cfg).numpy(), f["state_pos"])


def test_minimal_equals_full_on_consumed_keys():
    f, cfg, sd = _load_model_fixture("model_tiny_dpt")
    imgs = O.normalize(torch.from_numpy(f["imgs"]))
    a = O.forward_views(cfg, sd, imgs, minimal=True)
    for i, p in enumerate(a):
        assert set(p) == {"camera_pose", "pts3d_in_self_view", "conf_self"}
        for k, v in p.items():
            g = f[f"pred{i}_{k}"]
            assert np.abs(v.numpy() - g).max() <= 2e-5 * max(1.0, np.abs(g).max())


def test_rope_oracles_match_reference_cpu_kernel():
    f = np.load(os.path.join(GOLD, "rope2d.npz"))
    for D in (16, 48, 64):
        tok, pos = f[f"D{D}_tok"], f[f"D{D}_pos"]
        assert (pos < 0).any()                       # the pose-token position -1 is covered
        for F0 in (1, -1):
            ref = f[f"D{D}_F{F0}_out"]
            c = G.rope2d(tok, pos, 100.0, float(F0))
            # the reference CPU op evaluates fwd*p/powf(..) (curope.cpp:36), its CUDA kernel p*(fwd/powf(..))
            # (kernels.cu:43,53): a 1-ulp difference in an angle of up to 40 rad -> <= 1e-5 on |tokens| ~ 3
            np.testing.assert_allclose(c, ref, rtol=0, atol=1e-5)
            t = torch.from_numpy(tok).permute(0, 2, 1, 3)          # [B,H,N,D] as the model calls it
            p = O.rope2d(t, torch.from_numpy(pos), 100.0, float(F0)).permute(0, 2, 1, 3).numpy()
            np.testing.assert_allclose(p, ref, rtol=0, atol=1e-5)


def test_rope_oracle_vs_live_reference_build():
    from oracle import ref_curope
    if not ref_curope.available():
        pytest.skip("oracle/_ref/curope.so not built (needs /root/reference)")
    g = np.random.default_rng(0)
    tok = g.standard_normal((1, 7, 2, 32)).astype(np.float32)
    pos = g.integers(-1, 30, size=(1, 7, 2)).astype(np.int64)
    t = torch.from_numpy(tok.copy())
    ref_curope.rope_2d_ref(t, torch.from_numpy(pos), 100.0, 1.0)
    np.testing.assert_allclose(G.rope2d(tok, pos), t.numpy(), atol=1e-5)


def test_geometry_oracle_matches_reference_graph_outputs():
    f = np.load(os.path.join(GOLD, "graph.npz"))
    pm, c2w, K = f["pointmaps"], f["c2w"], f["K"]
    n, H, W, _ = pm.shape
    K4 = [K[0, 0], K[1, 1], K[0, 2], K[1, 2]]
    i = n - 1
    w2c = G.w2c_rows(c2w)
    fwd = G.overlap_fwd(pm[i], w2c[:i], K4, W, H) / float(H * W)
    bwd = G.overlap_bwd(pm[:i], w2c[i], K4, W, H) / float(H * W)
    # the reference sums through torch.bmm/inverse (backend-defined order): allow <= 2 border pixels of 768
    assert np.abs(fwd - f["ovl_batch_last"]).max() <= 2.0 / (H * W) + 1e-7
    assert np.abs(bwd - f["ovl_bi_last"].reshape(-1)).max() <= 2.0 / (H * W) + 1e-7
    # and the 0.3 decisions (what the graph topology depends on) are identical
    np.testing.assert_array_equal(fwd > 0.3, f["ovl_batch_last"] > 0.3)
    np.testing.assert_array_equal(bwd > 0.3, f["ovl_bi_last"].reshape(-1) > 0.3)


def test_patch_overlap_oracle_matches_reference():
    f = np.load(os.path.join(GOLD, "graph.npz"))
    for k, ref in enumerate(f["patch_ratios"]):
        r, _ = G.patch_overlap_ratio(f["feat0"], f[f"feat1_{k}"])
        assert abs(r - ref) < 1e-6
