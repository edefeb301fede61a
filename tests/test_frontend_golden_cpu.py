"""CPU: the oracle's tracker composition (oracle/slam_run.SlamOracle.track = TrackFrontend.track, hislam2/track_frontend.py:166-262) against
the REFERENCE'S OWN TrackFrontend.track, run on the CPU by tests/golden/make_fixtures.py (`frontend.npz`: initialisation window + two chained
windows of 16 keyframes, medium network, seeded weights): poses, depths, stride-2 submaps and confidences of every window, and the edge
list of the covisibility graph after every window.  This pins row A10's composition -- scale chaining, pose composition, store layout,
the order of the graph calls and the stores each `graph.add` sees -- which the earlier fixtures covered only piecewise."""
import os

import numpy as np
import torch

from cut3r_slam_amd import synth
from oracle import slam_run as SR

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load_fixture():
    f = np.load(os.path.join(GOLD, "frontend.npz"))
    cfg = synth.medium_config()
    sd = synth.tracking_state_dict(cfg, int(f["seed"]))
    H, W = cfg.img_size
    frames = synth.pan_stream(16, H, W, pool=5, num=6, den=1, seed=4)
    assert int(frames.long().sum()) == int(f["frames_sum"]), "the regenerated frames are not the fixture's"
    return f, cfg, sd, frames


def test_oracle_tracker_equals_the_reference_trackfrontend():
    f, cfg, sd, frames = load_fixture()
    H, W = cfg.img_size
    so = SR.SlamOracle(cfg, sd, (H, W), 20, {"thresh": 0.9, "skip": 1, "kf_every": 1})
    n = frames.shape[0]
    so.image[:n] = frames
    so.intrinsic[:n] = f["intrinsic"]
    so.counter = n
    worst = {}
    for w, (t0, t1, init) in enumerate(f["windows"].tolist()):
        so.track(t0, t1, init=bool(init))
        st = so.state
        got = {"pose": st["pose"][t0:t1].numpy(), "depth": st["depth"][t0:t1].numpy(), "submap_ds": st["submap_ds"][t0 // 5].numpy(),
               "conf_ds": st["conf_ds"][t0 // 5].numpy()}
        for k, v in got.items():
            ref = f[f"{k}_{w}"]
            err = float(np.abs(v - ref).max() / max(1e-12, np.abs(ref).max()))
            worst[k] = max(worst.get(k, 0.0), err)
            # fp32 CPU arithmetic on both sides (different GEMM blocking); seen: 5e-6 on the submaps, 2e-6 on poses and depths
            assert err < 2e-5, (w, k, err)
        ii, jj = so.graph.edges_numpy()[:2]
        np.testing.assert_array_equal(ii, f[f"ii_{w}"], err_msg=f"ii after window {w}")
        np.testing.assert_array_equal(jj, f[f"jj_{w}"], err_msg=f"jj after window {w}")
    print("[oracle tracker vs reference TrackFrontend.track] worst relative errors:", {k: f"{v:.1e}" for k, v in worst.items()},
          "| edges", [len(f[f"ii_{w}"]) for w in range(3)])
    # TrackFrontend.predict (:102-162) of the reference on the tracked map: frame 13 against keyframe 9
    from oracle import cut3r_oracle as O
    from oracle import slam_oracle as SO
    new, kfi = f["predict_args"].tolist()
    preds = O.forward_views(cfg, sd, O.normalize(torch.stack([so.image[kfi], frames[new]], 0)), minimal=True)
    pose, depth, pm, cf = SO.predict(torch.cat([p["pts3d_in_self_view"] for p in preds], 0), torch.cat([p["conf_self"] for p in preds], 0),
                                     torch.cat([p["camera_pose"] for p in preds], 0), so.state["pose"][kfi], so.state["depth"][kfi])
    for name, got in (("pose", pose), ("depth", depth), ("pointmap", pm), ("conf", cf)):
        ref = f[f"predict_{name}"]
        assert float(np.abs(got.numpy() - ref).max() / np.abs(ref).max()) < 2e-5, name
    # the fixture exercises the chain: the second and third window start from a non-trivial scale and pose
    assert abs(float(np.log(f["depth_1"][0]).mean())) > 0.1 and float(np.abs(f["pose_2"][0, :3]).max()) > 1e-3


STREAMS = {"overlap": (lambda H, W: synth.slideshow_stream(41, H, W, hold=4, seed=3), {"thresh": 0.9, "skip": 2, "kf_every": -1}),
           "blend": (lambda H, W: synth.blend_stream(40, H, W, period=12, seed=5), {"thresh": 0.9, "skip": 1, "kf_every": -1}),
           "cadence": (lambda H, W: synth.pan_stream(20, H, W, pool=5, num=2, den=1, seed=0), {"thresh": 0.9, "skip": 1, "kf_every": 3})}


def load_motion_filter_stream(name):
    f = np.load(os.path.join(GOLD, "motion_filter.npz"))
    cfg = synth.medium_config()
    sd = synth.tracking_state_dict(cfg, int(f["seed"]))
    frames = STREAMS[name][0](*cfg.img_size)
    assert int(frames.long().sum()) == int(f[f"{name}_frames_sum"]), "the regenerated frames are not the fixture's"
    return f, cfg, sd, frames, STREAMS[name][1]


def test_oracle_motion_filter_equals_the_reference_kffilter():
    """MotionFilter.kfFilter of the reference, run on the CPU by make_fixtures.py (`motion_filter.npz`): which frames become keyframes in
    overlap mode (ratios 0 / 1), along cross-fades (intermediate ratios on both sides of the 0.9 threshold), at a fixed cadence, with
    the second-last / last frame rules -- and every overlap ratio the reference computed."""
    for name in STREAMS:
        f, cfg, sd, frames, mf = load_motion_filter_stream(name)
        so = SR.SlamOracle(cfg, sd, cfg.img_size, 48, mf)
        n = frames.shape[0]
        intr = np.asarray([80.0, 80.0, 47.5, 31.5], np.float32)
        for t in range(n):
            so.kf_filter(t, frames[t], intr, second_last_frame=(t == n - 2), last_frame=(t == n - 1))
        np.testing.assert_array_equal(so.tstamp[:so.counter].astype(np.int64), f[f"{name}_keyframes"], err_msg=name)
        ratios = np.asarray([r for _, r in so.ratios])
        np.testing.assert_allclose(ratios, f[f"{name}_ratios"], atol=1e-9, err_msg=name)
        np.testing.assert_allclose(so.featI[so.counter - 1].numpy(), f[f"{name}_feat_last"], atol=2e-5 * np.abs(f[f"{name}_feat_last"]).max())
    inter = f["blend_ratios"]
    assert ((inter > 0.05) & (inter < 0.9)).any() and ((inter >= 0.9) & (inter < 1.0)).any()       # both sides of the threshold are exercised


def load_loop_fixture():
    f = np.load(os.path.join(GOLD, "loop.npz"))
    cfg = synth.medium_config()
    sd = synth.tracking_state_dict(cfg, int(f["seed"]))
    frames = synth.pan_stream(45, *cfg.img_size, pool=5, num=3, den=1, seed=2)
    assert int(frames.long().sum()) == int(f["frames_sum"]), "the regenerated frames are not the fixture's"
    return f, cfg, sd, frames


def test_oracle_loop_equals_the_reference_kffilter_plus_trackfrontend_run():
    """the per-frame loop of Hi2.run (hi2.py:101-111: kfFilter, then TrackFrontend.run) as the reference itself executed it on the CPU
    (`loop.npz`): which frames trigger which window (warm-up, steady state, the closing window of the last frame), the run_backend flag
    of every call, the keyframes, and the stores / edge list at the end."""
    f, cfg, sd, frames = load_loop_fixture()
    so = SR.SlamOracle(cfg, sd, cfg.img_size, 32, {"thresh": 0.9, "skip": 1, "kf_every": 2})
    n = frames.shape[0]
    calls = []
    for t in range(n):
        before = len(so.windows)
        so.kf_filter(t, frames[t], f["intrinsic"], second_last_frame=(t == n - 2), last_frame=(t == n - 1))
        flag = so.tracker_run(last_frame=(t == n - 1))
        if len(so.windows) != before:
            t0, t1, _ = so.windows[-1]
            calls.append([t, int(bool(flag)), t0, t1, t0 // 5])
    np.testing.assert_array_equal(np.asarray(calls), f["calls"])
    np.testing.assert_array_equal(so.tstamp[:so.counter].astype(np.int64), f["keyframes"])
    t1 = int(f["t1"])
    assert so.t1 == t1
    st = so.state
    nsub = (t1 - 1) // 5 + 1
    for name, got, ref in (("pose", st["pose"][:t1].numpy(), f["pose"]), ("depth mean", st["depth"][:t1].mean(dim=(1, 2)).numpy(), f["depth_mean"]),
                           ("submap_ds", st["submap_ds"][:nsub].numpy(), f["submap_ds"]), ("conf mean", st["conf_ds"][:nsub].mean(dim=(2, 3)).numpy(), f["conf_mean"])):
        err = float(np.abs(got - ref).max() / np.abs(ref).max())
        assert err < 3e-5, (name, err)
    ii, jj = so.graph.edges_numpy()[:2]
    np.testing.assert_array_equal(ii, f["ii"])
    np.testing.assert_array_equal(jj, f["jj"])


def load_backend_fixture(production=False):
    if production:
        from cut3r_slam_amd.config import production_config
        f = np.load(os.path.join(GOLD, "backend_production.npz"))
        cfg = production_config()
        sd = synth.loop_state_dict(cfg, int(f["seed"]), enc_residual_gain=0.1)
        frames = synth.pan_stream(40, 384, 512, pool=9, num=6, den=1, seed=0)
        assert int(frames.long().sum()) == int(f["frames_sum"]), "the regenerated frames are not the fixture's"
        return f, cfg, sd, frames
    f = np.load(os.path.join(GOLD, "backend.npz"))
    cfg = synth.medium_config()
    sd = synth.loop_state_dict(cfg, int(f["seed"]))
    frames = synth.pan_stream(90, *cfg.img_size, pool=5, num=2, den=1, seed=0)
    assert int(frames.long().sum()) == int(f["frames_sum"]), "the regenerated frames are not the fixture's"
    return f, cfg, sd, frames


def test_oracle_backend_equals_the_reference_trackbackend_up_to_the_optimiser():
    """TrackBackend.run of the reference, run on the CPU up to its optimiser call (`backend.npz`; the optimiser needs lietorch, absent from
    the reference tree): WHEN the backend first fires in the per-frame loop (every other eligible window, hi2.py:112-121), the keyframe
    its detect_loop scan stops at, the candidate list, the NMS choice, the six keyframes it re-tracks and the re-tracked submap, its
    confidences and poses (TrackBackend.track, :137-217) -- the inputs of loop_closure_init."""
    f, cfg, sd, frames = load_backend_fixture()
    mf = {"thresh": 0.9, "skip": 1, "kf_every": 2}
    t_fire = int(f["fired_at_frame"])
    # (a) the loop with the backend on closes its first loop at the same frame, on the same pair, from the same candidates
    so = SR.SlamOracle(cfg, sd, cfg.img_size, 64, mf, iteration=1)
    for t in range(t_fire + 1):
        so.run(t, frames[t], f["intrinsic"])
        assert len(so.closures) == (1 if t == t_fire else 0), t
    c = so.closures[0]
    assert (c["idx_current"], c["idx_matched"], c["at_keyframe"]) == (int(f["idx_current"]), int(f["idx_matched"]), len(f["keyframes"]))
    np.testing.assert_array_equal(np.sort(c["candidates"]), np.sort(f["candidates"]))
    np.testing.assert_array_equal(so.tstamp[:so.counter].astype(np.int64), f["keyframes"])
    np.testing.assert_array_equal(np.asarray([[a, b] for a, b, _ in so.windows]), f["windows"][:, 2:4])
    # (b) the pieces, on the state the reference had when it called the backend (no closure applied)
    so = SR.SlamOracle(cfg, sd, cfg.img_size, 64, mf, iteration=0)
    for t in range(t_fire + 1):
        so.run(t, frames[t], f["intrinsic"])
    np.testing.assert_allclose(so.state["pose"][:so.t1].numpy(), f["pose_before"], atol=2e-5)
    ii, jj = so.graph.edges_numpy()[:2]
    np.testing.assert_array_equal(ii, f["ii"])
    np.testing.assert_array_equal(jj, f["jj"])
    t1 = so.counter - 1
    scan = []
    for idx_current in range(t1 - 6, t1 - 1):
        ids = so.graph.detect_loop(idx_current)
        scan.append(idx_current)
        if ids is not None:
            break
    assert scan == f["scan_idx"].tolist() and sorted(np.asarray(ids).tolist()) == sorted(f["candidates"].tolist())
    K4 = (so.intrinsic[0] / np.float32(2)).astype(np.float32)
    scores = so.nms_scores(f["candidates"], idx_current, K4)
    assert float(scores.max()) > 0.4 and int(np.argmax(scores)) == int(f["k_th"])
    sel = f["selected_idx"].tolist()
    anchor = int(f["idx_matched"]) // 5
    assert sel == list(range(anchor * 5, anchor * 5 + 5)) + [idx_current] and anchor == int(f["anchor_sub_num"])
    pm, cf, ps = so.backend_track(sel, anchor)
    for name, got, ref in (("pointmaps_lc", pm.numpy(), f["pointmaps_lc"]), ("confs_lc", cf.numpy(), f["confs_lc"]), ("poses_lc", ps.numpy(), f["poses_lc"])):
        err = float(np.abs(got - ref).max() / np.abs(ref).max())
        assert err < 2e-5, (name, err)
    np.testing.assert_allclose(pm[-1:].numpy(), f["pointmap_current_lc"], atol=2e-5 * np.abs(f["pointmap_current_lc"]).max())
