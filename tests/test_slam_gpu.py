"""GPU: the tracking loop (keyframe filter -> window tracker -> covisibility graph) on HBM-resident buffers and HIP
kernels, checked against the CPU restatement of the reference loop (oracle/slam_oracle.py) on IDENTICAL network
outputs: keyframe store within fp32 tolerance, graph topology and keyframe decisions exact."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from cut3r_slam_amd.config import tiny_config  # noqa: E402
from cut3r_slam_amd.factor_graph import FactorGraph, SubmapStore  # noqa: E402
from cut3r_slam_amd.model import Cut3rModel  # noqa: E402
from cut3r_slam_amd.slam import Cut3rSlam  # noqa: E402
from cut3r_slam_amd.weights import synth_state_dict  # noqa: E402
from cut3r_slam_amd import geom_host as gh  # noqa: E402
from oracle import geom as G  # noqa: E402
from oracle import slam_oracle as SO  # noqa: E402

DEV = "cuda:0"
H, W = 32, 48


def _frames(n, seed=0):
    g = torch.Generator().manual_seed(seed)
    base = torch.rand(3, H + 2 * n, W + 2 * n, generator=g)
    base = torch.nn.functional.avg_pool2d(base[None], 5, 1, 2)[0]
    base = (base - base.min()) / (base.max() - base.min())
    return torch.stack([(base[:, t:t + H, 2 * t % n:2 * t % n + W] * 255).round().to(torch.uint8) for t in range(n)])


def _model():
    cfg = tiny_config("dpt")
    return Cut3rModel(cfg, synth_state_dict(cfg, 3), DEV, minimal=True)


def test_tracking_loop_matches_oracle_on_same_network_outputs():
    model = _model()
    cfgd = {"Tracking": {"motion_filter": {"thresh": 0.9, "skip": 1, "kf_every": 2}, "frontend": {"iteration": 0}}}
    n = 60
    slam = Cut3rSlam(model, cfgd, (H, W), buffer=40, device=DEV)
    captured = []
    real_infer = slam.tracker.infer

    def spy(*a, **k):
        out = real_infer(*a, **k)
        captured.append(tuple(o.detach().cpu().clone() for o in out))
        return out

    slam.tracker.infer = spy
    frames = _frames(n)
    intr = torch.tensor([40.0, 40.0, 23.5, 15.5])
    windows = []
    for t in range(n):
        before = slam.tracker.t1
        slam.run(t, frames[t:t + 1], intr, frames[t:t + 1], intr, last_frame=(t == n - 1))
        if slam.tracker.t1 != before:
            windows.append((0 if before == 0 else before - 1, slam.tracker.t1, before == 0))
    assert len(windows) >= 4 and len(windows) == len(captured)
    # ---- oracle replay on the captured outputs, view by view (graph calls interleave with the writes)
    nkf = slam.keyframes.counter.value
    S = nkf // 5 + 2
    K = np.array([[40.0, 0, 23.5], [0, 40.0, 15.5], [0, 0, 1]])
    st = {"pose": torch.zeros(nkf + 1, 7), "depth": torch.ones(nkf + 1, H, W),
          "submap_ds": torch.ones(S, 6, H // 2, W // 2, 3), "conf_ds": torch.zeros(S, 6, H // 2, W // 2)}
    st["pose"][:, 6] = 1
    graph = FactorGraph(None, device="cpu", max_factors=48, backend=SO.OracleOverlapBackend())
    for (t0, t1, init), (pts, conf, enc) in zip(windows, captured):
        if init:
            graph.add_neighborhood_factors(0, 3, r=3)
        full = SO.track_window({k: v.clone() for k, v in st.items()}, t0, t1, pts, conf, enc, init)
        tmp = {k: v.clone() for k, v in st.items()}
        SO.track_window(tmp, t0, t1, pts, conf, enc, init)
        for i in range(t0, t1):
            if not init:
                graph.add_neighborhood_factors(i - 3, i + 1, r=3)
            v = i - t0
            sub = t0 // 5
            st["submap_ds"][sub, v] = tmp["submap_ds"][sub, v]
            st["conf_ds"][sub, v] = tmp["conf_ds"][sub, v]
            st["pose"][i] = tmp["pose"][i]
            st["depth"][i] = tmp["depth"][i]
            if i > 2:
                all_c2w = SO.pose_vec_to_matrix(st["pose"][:i])
                cur_c2w = SO.pose_vec_to_matrix(st["pose"][i][None])[0]
                graph.add(i, all_c2w, SubmapStore(st["submap_ds"], i), cur_c2w, full[v][2], K)
    kf = slam.keyframes
    ntr = slam.tracker.t1
    np.testing.assert_allclose(kf.pose[:ntr].numpy(), st["pose"][:ntr].numpy(), atol=2e-5)
    np.testing.assert_allclose(kf.depth[:ntr].cpu().numpy(), st["depth"][:ntr].numpy(), rtol=2e-5, atol=1e-6)
    nsub = (ntr - 1) // 5 + 1
    got_pm, ref_pm = kf.submap_ds[:nsub].cpu().numpy(), st["submap_ds"][:nsub].numpy()
    scale = np.abs(ref_pm).max()
    assert np.abs(got_pm - ref_pm).max() <= 3e-5 * scale
    np.testing.assert_allclose(kf.conf_ds[:nsub].cpu().numpy(), st["conf_ds"][:nsub].numpy(), rtol=1e-5, atol=1e-6)
    ii, jj, age = slam.graph.edges_numpy()
    rii, rjj, rage = graph.edges_numpy()
    np.testing.assert_array_equal(ii, rii)
    np.testing.assert_array_equal(jj, rjj)
    np.testing.assert_array_equal(age, rage)
    assert len(ii) > 4 * ntr          # neighbourhood + covisibility edges were really added


def test_keyframe_filter_overlap_mode_decisions_match_oracle():
    model = _model()
    cfgd = {"Tracking": {"motion_filter": {"thresh": 0.9, "skip": 2, "kf_every": -1}, "frontend": {"iteration": 0}}}
    slam = Cut3rSlam(model, cfgd, (H, W), buffer=64, device=DEV)
    frames = _frames(40, seed=1)
    g = torch.Generator().manual_seed(7)
    frames[20:] = torch.randint(0, 256, frames[20:].shape, generator=g, dtype=torch.uint8)   # scene cut: unrelated images
    f = slam.filterx
    decisions, ref_dec = [], []
    for t in range(40):
        feat_last = slam.keyframes.featI[slam.keyframes.counter.value - 1].cpu().numpy() if slam.keyframes.counter.value else None
        took = f.kfFilter(t, frames[t:t + 1], intrinsics=torch.ones(4))
        decisions.append(took)
        if t == 0:
            ref_dec.append(True)
        elif t % 2 == 0:
            feat1, _ = f.encode(frames[t:t + 1])
            ratio, mx = G.patch_overlap_ratio(feat_last, feat1.cpu().numpy())
            assert np.abs(mx - 0.7).min() > 1e-5          # no razor-edge rows in this fixture
            ref_dec.append(ratio < 0.9)
        else:
            ref_dec.append(False)
    assert decisions == ref_dec
    assert 1 <= sum(decisions) < 40
    print("keyframes taken at", [t for t, d in enumerate(decisions) if d])


def test_window_batch_gives_the_same_trajectory_and_graph():
    """window_batch=3 (batched decoder inference of 3 windows) vs the reference schedule (one window at a time)."""
    frames = _frames(80, seed=2)
    intr = torch.tensor([40.0, 40.0, 23.5, 15.5])
    out = []
    for wb in (1, 3):
        model = _model()
        cfgd = {"Tracking": {"motion_filter": {"thresh": 0.9, "skip": 1, "kf_every": 2}, "frontend": {"iteration": 0, "window_batch": wb}}}
        slam = Cut3rSlam(model, cfgd, (H, W), buffer=48, device=DEV)
        for t in range(80):
            slam.run(t, frames[t:t + 1], intr, frames[t:t + 1], intr, last_frame=(t == 79))
        n = slam.tracker.t1
        out.append((n, slam.keyframes.pose[:n].numpy().copy(), slam.graph.edges_numpy(), slam.keyframes.depth[:n].cpu().numpy()))
    assert out[0][0] == out[1][0] and out[0][0] > 20
    np.testing.assert_allclose(out[0][1], out[1][1], atol=1e-5)
    np.testing.assert_allclose(out[0][3], out[1][3], rtol=1e-4, atol=1e-5)
    for a, b in zip(out[0][2], out[1][2]):
        np.testing.assert_array_equal(a, b)


def test_predict_matches_oracle_on_same_network_outputs():
    """TrackFrontend.predict (track_frontend.py:102-162) against its CPU restatement on identical 2-view outputs"""
    frames = _frames(8, seed=7)
    intr = torch.tensor([40.0, 40.0, 23.5, 15.5])
    model = _model()
    cfgd = {"Tracking": {"motion_filter": {"thresh": 0.9, "skip": 1, "kf_every": 2}, "frontend": {"iteration": 0}}}
    slam = Cut3rSlam(model, cfgd, (H, W), buffer=16, device=DEV)
    tr = slam.tracker
    outs = tuple(t.clone() for t in tr.infer(torch.stack([frames[0], frames[3]], 0).to(DEV)))
    g = torch.Generator().manual_seed(1)
    kf_pose = torch.cat([torch.randn(3, generator=g) * 0.3, torch.nn.functional.normalize(torch.randn(4, generator=g), dim=0)])
    kf_depth = outs[0][0, ..., 2].abs().cpu() * 1.7 + 0.05
    new_pose, new_depth, new_pm, new_conf = tr.predict(frames[3], frames[0], kf_pose, kf_depth, outputs=outs)
    ref_pose, ref_depth, ref_pm, ref_conf = SO.predict(outs[0].cpu(), outs[1].cpu(), outs[2].cpu(), kf_pose, kf_depth)
    torch.cuda.synchronize()
    np.testing.assert_allclose(new_pose.numpy(), ref_pose.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(new_depth.cpu().numpy(), ref_depth.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(new_pm.cpu().numpy(), ref_pm.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_array_equal(new_conf.cpu().numpy(), ref_conf.numpy())
    # and the inference path itself (no precomputed outputs) runs and agrees with it
    p2, d2, m2, c2 = tr.predict(frames[3], frames[0], kf_pose, kf_depth)
    np.testing.assert_allclose(p2.numpy(), new_pose.numpy(), rtol=1e-5, atol=1e-6)


def test_pipelined_driver_equals_the_frame_by_frame_loop():
    """dist.ShardedTracker (window batch 3, encoder look-ahead on its own stream, replay on the side stream behind the
    encoder graph, decoder graph last) must leave the same keyframe poses, depths and ordered edge lists as feeding the
    frames one by one through Cut3rSlam.run with the reference schedule."""
    from cut3r_slam_amd import dist as cdist
    kf_every, win, wb, steps = 2, 5, 3, 2
    n = (7 + win * wb * steps + 1) * kf_every + 1
    frames = _frames(n, seed=4).to(DEV)
    intr = torch.tensor([40.0, 40.0, 23.5, 15.5])
    cfgd = {"Tracking": {"motion_filter": {"thresh": 0.9, "skip": 1, "kf_every": kf_every}, "frontend": {"iteration": 0}}}
    # reference schedule: one frame at a time, one window at a time
    ref = Cut3rSlam(_model(), cfgd, (H, W), buffer=64, device=DEV)
    for t in range(n - 1):
        ref.run(t, frames[t:t + 1], intr, frames[t:t + 1], intr)
    # pipelined driver
    slam = Cut3rSlam(_model(), cfgd, (H, W), buffer=64, device=DEV)
    t = 0
    while not slam.keyframes.is_initialized:
        slam.run(t, frames[t:t + 1], intr, frames[t:t + 1], intr)
        t += 1
    runner = cdist.ShardedTracker(slam, 1, 0, wb=wb, pipelined=True)
    assert frames.shape[0] >= runner.frames_needed(steps, kf_every, win)
    for _ in range(steps):
        t = runner.step(frames, t, kf_every, win, intr)
    runner.flush()
    torch.cuda.synchronize()
    k = slam.tracker.t1
    assert k == 6 + win * wb * steps and ref.tracker.t1 >= k
    np.testing.assert_allclose(slam.keyframes.pose[:k].numpy(), ref.keyframes.pose[:k].numpy(), atol=1e-5)
    np.testing.assert_allclose(slam.keyframes.depth[:k].cpu().numpy(), ref.keyframes.depth[:k].cpu().numpy(), rtol=1e-4, atol=1e-5)
    e_ref = [np.asarray(x) for x in ref.graph.edges_numpy()]
    keep = (e_ref[0] < k) & (e_ref[1] < k)                    # the frame-by-frame loop may already have tracked one window more
    e_got = slam.graph.edges_numpy()
    np.testing.assert_array_equal(e_got[0], e_ref[0][keep])
    np.testing.assert_array_equal(e_got[1], e_ref[1][keep])


@pytest.mark.parametrize("driver", ["frame_by_frame_wb2", "pipelined_wb2", "pipelined_scan_wb2"])
def test_sequence_cuts_equal_fresh_runs(driver):
    """TrackFrontend.sequence_windows = 3: the stream is cut every 3 windows (15 keyframes); the cut keyframe starts a new
    sequence whose first window is an initialisation window, with an empty covisibility graph and overlap tests that only see
    the sequence's own keyframes.  Every sequence must equal a FRESH Cut3rSlam run over the frames from its cut keyframe on:
    poses, depths, stride-2 stores, ordered edge lists (relative numbering).  Cuts fall on the first and on the second window
    of a decoder batch; the pipelined driver is checked in both replay forms."""
    from cut3r_slam_amd import dist as cdist
    kf_every, S, nseq = 2, 3, 3
    per = 5 * S                                               # keyframes per sequence (the cut keyframe is shared)
    n = 2 * (per * nseq + 1 + 6) + 12
    frames = _frames(n, seed=7).to(DEV)
    intr = torch.tensor([40.0, 40.0, 23.5, 15.5])
    fe = {"iteration": 0, "sequence_windows": S}
    cfgd = {"Tracking": {"motion_filter": {"thresh": 0.9, "skip": 1, "kf_every": kf_every}, "frontend": dict(fe)}}
    if driver == "frame_by_frame_wb2":
        cfgd["Tracking"]["frontend"]["window_batch"] = 2
        slam = Cut3rSlam(_model(), cfgd, (H, W), buffer=80, device=DEV)
        for t in range(n):
            slam.run(t, frames[t:t + 1], intr, frames[t:t + 1], intr)
            if slam.tracker.t1 >= per * nseq + 1:
                break
    else:
        slam = Cut3rSlam(_model(), cfgd, (H, W), buffer=80, device=DEV)
        t = 0
        while not slam.keyframes.is_initialized:
            slam.run(t, frames[t:t + 1], intr, frames[t:t + 1], intr)
            t += 1
        runner = cdist.ShardedTracker(slam, 1, 0, wb=2, pipelined=True)
        runner.scan = driver == "pipelined_scan_wb2"
        for _ in range(4):
            t = runner.step(frames, t, kf_every, 5, intr)
        runner.flush()
    torch.cuda.synchronize()
    k = slam.tracker.t1
    assert k >= per * (nseq - 1) + 11, k                      # at least two windows into the third sequence
    kf, g = slam.keyframes, slam.graph
    assert len(g.closed) == nseq - 1 and g.base == per * (nseq - 1)
    tol = dict(atol=1e-5) if driver != "pipelined_scan_wb2" else dict(atol=2e-4)      # (scan form: log s + log d vs log(s d))
    for sq in range(nseq):
        b0 = per * sq
        last = min(k, b0 + per + 1) - b0                      # keyframes of this sequence tracked so far (relative end)
        fresh = Cut3rSlam(_model(), {"Tracking": {"motion_filter": cfgd["Tracking"]["motion_filter"], "frontend": {"iteration": 0}}},
                          (H, W), buffer=40, device=DEV)
        t = 0
        while fresh.tracker.t1 < last and 2 * b0 + t < n:
            fresh.run(t, frames[2 * b0 + t:2 * b0 + t + 1], intr, frames[2 * b0 + t:2 * b0 + t + 1], intr)
            t += 1
        assert fresh.tracker.t1 == last, (sq, fresh.tracker.t1, last)
        own = last - 1 if (sq < nseq - 1) else last           # the cut keyframe's pose / depth rows belong to the NEXT sequence afterwards
        np.testing.assert_allclose(kf.pose[b0:b0 + own].numpy(), fresh.keyframes.pose[:own].numpy(), err_msg=f"pose seq {sq}", **tol)
        np.testing.assert_allclose(kf.depth[b0:b0 + own].cpu().numpy(), fresh.keyframes.depth[:own].cpu().numpy(), rtol=2e-4, atol=2e-5,
                                   err_msg=f"depth seq {sq}")
        nsub = (last - 1) // 5
        np.testing.assert_allclose(kf.submap_ds[b0 // 5:b0 // 5 + nsub].cpu().numpy(), fresh.keyframes.submap_ds[:nsub].cpu().numpy(),
                                   rtol=2e-4, atol=2e-4 if "scan" in driver else 2e-5, err_msg=f"submaps seq {sq}")
        if sq < nseq - 1:
            _, ii, jj, _ = g.closed[sq]
            ii, jj = ii - b0, jj - b0
        else:
            ii, jj, _ = g.edges_numpy()
        fi, fj, _ = fresh.graph.edges_numpy()
        np.testing.assert_array_equal(ii, fi, err_msg=f"edges seq {sq}")
        np.testing.assert_array_equal(jj, fj, err_msg=f"edges seq {sq}")
        assert len(ii) > 20
    ai, aj = g.edges_absolute()
    assert len(ai) == sum(len(c[1]) for c in g.closed) + len(g.edges_numpy()[0]) and ai.max() < k


def test_sharded_overlap_counting_two_ranks_in_one_process():
    """Multi-GPU replay (dist.ShardedTracker): every rank chains and stores every window, the overlap counting of a window
    runs only on its owner and the owners' counts are summed before the decisions.  Two trackers play the two ranks here
    (the sum stands in for the all-reduce); poses, stores and the ordered edge lists must equal the single-rank replay."""
    frames = _frames(80, seed=5)
    intr = torch.tensor([40.0, 40.0, 23.5, 15.5])
    model = _model()
    cfgd = {"Tracking": {"motion_filter": {"thresh": 0.9, "skip": 1, "kf_every": 2}, "frontend": {"iteration": 0}}}
    slams = [Cut3rSlam(model, cfgd, (H, W), buffer=48, device=DEV) for _ in range(3)]
    nwin = 4
    for slam in slams:
        t = 0
        while not slam.keyframes.is_initialized:
            slam.run(t, frames[t:t + 1], intr, frames[t:t + 1], intr)
            t += 1
        while slam.keyframes.counter.value < slam.tracker.t1 + 5 * nwin:
            if t % 2 == 0:
                slam.keyframes.append(t, frames[t], None, None, None, None, intr, None, None)
            t += 1
    first = slams[0].tracker.t1 - 1
    ranges = [(first + 5 * j, first + 5 * j + 6) for j in range(nwin)]
    tr0 = slams[0].tracker
    tr0.window_features(ranges[0][0], ranges[-1][1])
    feats = torch.stack([tr0.window_features(a, b) for a, b in ranges], 0)
    res = model.decode_windows(feats, H, W)
    outs = [tuple(res[k][6 * j:6 * j + 6].clone() for k in ("pts3d_in_self_view", "conf_self", "camera_pose")) for j in range(nwin)]
    tr0.track_many(ranges, outs)
    trA, trB = slams[1].tracker, slams[2].tracker
    maskA = [j % 2 == 0 for j in range(nwin)]
    maskB = [not m for m in maskA]

    def exchange_a(counts_a):
        box = {}

        def exchange_b(counts_b):
            assert int((counts_a != 0).sum()) > 0 and int((counts_b != 0).sum()) > 0
            assert int(((counts_a != 0) & (counts_b != 0)).sum()) == 0          # disjoint ownership
            box["total"] = counts_a + counts_b
            return box["total"]
        trB.track_many(ranges, outs, count_mask=maskB, exchange=exchange_b)
        return box["total"]

    trA.track_many(ranges, outs, count_mask=maskA, exchange=exchange_a)
    torch.cuda.synchronize()
    n = ranges[-1][1]
    e0 = slams[0].graph.edges_numpy()
    assert len(e0[0]) > 40
    for slam in slams[1:]:
        np.testing.assert_array_equal(slam.keyframes.pose[:n].numpy(), slams[0].keyframes.pose[:n].numpy())
        assert torch.equal(slam.keyframes.submap_ds, slams[0].keyframes.submap_ds)
        assert torch.equal(slam.keyframes.depth[:n], slams[0].keyframes.depth[:n])
        assert torch.equal(slam.keyframes.w2c[:n], slams[0].keyframes.w2c[:n])
        for a, b in zip(slam.graph.edges_numpy(), e0):
            np.testing.assert_array_equal(a, b)


def test_loop_closure_backend_runs_end_to_end_and_reduces_the_disagreement():
    """TrackBackend.run(): detection -> NMS -> re-tracking -> fused optimiser -> in-place rewrite, on a stream whose second
    half replays the first (so late keyframes are covisible with early ones).  Functional check (the tiny random network
    carries no geometry): the backend must run, return the reference's update dict and keep poses finite; the optimiser's
    loss must not increase."""
    model = _model()
    cfgd = {"Tracking": {"motion_filter": {"thresh": 0.9, "skip": 1, "kf_every": 2},
                         "frontend": {"iteration": 60, "window_batch": 1}}}
    slam = Cut3rSlam(model, cfgd, (H, W), buffer=64, device=DEV)
    base = _frames(30, seed=3)
    frames = torch.cat([base, base.flip(0), base], 0)            # forward, backward, forward: revisits
    intr = torch.tensor([40.0, 40.0, 23.5, 15.5])
    did = []
    for t in range(len(frames)):
        _, _, lc = slam.run(t, frames[t:t + 1], intr, frames[t:t + 1], intr, last_frame=(t == len(frames) - 1))
        did.append(bool(lc))
    assert np.isfinite(slam.keyframes.pose[:slam.tracker.t1].numpy()).all()
    # force one closure through the public pieces even if the random network produced no covisible revisit
    be = slam.backend
    n = slam.tracker.t1
    idx_current, idx_matched = n - 3, 2
    pm_lc, conf_lc, poses_lc = be.track(list(range(0, 5)) + [idx_current], 0)
    assert pm_lc.shape == (6, H // 2, W // 2, 3) and np.isfinite(poses_lc).all()
    before = slam.keyframes.submap_ds[: idx_current // 5 + 1].clone()
    upd = be.loop_closure_init(pm_lc[-1], idx_matched, idx_current, return_loss=True)
    torch.cuda.synchronize()
    loss = upd["loss"].cpu().numpy()
    assert np.isfinite(loss).all() and loss[-1] <= loss[0] * 1.0001
    assert set(upd) >= {"pose_updates", "submap_idx", "camera_idx", "camera_pose"}
    B = idx_current // 5 + 1
    assert upd["pose_updates"].shape == (B, 7) and upd["camera_pose"].shape[1] == 7
    after = slam.keyframes.submap_ds[:B]
    assert torch.equal(after[0], before[0])                       # the first submap is the fixed gauge (T_0 = I)
    assert torch.isfinite(after).all()
    # NMS path executes with the HIP overlap + feature kernels
    k = be.nms(np.array([0, 1, 2]), idx_current, [20.0, 20.0, 11.75, 7.75])
    assert k is None or 0 <= k < 3


def test_terminate_add_kf_densifies_wide_keyframe_gaps():
    """Hi2.terminate(add_kf=True) (hi2.py:177-214; demo_s.py:171): every pair of consecutive keyframes more than 30 frames apart
    gets one extra view at the middle frame, relocalised by TrackFrontend.predict against the earlier keyframe -- same
    frames, same outputs as calling predict directly; without kept frames it fails loudly."""
    model = _model()
    cfgd = {"Tracking": {"motion_filter": {"thresh": 0.9, "skip": 1, "kf_every": 40}, "frontend": {"iteration": 0}}}
    slam = Cut3rSlam(model, cfgd, (H, W), buffer=24, device=DEV)
    frames = _frames(330, seed=5)
    intr = torch.tensor([40.0, 40.0, 23.5, 15.5])
    with pytest.raises(RuntimeError):
        slam.terminate(add_kf=True)
    slam.keep_images = True
    n = len(frames)
    for t in range(n):
        slam.run(t, frames[t:t + 1], intr, frames[t:t + 1], intr, second_last_frame=(t == n - 2), last_frame=(t == n - 1))
    kf = slam.keyframes
    traj, views = slam.terminate(add_kf=True)
    ts = kf.tstamp[:kf.counter.value - 1].numpy()
    gaps = [i for i in range(len(ts) - 1) if ts[i + 1] - ts[i] > 30]
    assert len(gaps) >= 6 and len(views) == len(gaps)
    assert traj.shape == (kf.buffer, 7)
    for i, v in zip(gaps, views):
        assert v["tstamp"] == int(ts[i] + (ts[i + 1] - ts[i]) // 2) and v["submap"] == i // 5
        pose, depth, pm, conf = slam.tracker.predict(frames[v["tstamp"]].to(DEV), kf.image[i], kf.pose[i], kf.depth[i])
        assert torch.equal(pose, v["pose"]) and torch.equal(depth, v["depth"]) and torch.equal(pm, v["pointmap"]) and torch.equal(conf, v["conf"])
        assert torch.isfinite(v["pose"]).all() and v["pointmap"].shape == (H // 2, W // 2, 3)
    _, none = slam.terminate(add_kf=False)
    assert none == []


@pytest.mark.parametrize("n,skip,lookahead,wb,mode", [
    (83, 2, 7, 2, "buffered"), (83, 2, 7, 2, "pipeline"), (83, 2, 40, 3, "pipeline"), (61, 3, 4, 1, "pipeline"),
    (83, 2, 5, 2, "stream"), (47, 1, 9, 2, "stream"), (83, 2, 1, 2, "stream")])
def test_overlap_mode_lookahead_drivers_equal_the_frame_by_frame_loop(n, skip, lookahead, wb, mode):
    """Overlap mode (kf_every = -1) through every look-ahead driver -- run_buffered, run_buffered(pipeline=True: the next chunk's encoder pass +
    decision chain on a side stream), run_stream (an item iterator) -- with window batches, chunk sizes that do and do not divide the stream,
    a look-ahead longer than the stream's tested frames, and the always-kept second-last / last frames: keyframes, poses, depths and ordered
    edges are those of the frame-by-frame loop, bit for bit."""
    from cut3r_slam_amd import synth
    frames = synth.slideshow_stream(n, H, W, hold=3, seed=n).to(DEV)      # a new texture every third frame: some tested frames are keyframes, some are not
    intr = torch.tensor([40.0, 40.0, 23.5, 15.5])
    cfg = tiny_config("dpt")
    model = Cut3rModel(cfg, synth.tracking_state_dict(cfg, 3, enc_residual_gain=0.1), DEV, minimal=True)
    mf = {"thresh": 0.9, "skip": skip, "kf_every": -1}
    out = []
    for drv in ("plain", mode):
        cfgd = {"Tracking": {"motion_filter": dict(mf), "frontend": {"iteration": 0, "window_batch": 1 if drv == "plain" else wb}}}
        slam = Cut3rSlam(model, cfgd, (H, W), buffer=n + 8, device=DEV)
        if drv == "plain":
            for t in range(n):
                slam.run(t, frames[t:t + 1], intr, frames[t:t + 1], intr, second_last_frame=(t == n - 2), last_frame=(t == n - 1))
        elif drv == "stream":
            slam.run_stream(((t, frames[t:t + 1], intr, frames[t:t + 1], intr, t == n - 2, t == n - 1) for t in range(n)), lookahead=lookahead)
        else:
            slam.run_buffered(frames, intr, lookahead=lookahead, pipeline=(drv == "pipeline"))
        torch.cuda.synchronize()
        k = slam.tracker.t1
        out.append((k, slam.keyframes.counter.value, slam.keyframes.tstamp[:slam.keyframes.counter.value].clone(), slam.keyframes.pose[:k].clone(),
                    slam.keyframes.depth[:k].clone(), slam.graph.edges_numpy(), dict(slam.filterx.stats)))
    a, b = out
    assert a[0] == b[0] and a[1] == b[1] and a[1] >= 8, (a[0], b[0], a[1], b[1])
    assert torch.equal(a[2], b[2]) and torch.equal(a[3], b[3]) and torch.equal(a[4], b[4])
    for x, y in zip(a[5], b[5]):
        np.testing.assert_array_equal(x, y)
    if lookahead > 1:
        assert b[6]["encoded"] <= 2, b[6]          # (everything else came from the batched look-ahead passes)
