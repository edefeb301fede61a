"""GPU: END-TO-END trajectory parity -- frames in, keyframe trajectory out -- of the HIP tracking loop (`Cut3rSlam.run`,
everything through the C ABI) against the CPU restatement of the reference loop (oracle/slam_run.py: fp32 network, fp32
geometry) on the SAME seeded stream and weights.  This is the second half of the BASELINE metric: the reference's run
scripts score `traj_kf.txt` with `evo_ape tum ... -vas` (scripts/run_scannet.py:34-36) = Sim(3)-aligned ATE-RMSE.

What is asserted (medium config: production head widths at 64x96, >= 7 tracking windows, path length ~3 m):
  * keyframe selection: identical time stamps (fixed cadence AND overlap mode, whose decisions come from the features);
  * graph topology: identical edge lists, except edges whose deciding overlap ratio sits within +-0.02 of the 0.3 threshold in
    the oracle (listed; a TF32 run of the oracle flips the same kind of edges against its own fp32 run);
  * ATE-RMSE(GPU, CPU fp32) <= 1 mm (BASELINE target) on the smooth 2.9 m fixed-cadence stream, <= 1 mm per metre on the
    13.7 m slideshow stream, and in both cases <= 2.5 x ATE-RMSE(CPU TF32-emulated, CPU fp32): the fp16-operand MFMA path
    deviates from exact fp32 no more than the reference's own TF32 arithmetic does.
Parity status: the network, overlap counts and edge bookkeeping of the oracle are pinned to reference fixtures; the
composition of the tracker drivers is read-faithful, unpinned (they hard-code 'cuda' in the reference).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from cut3r_slam_amd import synth  # noqa: E402
from cut3r_slam_amd.eval_ate import ate_rmse  # noqa: E402
from cut3r_slam_amd.model import Cut3rModel  # noqa: E402
from cut3r_slam_amd.slam import Cut3rSlam  # noqa: E402
from oracle import slam_run as SR  # noqa: E402

DEV = "cuda:0"
H, W = 64, 96
INTR = np.array([80.0, 80.0, 47.5, 31.5], np.float32)
NEAR = 0.02                     # half-width of the "near the 0.3 overlap threshold" band used to explain edge differences


def _gpu_run(cfg, sd, frames, mf, buffered=False, window_batch=1, streamed=0):
    model = Cut3rModel(cfg, sd, DEV, minimal=True)
    conf = {"Tracking": {"motion_filter": dict(mf), "frontend": {"iteration": 0, "window_batch": window_batch}}}
    slam = Cut3rSlam(model, conf, (H, W), buffer=frames.shape[0] + 8, device=DEV)
    intr = torch.from_numpy(INTR)
    fr = frames.to(DEV)
    n = fr.shape[0]
    if streamed:                               # demo.py --lookahead: an iterator of per-frame items, `streamed` tested frames held back
        slam.run_stream(((t, fr[t:t + 1], intr, fr[t:t + 1], intr, t == n - 2, t == n - 1) for t in range(n)), lookahead=streamed)
    elif buffered:
        slam.run_buffered(fr, intr, lookahead=6, pipeline=(buffered == "pipeline"))
    else:
        for t in range(n):
            slam.run(t, fr[t:t + 1], intr, fr[t:t + 1], intr, second_last_frame=(t == n - 2), last_frame=(t == n - 1))
    torch.cuda.synchronize()
    ts, poses = slam.trajectory()
    return slam, np.concatenate([ts.reshape(-1, 1).astype(np.float64), poses.astype(np.float64)], 1)


def _path_length(traj):
    return float(np.linalg.norm(np.diff(traj[:, 1:4], axis=0), axis=1).sum())


def _explain_edge_differences(edges_a, edges_b, so):
    """every edge in the symmetric difference must be decided by a ratio within NEAR of 0.3 in the oracle"""
    unexplained, near = [], []
    for (i, j) in sorted(set(edges_a) ^ set(edges_b)):
        r = so.graph.ratios.get((max(i, j), min(i, j)))
        ok = r is not None and any(v is not None and abs(v - 0.3) <= NEAR for v in r)
        (near if ok else unexplained).append(((i, j), r))
    return near, unexplained


def _compare(tag, slam, traj_gpu, so_fp32, so_tf32, ate_limit=1e-3):
    traj_ref, traj_tf = so_fp32.trajectory(), so_tf32.trajectory()
    assert np.array_equal(traj_gpu[:, 0], traj_ref[:, 0]), f"{tag}: keyframe time stamps differ\n{traj_gpu[:, 0]}\n{traj_ref[:, 0]}"
    assert len(so_fp32.windows) >= 7 and slam.tracker.t1 == so_fp32.t1
    path = _path_length(traj_ref)
    ate = ate_rmse(traj_gpu, traj_ref, 0.01, True)
    ate_tf = ate_rmse(traj_tf, traj_ref, 0.01, True) if np.array_equal(traj_tf[:, 0], traj_ref[:, 0]) else None
    ii, jj, _ = slam.graph.edges_numpy()
    e_gpu, e_ref = list(zip(ii.tolist(), jj.tolist())), list(zip(so_fp32.graph.ii, so_fp32.graph.jj))
    near, unexplained = _explain_edge_differences(e_gpu, e_ref, so_fp32)
    first_div = next((k for k, (a, b) in enumerate(zip(e_gpu, e_ref)) if a != b), None)
    pos_err = np.abs(traj_gpu[:, 1:4] - traj_ref[:, 1:4]).max()
    print(f"[e2e {tag}] keyframes {len(traj_ref)} windows {len(so_fp32.windows)} path {path:.3f} m | ATE-RMSE GPU vs CPU-fp32 "
          f"{ate['rmse'] * 1e3:.3f} mm (scale {ate['scale']:.6f}, max {ate['max'] * 1e3:.3f} mm, unaligned max |dt| {pos_err * 1e3:.3f} mm) | "
          f"CPU-tf32 vs CPU-fp32 {ate_tf['rmse'] * 1e3 if ate_tf else float('nan'):.3f} mm | edges {len(e_gpu)} vs {len(e_ref)}, "
          f"first divergent edge {first_div}, near-threshold differences {near}")
    assert not unexplained, f"{tag}: edge differences not explained by a near-threshold overlap ratio: {unexplained}"
    assert len(near) <= 0.02 * len(e_ref) + 2
    assert path >= 1.0
    assert ate["rmse"] <= ate_limit(path) if callable(ate_limit) else ate["rmse"] <= ate_limit, ate
    if ate_tf is not None:
        assert ate["rmse"] <= 2.5 * ate_tf["rmse"] + 5e-5, (ate, ate_tf)
    return ate, ate_tf


def test_fixed_cadence_stream_trajectory_matches_cpu_path():
    cfg = synth.medium_config()
    sd = synth.tracking_state_dict(cfg, 11)
    mf = {"thresh": 0.9, "skip": 1, "kf_every": 2}
    frames = synth.pan_stream(70, H, W, pool=5, num=2, den=1, seed=0)
    so32 = SR.run_stream(cfg, sd, frames, INTR, mf, precision="fp32")
    sotf = SR.run_stream(cfg, sd, frames, INTR, mf, precision="tf32")
    slam, traj = _gpu_run(cfg, sd, frames, mf)
    _compare("kf_every=2", slam, traj, so32, sotf)


def test_overlap_mode_stream_trajectory_matches_cpu_path_and_buffered_driver_is_identical():
    cfg = synth.medium_config()
    sd = synth.tracking_state_dict(cfg, 11)
    mf = {"thresh": 0.9, "skip": 2, "kf_every": -1}
    frames = synth.slideshow_stream(150, H, W, hold=4, seed=3)
    so32 = SR.run_stream(cfg, sd, frames, INTR, mf, precision="fp32")
    sotf = SR.run_stream(cfg, sd, frames, INTR, mf, precision="tf32")
    # the decisions are content driven: some tested frames are kept, some are not
    kept = {int(t) for t in so32.trajectory()[:, 0]}
    tested = [t for t, _ in so32.ratios]
    assert 0.3 < len(kept & set(tested)) / len(tested) < 0.7
    slam, traj = _gpu_run(cfg, sd, frames, mf)
    # ratios of the HIP filter vs the oracle's on every tested frame: both far from the 0.9 decision threshold
    # (unrelated textures make the random-weight network jump ~0.35 m per keyframe: a 13.7 m path, on which the reference's
    #  own TF32 arithmetic already deviates 3.2 mm from exact fp32 -- the bound here is 1 mm per metre of path)
    _compare("overlap skip=2", slam, traj, so32, sotf, ate_limit=lambda path: 1e-3 * path)
    # buffered driver (batched look-ahead encode + on-device decision chain): bit-identical to the frame-by-frame loop
    slam_b, traj_b = _gpu_run(cfg, sd, frames, mf, buffered=True)
    assert np.array_equal(traj, traj_b)
    assert slam_b.filterx.stats["encoded"] <= 2 and slam_b.filterx.stats["cache_hits"] >= len(tested) - 2
    for a, b in zip(slam.graph.edges_numpy(), slam_b.graph.edges_numpy()):
        assert np.array_equal(a, b)
    k = slam.tracker.t1
    assert torch.equal(slam.keyframes.featI[:k], slam_b.keyframes.featI[:k])
    assert torch.equal(slam.keyframes.depth[:k], slam_b.keyframes.depth[:k])
    # ... and with the next chunk's encoder pass + decision chain running on a side stream beside the current chunk's windows
    slam_p, traj_p = _gpu_run(cfg, sd, frames, mf, buffered="pipeline", window_batch=2)
    assert np.array_equal(traj, traj_p) and slam_p.tracker.t1 == k
    assert slam_p.filterx.stats["encoded"] <= 2 and slam_p.filterx.stats["cache_hits"] >= len(tested) - 2
    for a, b in zip(slam.graph.edges_numpy(), slam_p.graph.edges_numpy()):
        assert np.array_equal(a, b)
    assert torch.equal(slam.keyframes.featI[:k], slam_p.keyframes.featI[:k])
    assert torch.equal(slam.keyframes.depth[:k], slam_p.keyframes.depth[:k])
    # the reference's own mode at buffered throughput (demo.py --lookahead N --window-batch W): an item iterator with 12 tested frames held
    # back, the keyframes they yield tracked two windows at a time -- the same keyframes, poses, depths and ordered edges, bit for bit
    slam_s, traj_s = _gpu_run(cfg, sd, frames, mf, streamed=12, window_batch=2)
    assert np.array_equal(traj, traj_s)
    assert slam_s.filterx.stats["encoded"] <= 2 and slam_s.tracker.t1 == k
    for a, b in zip(slam.graph.edges_numpy(), slam_s.graph.edges_numpy()):
        assert np.array_equal(a, b)
    assert torch.equal(slam.keyframes.depth[:k], slam_s.keyframes.depth[:k])
    assert torch.equal(slam.keyframes.submap_ds[:(k - 1) // 5], slam_s.keyframes.submap_ds[:(k - 1) // 5])


def test_overlap_decisions_at_production_encoder_shape_match_oracle_on_same_features():
    """384x512, ViT-L encoder (24 x 1024/16): the keyframe decisions of the batched look-ahead path == decisions taken by the
    fp64 oracle ratio (oracle/geom.py) on the SAME (HIP) features, and == the sequential kfFilter path."""
    from cut3r_slam_amd.config import production_config
    from oracle import geom as G
    cfg = production_config()
    sd = synth.tracking_state_dict(cfg, 0, enc_residual_gain=0.1)
    model = Cut3rModel(cfg, sd, DEV, minimal=True)
    Hp, Wp = 384, 512
    frames = synth.slideshow_stream(61, Hp, Wp, hold=10, seed=1, device=DEV)
    conf = {"Tracking": {"motion_filter": {"thresh": 0.9, "skip": 5, "kf_every": -1}, "frontend": {"iteration": 0}}}
    intr = torch.tensor([300.0, 300.0, 255.5, 191.5])
    slam = Cut3rSlam(model, conf, (Hp, Wp), buffer=32, device=DEV)
    f = slam.filterx
    idx = list(range(0, 61, 5))
    f.prefetch(frames[0:61:5], idx, None)
    ahead = {t: (feat.clone(), took, cnt) for t, (feat, took, cnt, _) in f._ahead.items()}
    # sequential reference decisions on the same features, fp64 ratio
    last, ref = None, {}
    for t in idx:
        feat = ahead[t][0].cpu().numpy()
        if last is None:
            ref[t], last = True, feat
            continue
        ratio, _ = G.patch_overlap_ratio(last, feat)
        ref[t] = ratio < 0.9
        assert abs(ratio - ahead[t][2] / (feat.shape[0] - 1)) < 2e-3, (t, ratio, ahead[t][2])
        if ref[t]:
            last = feat
    assert {t: a[1] for t, a in ahead.items()} == ref
    assert sum(ref.values()) == 7                                   # one keyframe per texture
    # the plain per-frame filter takes the same decisions with bit-identical features
    slam2 = Cut3rSlam(model, conf, (Hp, Wp), buffer=32, device=DEV)
    took = {}
    for t in range(61):
        took[t] = slam2.filterx.kfFilter(t, frames[t:t + 1], intrinsics=intr)
    assert {t: took[t] for t in idx} == ref
    k = slam2.keyframes.counter.value
    assert k == 7
    for i, t in enumerate([t for t in idx if ref[t]]):
        assert torch.equal(slam2.keyframes.featI[i], ahead[t][0])


def test_loop_closure_fires_inside_the_loop_and_matches_the_cpu_restatement():
    """BASELINE configs[2] as a LOOP: `Cut3rSlam.run` with Tracking.frontend.iteration > 0 on a stream whose
    network (synth.loop_state_dict) keeps every keyframe covisible with every other, so that `TrackBackend.run` fires by itself --
    detect_loop -> NMS -> 6-view re-tracking against the matched submap -> submap-level Adam -> rewrite of every pointmap and pose
    -- every other eligible window (hi2.py:112-121 `freeze_counter`), and later windows chain on the rewritten stores.  Compared with
    oracle/slam_run.py extended by oracle/lc_oracle.LoopCloser (fp64 autograd, matrix_exp for the absent lietorch: optimiser parity
    unpinned, see tests/test_loop_closure_gpu.py for its tolerances):
      * the same closures in the same order: (current keyframe, matched keyframe), the same candidate lists, NMS scores within 5e-3;
      * the first closure takes the `loop_closure_init` path, the later ones the general term list;
      * keyframes, ordered edge lists (near-threshold exceptions as above), the trajectory and the rewritten stores, against the
        TF32 budget: the L1 objectives' sign() gradients amplify the network's rounding noise (the CPU loop run with TF32-rounded
        operands -- the reference's own arithmetic -- ends 3.4 mm ATE / 11 mm per keyframe away from its fp32 run), so the bounds are
        2.5 x what that TF32 run deviates."""
    cfg = synth.medium_config()
    sd = synth.loop_state_dict(cfg, 11)
    mf = {"thresh": 0.9, "skip": 1, "kf_every": 2}
    iters = 100
    frames = synth.pan_stream(90, H, W, pool=5, num=2, den=1, seed=0)
    so = SR.run_stream(cfg, sd, frames, INTR, mf, precision="fp32", iteration=iters)
    sotf = SR.run_stream(cfg, sd, frames, INTR, mf, precision="tf32", iteration=iters)
    assert len(so.closures) >= 3, "the oracle run closed fewer than three loops"
    # ---- HIP
    model = Cut3rModel(cfg, sd, DEV, minimal=True)
    conf = {"Tracking": {"motion_filter": dict(mf), "frontend": {"iteration": iters, "window_batch": 1}}}
    slam = Cut3rSlam(model, conf, (H, W), buffer=frames.shape[0] + 8, device=DEV)
    be = slam.backend
    seen = []
    real_scores = be.nms_scores

    def spy(ids, cur, K4):
        s = real_scores(ids, cur, K4)
        seen.append((int(cur), np.asarray(ids).copy(), s.numpy().copy()))
        return s
    be.nms_scores = spy
    intr, fr, n = torch.from_numpy(INTR), frames.to(DEV), frames.shape[0]
    did = []
    for t in range(n):
        _, _, lc = slam.run(t, fr[t:t + 1], intr, fr[t:t + 1], intr, second_last_frame=(t == n - 2), last_frame=(t == n - 1))
        if lc:
            did.append(slam.keyframes.counter.value)
    torch.cuda.synchronize()
    # ---- the same closures, organically
    assert be.closed_loop["idx_current"] == [c["idx_current"] for c in so.closures], (be.closed_loop["idx_current"], [c["idx_current"] for c in so.closures])
    assert be.closed_loop["idx_matched"] == [c["idx_matched"] for c in so.closures]
    assert did == [c["at_keyframe"] for c in so.closures]
    assert len(seen) == len(so.closures)
    worst_score = 0.0
    for (cur, ids, sc), c in zip(seen, so.closures):
        assert cur == c["idx_current"] and sorted(ids.tolist()) == sorted(c["candidates"].tolist())
        order = {int(j): k for k, j in enumerate(c["candidates"])}
        ref_sc = np.array([c["scores"][order[int(j)]] for j in ids])
        worst_score = max(worst_score, float(np.abs(sc - ref_sc).max()))
        assert all(abs(int(j) - cur) > 8 for j in ids)
    assert worst_score < 5e-3, worst_score
    assert be.lc_initialized and len(be.closed_loop["pointmaps_lc"]) == len(so.closures)
    # ---- trajectory, graph, stores
    ts, poses = slam.trajectory()
    traj = np.concatenate([ts.reshape(-1, 1).astype(np.float64), poses.astype(np.float64)], 1)
    ref = so.trajectory()
    assert np.array_equal(traj[:, 0], ref[:, 0]) and slam.tracker.t1 == so.t1
    ate = ate_rmse(traj, ref, 0.01, True)
    ii, jj, _ = slam.graph.edges_numpy()
    e_gpu, e_ref = list(zip(ii.tolist(), jj.tolist())), list(zip(so.graph.ii, so.graph.jj))
    near, unexplained = _explain_edge_differences(e_gpu, e_ref, so)
    k = so.t1
    nsub = (k - 1) // 5
    e_pm = float((slam.keyframes.submap_ds[:nsub].cpu() - so.state["submap_ds"][:nsub]).abs().max())
    e_t = float(np.abs(traj[:, 1:4] - ref[:, 1:4]).max())
    print(f"[e2e loop closure on] closures {[(c['idx_current'], c['idx_matched']) for c in so.closures]} at keyframe counts {did} | NMS score max |diff| "
          f"{worst_score:.2e} | keyframes {len(ref)} windows {len(so.windows)} path {_path_length(ref):.3f} m | ATE-RMSE HIP vs CPU {ate['rmse'] * 1e3:.3f} mm "
          f"(unaligned max |dt| {e_t * 1e3:.3f} mm) | stride-2 world pointmaps max |diff| {e_pm * 1e3:.2f} mm | edges {len(e_gpu)} vs {len(e_ref)}, "
          f"near-threshold differences {near}")
    assert not unexplained and len(near) <= 0.02 * len(e_ref) + 2
    tf = sotf.trajectory()
    assert [(c["idx_current"], c["idx_matched"]) for c in sotf.closures] == [(c["idx_current"], c["idx_matched"]) for c in so.closures]
    ate_tf = ate_rmse(tf, ref, 0.01, True)
    e_t_tf = float(np.abs(tf[:, 1:4] - ref[:, 1:4]).max())
    e_pm_tf = float((sotf.state["submap_ds"][:nsub] - so.state["submap_ds"][:nsub]).abs().max())
    print(f"[e2e loop closure on] CPU-tf32 vs CPU-fp32: ATE-RMSE {ate_tf['rmse'] * 1e3:.3f} mm, unaligned max |dt| {e_t_tf * 1e3:.3f} mm, pointmaps {e_pm_tf * 1e3:.2f} mm")
    assert ate["rmse"] <= 2.5 * ate_tf["rmse"] + 5e-5 and e_t <= 2.5 * e_t_tf + 1e-3 and e_pm <= 2.5 * e_pm_tf + 1e-3, (ate, ate_tf, e_t, e_t_tf, e_pm, e_pm_tf)
    assert ate["rmse"] <= 1e-2 and e_t <= 4e-2 and e_pm <= 4e-2                   # 3 x measured
    # the closures did something: poses of early keyframes differ from a run with the backend off
    so_off = SR.run_stream(cfg, sd, frames, INTR, mf, precision="fp32", iteration=0)
    assert np.abs(so_off.trajectory()[:, 1:4] - ref[:, 1:4]).max() > 10 * e_t
    # ---- the same loop with the windows decoded THREE at a time (round 4): the backend takes its turn after every window of a batch, looking
    # at the keyframes tracked so far -- the same closures, and every store the same bits as one window at a time
    for wb in (3, 2):
        conf_b = {"Tracking": {"motion_filter": dict(mf), "frontend": {"iteration": iters, "window_batch": wb}}}
        slam_b = Cut3rSlam(model, conf_b, (H, W), buffer=frames.shape[0] + 8, device=DEV)
        for t in range(n):
            slam_b.run(t, fr[t:t + 1], intr, fr[t:t + 1], intr, second_last_frame=(t == n - 2), last_frame=(t == n - 1))
        torch.cuda.synchronize()
        bb = slam_b.backend
        assert bb.closed_loop["idx_current"] == be.closed_loop["idx_current"] and bb.closed_loop["idx_matched"] == be.closed_loop["idx_matched"], wb
        ts_b, poses_b = slam_b.trajectory()
        assert np.array_equal(ts_b, ts) and np.array_equal(poses_b, poses), wb
        for a_, b_ in zip(slam.graph.edges_numpy(), slam_b.graph.edges_numpy()):
            assert np.array_equal(a_, b_), wb
        assert slam_b.tracker.t1 == slam.tracker.t1
        assert torch.equal(slam.keyframes.submap_ds[:nsub], slam_b.keyframes.submap_ds[:nsub]) and torch.equal(slam.keyframes.depth[:k], slam_b.keyframes.depth[:k]), wb
