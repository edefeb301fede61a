"""GPU: END-TO-END trajectory parity AT PRODUCTION SHAPE -- 384x512 frames, ViT-L/24 encoder, 768/12 dual decoder with 16 x 48
state heads, DPT head (cut3r_slam_amd.config.production_config) -- of `Cut3rSlam.run` on HIP against the CPU restatement of the
reference loop (oracle/slam_run.py) in exact fp32 AND with TF32-rounded operands (what the reference's own arithmetic is on its
GPUs, src/croco/models/croco.py:13), on the same seeded stream and weights.

The metric's second half is defined on full-size runs (/root/reference/scripts/run_scannet.py:34-36: `evo_ape tum <gt> traj_kf.txt
-vas` = Sim(3)-aligned ATE-RMSE of the keyframe trajectory); tests/test_e2e_gpu.py covers the same loop on a 64x96 network with
more windows.  Here: 33 frames at kf_every=2 -> 18 keyframes, 3 six-view tracking windows + the closing 2-view window (about 20 s of
CPU oracle per window and precision).

Asserted:
  * identical keyframe time stamps and tracking windows;
  * identical ordered edge lists, except edges whose deciding overlap ratio lies within +-0.02 of the 0.3 threshold in the oracle;
  * ATE-RMSE(HIP, CPU fp32) <= 1 mm per metre of path (BASELINE: "ATE-RMSE within 1 mm of the reference"; the random-weight network
    moves the camera by decimetres per keyframe, so the bound is stated per metre and the absolute figure is printed);
  * ATE-RMSE(HIP, CPU fp32) <= 2.5 x ATE-RMSE(CPU TF32, CPU fp32) + 0.05 mm: fp16 operands with fp32 accumulation deviate from exact
    fp32 no more than the reference's own TF32 arithmetic does;
  * per-keyframe pose agreement (translation relative to the path's extent, rotation angle) within 3x what was measured.
The achieved figures are printed (`[e2e production]`) and recorded in profiles/r03/achieved_errors.txt.
Parity status: as tests/test_e2e_gpu.py (network / overlap counts / edge bookkeeping pinned to reference fixtures; the driver
composition read-faithful, unpinned).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from cut3r_slam_amd import synth  # noqa: E402
from cut3r_slam_amd.config import production_config  # noqa: E402
from cut3r_slam_amd.eval_ate import ate_rmse  # noqa: E402
from cut3r_slam_amd.model import Cut3rModel  # noqa: E402
from cut3r_slam_amd.slam import Cut3rSlam  # noqa: E402
from oracle import slam_run as SR  # noqa: E402
from tests import tf32_budget  # noqa: E402

DEV = "cuda:0"
H, W = 384, 512
INTR = np.array([600.0 * W / 1200.0, 600.0 * H / 680.0, 599.5 * W / 1200.0, 339.5 * H / 680.0], np.float32)   # calib/replica.txt scaled
NEAR = 0.02


def _rot_angle(qa, qb):
    """angle between unit quaternions (xyzw) in radians"""
    d = np.abs((qa * qb).sum(-1) / (np.linalg.norm(qa, axis=-1) * np.linalg.norm(qb, axis=-1)))
    return 2 * np.arccos(np.clip(d, 0, 1))


def test_three_windows_at_production_shape_match_cpu_fp32_and_stay_inside_the_tf32_budget():
    torch.set_num_threads(min(16, torch.get_num_threads()))
    cfg = production_config()
    sd = synth.tracking_state_dict(cfg, 0, enc_residual_gain=0.1)
    mf = {"thresh": 0.9, "skip": 1, "kf_every": 2}
    frames = synth.pan_stream(33, H, W, pool=9, num=6, den=1, seed=0)
    # ---- HIP
    model = Cut3rModel(cfg, sd, DEV, minimal=True)
    conf = {"Tracking": {"motion_filter": dict(mf), "frontend": {"iteration": 0}}}
    slam = Cut3rSlam(model, conf, (H, W), buffer=frames.shape[0] + 8, device=DEV)
    fr, it, n = frames.to(DEV), torch.from_numpy(INTR), frames.shape[0]
    for t in range(n):
        slam.run(t, fr[t:t + 1], it, fr[t:t + 1], it, second_last_frame=(t == n - 2), last_frame=(t == n - 1))
    torch.cuda.synchronize()
    ts, poses = slam.trajectory()
    traj = np.concatenate([ts.reshape(-1, 1).astype(np.float64), poses.astype(np.float64)], 1)
    # ---- CPU restatement, exact fp32 and TF32 operands
    so32 = SR.run_stream(cfg, sd, frames, INTR, mf, precision="fp32")
    # the TF32 run of the restatement = the budget: cached (tests/tf32_budget.py; CUT3R_LIVE_TF32=1 recomputes it here, ~90 s of CPU)
    tfb = tf32_budget.cached("e2e_production_33") or tf32_budget.e2e_production(so32)
    ref, tf = so32.trajectory(), np.asarray(tfb["trajectory"], np.float64)
    assert np.isfinite(traj).all() and np.isfinite(ref).all()
    assert np.array_equal(traj[:, 0], ref[:, 0]), (traj[:, 0], ref[:, 0])
    assert len(ref) == 17 and [w[:2] for w in so32.windows] == [(0, 6), (5, 11), (10, 16), (15, 17)] and slam.tracker.t1 == so32.t1 == 17
    path = float(np.linalg.norm(np.diff(ref[:, 1:4], axis=0), axis=1).sum())
    extent = float(np.linalg.norm(ref[:, 1:4] - ref[:, 1:4].mean(0), axis=1).max())
    ate = ate_rmse(traj, ref, 0.01, True)
    ate_tf = ate_rmse(tf, ref, 0.01, True)
    dt_hip = float(np.abs(traj[:, 1:4] - ref[:, 1:4]).max())
    dt_tf = float(np.abs(tf[:, 1:4] - ref[:, 1:4]).max())
    dr_hip = float(_rot_angle(traj[:, 4:8], ref[:, 4:8]).max())
    dr_tf = float(_rot_angle(tf[:, 4:8], ref[:, 4:8]).max())
    ii, jj, _ = slam.graph.edges_numpy()
    e_gpu, e_ref, e_tf = list(zip(ii.tolist(), jj.tolist())), list(zip(so32.graph.ii, so32.graph.jj)), [tuple(e) for e in tfb["edges"]]
    near, unexplained = [], []
    for (i, j) in sorted(set(e_gpu) ^ set(e_ref)):
        r = so32.graph.ratios.get((max(i, j), min(i, j)))
        ok = r is not None and any(v is not None and abs(v - 0.3) <= NEAR for v in r)
        (near if ok else unexplained).append(((i, j), r))
    print(f"[e2e production 384x512] keyframes {len(ref)} windows {len(so32.windows)} path {path:.3f} m extent {extent:.3f} m | ATE-RMSE HIP vs CPU-fp32 "
          f"{ate['rmse'] * 1e3:.4f} mm = {ate['rmse'] * 1e3 / path:.4f} mm/m (max {ate['max'] * 1e3:.4f} mm, scale {ate['scale']:.6f}) | CPU-tf32 vs CPU-fp32 "
          f"{ate_tf['rmse'] * 1e3:.4f} mm | unaligned max |dt| hip {dt_hip * 1e3:.4f} mm tf32 {dt_tf * 1e3:.4f} mm | max rotation angle hip {dr_hip:.2e} rad "
          f"tf32 {dr_tf:.2e} rad | edges hip {len(e_gpu)} cpu {len(e_ref)} cpu-tf32 {len(e_tf)} equal {e_gpu == e_ref} (tf32 equal {e_tf == e_ref}) "
          f"near-threshold differences {near}")
    assert not unexplained, unexplained
    assert len(near) <= 0.02 * len(e_ref) + 2
    assert path > 0.05
    assert ate["rmse"] <= 1e-3 * max(path, 1.0), ate                      # 1 mm per metre of path (1 mm absolute on paths below 1 m)
    assert ate["rmse"] <= 2.5 * ate_tf["rmse"] + 5e-5, (ate, ate_tf)
    assert dt_hip <= 2.5 * dt_tf + 1e-4 * max(extent, 1.0) and dr_hip <= 2.5 * dr_tf + 2e-4, (dt_hip, dt_tf, dr_hip, dr_tf)
    # stores the chain of later windows reads: depth and stride-2 pointmaps of the tracked keyframes
    k = so32.t1
    d_ref = so32.state["depth"][:k]
    e_depth = float((slam.keyframes.depth[:k].cpu() - d_ref).abs().max() / d_ref.abs().max())
    e_depth_tf = float(tfb["e_depth_vs_fp32"])
    nsub = (k - 1) // 5
    pm_ref = so32.state["submap_ds"][:nsub]
    e_pm = float((slam.keyframes.submap_ds[:nsub].cpu() - pm_ref).abs().max() / pm_ref.abs().max())
    e_pm_tf = float(tfb["e_submaps_vs_fp32"])
    print(f"[e2e production 384x512] stored depth: hip {e_depth:.2e} tf32 {e_depth_tf:.2e} | stored stride-2 world pointmaps: hip {e_pm:.2e} tf32 {e_pm_tf:.2e} "
          "(max abs error / max abs value, vs CPU fp32)")
    assert e_depth <= 2.0 * e_depth_tf + 2e-4 and e_pm <= 2.0 * e_pm_tf + 2e-4
    # ---- the REFERENCE'S OWN loop at this shape (tests/golden/loop_production.npz: kfFilter + TrackFrontend.run of the reference with its
    # ViT-L / DPT model on the CPU, the same weights and stream): the CPU restatement reproduces it (so the budgets above are budgets against
    # the reference itself), and the HIP loop sits inside the same budget against it
    import os
    f = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "loop_production.npz"))
    assert int(frames.long().sum()) == int(f["frames_sum"])
    assert [list(w[:2]) for w in so32.windows] == f["calls"][:, 2:4].tolist() and so32.t1 == int(f["t1"])
    assert np.array_equal(so32.tstamp[:so32.counter].astype(np.int64), f["keyframes"])
    assert e_ref == list(zip(f["ii"].tolist(), f["jj"].tolist()))

    def rel(got, ref_):
        return float(np.abs(np.asarray(got, np.float64) - ref_).max() / np.abs(ref_).max())
    nsub_f = f["submap_samples"].shape[0]
    e_or = {"pose": rel(so32.state["pose"][:k].numpy(), f["pose"]), "depth": rel(so32.state["depth"][:k, 8::24, 8::32].numpy(), f["depth_samples"]),
            "submaps": rel(so32.state["submap_ds"][:nsub_f, :, 4::12, 4::16].numpy(), f["submap_samples"]),
            "conf": rel(so32.state["conf_ds"][:nsub_f].double().mean(dim=(2, 3)).numpy(), f["conf_mean"])}
    kfs = slam.keyframes
    e_hip = {"pose": rel(kfs.pose[:k].cpu().numpy(), f["pose"]), "depth": rel(kfs.depth[:k, 8::24, 8::32].cpu().numpy(), f["depth_samples"]),
             "submaps": rel(kfs.submap_ds[:nsub_f, :, 4::12, 4::16].cpu().numpy(), f["submap_samples"]),
             "conf": rel(kfs.conf_ds[:nsub_f].double().mean(dim=(2, 3)).cpu().numpy(), f["conf_mean"])}
    e_tf = {a: float(b) for a, b in tfb["vs_reference_loop"].items()}
    print("[e2e production 384x512 vs the REFERENCE'S OWN loop] CPU restatement:", {a: f"{b:.1e}" for a, b in e_or.items()}, "| HIP:", {a: f"{b:.1e}" for a, b in e_hip.items()},
          "| CPU restatement with TF32 operands (the reference's arithmetic on its GPUs):", {a: f"{b:.1e}" for a, b in e_tf.items()})
    assert max(e_or.values()) < 1e-4, e_or                               # fp32 on both sides (different GEMM blocking at width 1024)
    for name in e_hip:                                                   # the protocol of tests/test_precision_gpu.py, now against the reference itself
        assert e_hip[name] <= 2.0 * e_tf[name] + 2e-4, (name, e_hip[name], e_tf[name])
    assert e_hip["pose"] < 1e-2 and e_hip["depth"] < 1.2e-2 and e_hip["submaps"] < 1.5e-2 and e_hip["conf"] < 7e-3, e_hip
