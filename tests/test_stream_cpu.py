"""CPU: the image-sequence front end (cut3r_slam_amd/stream.py) -- file ordering, size / intrinsics arithmetic and the
trajectory writers of demo_s.py:48-113 -- and the oracle's restatement of cv2.resize(INTER_LINEAR, u8) (parity unpinned:
no cv2 in this image; checked against an independent float bilinear and against the closed-form special cases)."""
import os

import numpy as np
import torch

from cut3r_slam_amd import stream
from oracle import geom as G


def test_natural_sort_and_timestamps(tmp_path):
    names = ["frame10.jpg", "frame9.jpg", "frame000100.jpg", "frame1.jpg", "Frame2.jpg"]
    assert stream.natsorted(names) == ["frame1.jpg", "Frame2.jpg", "frame9.jpg", "frame10.jpg", "frame000100.jpg"]
    for n in ["1305031102.175304.png", "1305031102.211214.png", "1305031102.143102.png"]:
        (tmp_path / n).write_bytes(b"")
    ts = stream.frame_timestamps(str(tmp_path))
    assert ts.shape == (3, 1) and np.allclose(ts[:, 0], [1305031102.143102, 1305031102.175304, 1305031102.211214])


def test_sizes_follow_demo_s():
    assert stream.tracking_size(480, 640) == (384, 512)            # BASELINE: 640x480 -> 384x512
    assert stream.tracking_size(680, 1200) == (288, 512)           # Replica
    assert stream.mapping_size(680, 1200) == (290, 512)
    assert stream.tracking_size(968, 1296) == (368, 512)           # ScanNet


def _bilinear_float(img, H1, W1):
    H0, W0, C = img.shape
    ys = (np.arange(H1) + 0.5) * (H0 / H1) - 0.5
    xs = (np.arange(W1) + 0.5) * (W0 / W1) - 0.5
    y0 = np.floor(ys).astype(int); fy = ys - y0
    x0 = np.floor(xs).astype(int); fx = xs - x0
    fx = np.where((x0 < 0) | (x0 >= W0 - 1), 0.0, fx)
    x0c = np.clip(x0, 0, W0 - 1); x1c = np.clip(x0c + 1, 0, W0 - 1)
    y0c = np.clip(y0, 0, H0 - 1); y1c = np.clip(y0 + 1, 0, H0 - 1)
    f = img.astype(np.float64)
    top = f[y0c][:, x0c] * (1 - fx)[None, :, None] + f[y0c][:, x1c] * fx[None, :, None]
    bot = f[y1c][:, x0c] * (1 - fx)[None, :, None] + f[y1c][:, x1c] * fx[None, :, None]
    return top * (1 - fy)[:, None, None] + bot * fy[:, None, None]


def test_resize_oracle_against_float_bilinear_and_closed_forms():
    rng = np.random.default_rng(0)
    for (H0, W0, H1, W1, C) in [(48, 64, 38, 51, 3), (68, 120, 29, 51, 3), (24, 32, 38, 51, 1), (31, 45, 16, 16, 4)]:
        img = rng.integers(0, 256, (H0, W0, C), dtype=np.uint8)
        out = G.resize_linear_u8(img, H1, W1)
        ref = _bilinear_float(img, H1, W1)
        assert out.shape == (H1, W1, C)
        assert np.abs(out.astype(np.float64) - ref).max() <= 1.0 + 0.26 * (H0 / H1 > 1 or W0 / W1 > 1)      # 11-bit coefficients + >>4 truncation
    img = rng.integers(0, 256, (20, 30, 3), dtype=np.uint8)
    assert np.array_equal(G.resize_linear_u8(img, 20, 30), img)                          # identity scale: exact
    box = (img[0::2, 0::2].astype(int) + img[0::2, 1::2] + img[1::2, 0::2] + img[1::2, 1::2] + 2) >> 2
    assert np.array_equal(G.resize_linear_u8(img, 10, 15), box.astype(np.uint8))         # exact 2x: rounded 2x2 box
    const = np.full((17, 23, 3), 201, np.uint8)
    assert np.array_equal(G.resize_linear_u8(const, 40, 9), np.full((40, 9, 3), 201, np.uint8))


class _FakeKF:
    def __init__(self, n):
        class C:
            value = n + 1
        self.counter = C()
        self.tstamp = torch.arange(0, 10 * n, 10, dtype=torch.float)
        self.pose = torch.cat([torch.rand(n, 3), torch.nn.functional.normalize(torch.rand(n, 4), dim=1)], 1)
        self.intrinsic = torch.tensor([[256.0, 339.0, 255.8, 191.7]])


def test_save_trajectory_format(tmp_path):
    img_dir = tmp_path / "colors"
    img_dir.mkdir()
    for i in range(50):
        (img_dir / f"frame{i:06d}.jpg").write_bytes(b"")

    class S:
        keyframes = _FakeKF(5)
    out = tmp_path / "out"
    out.mkdir()
    traj = stream.save_trajectory(S, str(img_dir), str(out))
    rows = open(out / "traj_kf.txt").read().strip().splitlines()
    assert len(rows) == 5 and all(len(r.split()) == 8 for r in rows)
    assert [float(r.split()[0]) for r in rows] == [0.0, 10.0, 20.0, 30.0, 40.0]          # stamp = number in the file name
    assert rows[1].split()[0] == "10.0000" and len(rows[1].split()[1].split(".")[1]) == 7
    np.testing.assert_allclose(np.load(out / "intrinsics.npy"), [256.0, 339.0, 255.8, 191.7])
    assert traj.shape == (5, 8)


def test_demo_config_loader_inherits_and_cli_defaults(tmp_path):
    import demo
    (tmp_path / "base.yaml").write_text("Tracking:\n  motion_filter:\n    thresh: 2.4\n    skip_blur: false\n  frontend:\n    frontend_nms: 1\n")
    (tmp_path / "child.yaml").write_text("inherit_from: base.yaml\nTracking:\n  motion_filter:\n    thresh: 0.9\n")
    cfg = demo.load_config(str(tmp_path / "child.yaml"))
    assert cfg["Tracking"]["motion_filter"] == {"thresh": 0.9, "skip_blur": False}
    assert cfg["Tracking"]["frontend"]["frontend_nms"] == 1


def test_ate_recovers_a_known_similarity_and_noise_level(tmp_path):
    from scipy.spatial.transform import Rotation
    from cut3r_slam_amd import eval_ate as E
    rng = np.random.default_rng(3)
    n = 200
    stamps = np.arange(n) * 0.1
    gt_p = np.cumsum(rng.normal(size=(n, 3)) * 0.05, axis=0)
    R = Rotation.from_rotvec([0.3, -0.8, 0.5]).as_matrix()
    s, t = 2.7, np.array([1.0, -2.0, 0.5])
    noise = rng.normal(size=(n, 3)) * 0.01
    est_p = ((gt_p - t) @ R) / s + noise                      # gt = s R est + t up to noise
    quat = np.tile([0, 0, 0, 1.0], (n, 1))
    gt = np.concatenate([stamps[:, None], gt_p, quat], 1)
    est = np.concatenate([stamps[:, None] + 0.002, est_p, quat], 1)[::2]     # every other frame, slightly shifted stamps
    (tmp_path / "gt.txt").write_text("# tum\n" + "\n".join(" ".join(f"{v:.9f}" for v in r) for r in gt))
    (tmp_path / "est.txt").write_text("\n".join(" ".join(f"{v:.9f}" for v in r) for r in est))
    res = E.ate_rmse(E.load_tum(str(tmp_path / "est.txt")), E.load_tum(str(tmp_path / "gt.txt")))
    assert res["n"] == n // 2 and abs(res["scale"] - s) < 0.02
    assert 0.6 * s * 0.01 * np.sqrt(3) < res["rmse"] < 1.2 * s * 0.01 * np.sqrt(3)      # residual = scaled noise
    exact = E.ate_rmse(np.concatenate([stamps[:, None], ((gt_p - t) @ R) / s, quat], 1), gt)
    assert exact["rmse"] < 1e-9


def test_checkpoint_constructor_string_is_parsed_without_eval():
    import pytest
    from cut3r_slam_amd.config import production_config
    from cut3r_slam_amd.model import _config_from_ctor_string
    s = ("ARCroco3DStereo(ARCroco3DStereoConfig(freeze='encoder', pos_embed='RoPE100', rgb_head=True, pose_head=True, img_size=(512, 512), "
         "head_type='dpt', output_mode='pts3d+pose', depth_mode=('exp', -inf, inf), conf_mode=('exp', 1, inf), pose_mode=('exp', -inf, inf), "
         "enc_embed_dim=1024, enc_depth=24, enc_num_heads=16, dec_embed_dim=768, dec_depth=12, dec_num_heads=12, landscape_only=False, state_size=768, "
         "state_pe='2d', local_mem_size=256))")
    c = _config_from_ctor_string(s, production_config())
    assert (c.state_size, c.head_type, c.img_size, c.enc_depth, c.dec_num_heads, c.rope_freq, c.rgb_head) == (768, "dpt", (512, 512), 24, 12, 100.0, True)
    c1 = _config_from_ctor_string("ARCroco3DStereo(ARCroco3DStereoConfig(state_size=256, head_type='linear', img_size=(224, 224), rgb_head=False))", production_config())
    assert (c1.state_size, c1.head_type, c1.img_size, c1.rgb_head) == (256, "linear", (224, 224), False)
    for bad in ("depth_mode=('square', 0, inf)", "conf_mode=('exp', 0, inf)", "pose_head=False", "pos_embed='cosine'"):
        with pytest.raises(NotImplementedError):
            _config_from_ctor_string(f"ARCroco3DStereo(ARCroco3DStereoConfig({bad}))", production_config())


def test_undistort_map_closed_forms_and_oracle_agreement():
    """cut3r_slam_amd.stream.undistort_map (vectorised numpy, product) vs oracle/geom.undistort_map (plain per-pixel loops) and
    closed forms: zero distortion -> identity map (32*j, 32*i); pure radial k1 at the principal point -> fixed point.
    PARITY UNPINNED against cv2.initUndistortRectifyMap (cv2 is not in the image)."""
    import numpy as np
    from cut3r_slam_amd.stream import undistort_map
    from oracle import geom as G
    H, W = 37, 53
    K4 = (40.0, 42.0, 25.5, 18.25)
    ix, iy = undistort_map(K4, [0, 0, 0, 0], H, W)
    assert np.array_equal(ix, np.tile(32 * np.arange(W, dtype=np.int32), (H, 1)))
    assert np.array_equal(iy, np.tile(32 * np.arange(H, dtype=np.int32)[:, None], (1, W)))
    tum = [0.2624, -0.9531, -0.0054, 0.0026, 1.1633]             # calib/tum.txt (fr1) coefficients
    ix, iy = undistort_map(K4, tum, H, W)
    rx, ry = G.undistort_map(K4, tum, H, W)
    assert np.array_equal(ix, rx) and np.array_equal(iy, ry)
    assert np.abs(ix - 32 * np.arange(W)[None, :]).max() > 32        # the coefficients really move pixels
    ix8, iy8 = undistort_map(K4, tum + [0.01, -0.02, 0.003], H, W)
    rx8, ry8 = G.undistort_map(K4, tum + [0.01, -0.02, 0.003], H, W)
    assert np.array_equal(ix8, rx8) and np.array_equal(iy8, ry8)


def test_remap_oracle_identity_shift_and_border():
    import numpy as np
    from oracle import geom as G
    g = np.random.default_rng(0)
    img = g.integers(0, 256, (9, 11, 3), dtype=np.uint8)
    jj, ii = np.meshgrid(np.arange(11), np.arange(9))
    assert np.array_equal(G.remap_linear_u8(img, 32 * jj, 32 * ii), img)                        # identity
    out = G.remap_linear_u8(img, 32 * jj + 16, 32 * ii)                                       # half a pixel to the right
    exp = (img[:, :-1].astype(np.int64) * 16384 + img[:, 1:].astype(np.int64) * 16384 + 16384) >> 15
    assert np.array_equal(out[:, :-1], exp.astype(np.uint8))
    assert np.array_equal(out[:, -1], ((img[:, -1].astype(np.int64) * 16384 + 16384) >> 15).astype(np.uint8))    # border value 0 beyond the edge
    assert np.array_equal(G.remap_linear_u8(img, 32 * jj - 64, 32 * ii)[:, :2], np.zeros((9, 2, 3), np.uint8))   # fully outside


def test_reference_import_names_resolve_from_the_compat_directory():
    """`import curope`, `from lietorch import SE3`, `import droid_backends` with cut3r_slam_amd/compat on sys.path (a fresh
    interpreter, nothing else pre-imported): the names the reference imports (curope2d.py:7-10, track_backend.py:6, corr.py:4)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, r'%s'); import curope, lietorch, droid_backends; from lietorch import SE3, SO3, Sim3; "
            "assert callable(curope.rope_2d) and hasattr(droid_backends, 'corr_index_forward') and hasattr(SE3, 'exp'); print('ok')"
            % os.path.join(root, "cut3r_slam_amd", "compat"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd="/tmp")
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr


def test_position_learning_rate_keeps_decaying_beyond_iteration_10000():
    """the reference calls update_learning_rate(iteration) on EVERY iteration of a densifying global BA
    (hislam2/gs_backend_per_frame.py:1043-1044; general_utils.py:41-56 log-linear from position_lr_init to position_lr_final over
    position_lr_max_steps + 1000 steps); only the densification statistics stop at 10000.  finalize() runs 20000-26000 iterations."""
    import math
    from cut3r_slam_amd.gs_mapper import position_lr
    op = {"position_lr_init": 1.6e-4, "position_lr_final": 1.6e-6, "position_lr_max_steps": 20000}
    assert abs(position_lr(op, 0) - 1.6e-4) < 1e-12
    assert abs(position_lr(op, 21000) - 1.6e-6) < 1e-12 and abs(position_lr(op, 30000) - 1.6e-6) < 1e-12
    for it in (9999, 10000, 15000, 20000):
        t = it / 21000.0
        assert abs(position_lr(op, it) - math.exp(math.log(1.6e-4) * (1 - t) + math.log(1.6e-6) * t)) < 1e-15
    assert position_lr(op, 15000) < 0.4 * position_lr(op, 9999)          # it did not freeze at its iteration-9999 value
    # both formulations of global_BA apply it outside the `iteration < 10000` guard
    import inspect
    from cut3r_slam_amd import gs_mapper, gs_step
    for src in (inspect.getsource(gs_mapper.GSMapper.global_BA), inspect.getsource(gs_step.FusedTrainer.global_BA)):
        guard = src.index("if densify and \"position_lr_final\" in op")
        line_start = src.rfind("\n", 0, guard) + 1
        indent = guard - line_start
        inner = src.index("if iteration < 10000 and densify")
        assert indent <= inner - (src.rfind("\n", 0, inner) + 1), "the learning-rate decay sits inside the 10000-iteration guard again"
