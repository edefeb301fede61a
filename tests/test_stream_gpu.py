"""GPU: cut3r_resize_linear_u8 bit-exact against the oracle's restatement of cv2.resize (INTER_LINEAR, u8), the stream
generator's contract, and the demo driver end to end on a synthetic PNG sequence."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cut3r_slam_amd import ops, stream  # noqa: E402
from oracle import geom as G  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("H0,W0,H1,W1,C", [(680, 1200, 288, 512, 3), (480, 640, 384, 512, 3), (768, 1024, 384, 512, 3),
                                           (240, 320, 384, 512, 3), (31, 45, 16, 16, 4), (7, 5, 64, 48, 1), (680, 1200, 290, 512, 3)])
def test_resize_matches_oracle_bit_exact(H0, W0, H1, W1, C):
    rng = np.random.default_rng(H0 + W1)
    img = rng.integers(0, 256, (H0, W0, C), dtype=np.uint8)
    ref = G.resize_linear_u8(img, H1, W1)
    src = torch.from_numpy(img).to(DEV)
    chw = ops.resize_linear_u8(src, H1, W1, chw_out=True)
    hwc = ops.resize_linear_u8(src, H1, W1, chw_out=False)
    torch.cuda.synchronize()
    assert np.array_equal(hwc.cpu().numpy(), ref)
    assert np.array_equal(chw.permute(1, 2, 0).cpu().numpy(), ref)


def _write_sequence(dirname, n, H=480, W=640):
    from PIL import Image
    rng = np.random.default_rng(1)
    base = rng.integers(0, 256, (H // 8 + 40, W // 8 + 40, 3), dtype=np.uint8)
    base = np.kron(base, np.ones((8, 8, 1), np.uint8))
    for t in range(n):
        Image.fromarray(np.ascontiguousarray(base[t:t + H, 2 * t:2 * t + W])).save(os.path.join(dirname, f"frame{t:06d}.png"))
    return base


def test_mono_stream_contract(tmp_path):
    d = tmp_path / "colors"
    d.mkdir()
    base = _write_sequence(str(d), 3)
    calib = tmp_path / "calib.txt"
    calib.write_text("600.0 600.0 599.5 339.5")
    items = list(stream.mono_stream(str(d), str(calib), device=DEV))
    assert [it[0] for it in items] == [0, 1, 2] and [it[5] for it in items] == [False, False, True]
    t, image, intr, image_ds, intr_ds, _ = items[1]
    assert image_ds.shape == (1, 3, 384, 512) and image_ds.dtype == torch.uint8 and image_ds.is_cuda
    assert image.shape == (1, 3, 384, 512)
    np.testing.assert_allclose(intr_ds[0].numpy(), [600 * 0.8, 600 * 0.8, 599.5 * 0.8, 339.5 * 0.8])
    ref = G.resize_linear_u8(np.ascontiguousarray(base[1:481, 2:642]), 384, 512)       # PNG is lossless
    assert np.array_equal(image_ds[0].permute(1, 2, 0).cpu().numpy(), ref)


def test_demo_driver_end_to_end(tmp_path):
    import demo
    d = tmp_path / "colors"
    d.mkdir()
    _write_sequence(str(d), 36)
    calib = tmp_path / "calib.txt"
    calib.write_text("600.0 600.0 320.0 240.0")
    out = tmp_path / "out"
    # seed 1: a random tiny network whose depths stay positive on this sequence (others hit log(depth <= 0) = NaN in the
    # window scale, exactly as the reference's torch.log would)
    rc = demo.main(["--imagedir", str(d), "--calib", str(calib), "--output", str(out), "--kf_every", "2", "--synthetic-weights", "--small",
                    "--seed", "1"])
    assert rc == 0
    rows = np.loadtxt(out / "traj_kf.txt")
    assert rows.ndim == 2 and rows.shape[1] == 8 and rows.shape[0] >= 12
    assert np.all(np.diff(rows[:, 0]) > 0) and np.isfinite(rows).all()
    np.testing.assert_allclose(np.linalg.norm(rows[:, 4:], axis=1), 1.0, atol=1e-4)     # unit quaternions
    np.testing.assert_allclose(np.load(out / "intrinsics.npy"), [480.0, 480.0, 256.0, 192.0], rtol=1e-6)


def test_undistort_remap_bit_exact_vs_oracle_and_through_mono_stream(tmp_path):
    """cv2.undistort (demo_s.py:63-64; scripts/run_tum.py passes --undistort with calib/tum.txt's 5 coefficients): HIP remap ==
    the oracle's C restatement of OpenCV's fixed-point bilinear remap, bit for bit; zero distortion == identity; mono_stream
    applies it before the crop.  PARITY UNPINNED against cv2 itself (absent here)."""
    import numpy as np
    from PIL import Image
    from cut3r_slam_amd import ops
    from cut3r_slam_amd.stream import mono_stream, undistort_map
    from oracle import geom as G
    g = np.random.default_rng(3)
    H, W = 120, 160
    img = g.integers(0, 256, (H, W, 3), dtype=np.uint8)
    K4 = [129.3, 129.1, 79.6, 63.8]
    dist = [0.2624, -0.9531, -0.0054, 0.0026, 1.1633]
    ix, iy = undistort_map(K4, dist, H, W)
    got = ops.remap_linear_u8(torch.from_numpy(img).to(DEV), torch.from_numpy(ix).to(DEV), torch.from_numpy(iy).to(DEV)).cpu().numpy()
    assert np.array_equal(got, G.remap_linear_u8(img, ix, iy))
    assert (got != img).mean() > 0.5
    ix0, iy0 = undistort_map(K4, [0, 0, 0, 0], H, W)
    same = ops.remap_linear_u8(torch.from_numpy(img).to(DEV), torch.from_numpy(ix0).to(DEV), torch.from_numpy(iy0).to(DEV)).cpu().numpy()
    assert np.array_equal(same, img)
    # through the stream generator: undistort, crop 4, resize -- equals the same chain on the oracle
    d = tmp_path / "frames"
    d.mkdir()
    Image.fromarray(img).save(d / "0001.png")
    calib = np.array(K4 + dist)
    (t, image, intr, image_ds, intr_ds, last), = list(mono_stream(str(d), calib, undistort=True, cropborder=4, device=DEV))
    und = G.remap_linear_u8(img, ix, iy)[4:-4, 4:-4]
    h0, w0 = und.shape[:2]
    h1 = int((512 / w0 * h0) // 16) * 16
    ref = G.resize_linear_u8(np.ascontiguousarray(und), h1, 512)
    assert np.array_equal(image_ds[0].permute(1, 2, 0).cpu().numpy(), ref)
    np.testing.assert_allclose(intr_ds[0, 2].item(), (K4[2] - 4) * 512 / w0)


def test_demo_driver_with_the_gs_mapper(tmp_path):
    """demo.py --gs: the per-frame loop with the Gaussian mapper attached (hi2.py:47-48,133), a config with the Mapping / Training /
    opt_params sections of the maintained configs (config/scannet_config.yaml:44-79, few iterations); writes the trajectory and the map"""
    import demo
    d = tmp_path / "colors"
    d.mkdir()
    _write_sequence(str(d), 24)
    calib = tmp_path / "calib.txt"
    calib.write_text("600.0 600.0 320.0 240.0")
    cfg = tmp_path / "cfg.yaml"
    cfg.write_text("""
Mapping: {itr_num: 2}
Training: {lambda_depth: 10.0, lambda_normal: 0.1, lambda_iso: 10.0, gaussian_th: 0.1, gaussian_extent: 1.0, size_threshold: 20, window_size: 4,
           compensate_exposure: true}
opt_params: {pose_lr: 0.0001, position_lr_init: 0.0005, feature_lr: 0.005, opacity_lr: 0.05, scaling_lr: 0.001, rotation_lr: 0.001, exposure_lr: 0.0005,
             percent_dense: 0.01, densify_grad_threshold: 0.0005}
""")
    out = tmp_path / "out"
    rc = demo.main(["--imagedir", str(d), "--calib", str(calib), "--config", str(cfg), "--output", str(out), "--kf_every", "2", "--synthetic-weights",
                    "--small", "--seed", "1", "--gs", "--gs-final-iters", "20"])
    assert rc == 0
    rows = np.loadtxt(out / "traj_kf.txt")
    assert rows.ndim == 2 and rows.shape[1] == 8 and rows.shape[0] >= 8 and np.isfinite(rows).all()
    from safetensors.torch import load_file
    m = load_file(str(out / "gaussians.safetensors"))
    assert m["theta"].shape[1] == 14 and m["theta"].shape[0] > 1000 and torch.isfinite(m["theta"]).all()


def test_demo_lookahead_gives_the_frame_by_frame_trajectory(tmp_path):
    """demo.py --lookahead L --window-batch W in the reference's overlap mode (kf_every = -1): the keyframe test runs L tested frames ahead
    (one batched encoder pass + on-device decision chain per chunk), the keyframes are tracked W windows at a time -- and traj_kf.txt is the
    file of the frame-by-frame run, byte for byte."""
    import demo
    d = tmp_path / "colors"
    d.mkdir()
    _write_sequence(str(d), 48)
    calib = tmp_path / "calib.txt"
    calib.write_text("600.0 600.0 320.0 240.0")
    cfg = tmp_path / "cfg.yaml"
    cfg.write_text("Tracking:\n  motion_filter:\n    thresh: 0.9999\n    skip: 2\n    kf_every: -1\n")      # (a random tiny network: nearly every tested frame is a keyframe)
    outs = []
    for tag, extra in (("plain", []), ("ahead", ["--lookahead", "5", "--window-batch", "2"])):
        out = tmp_path / tag
        rc = demo.main(["--imagedir", str(d), "--calib", str(calib), "--config", str(cfg), "--output", str(out), "--synthetic-weights", "--small",
                        "--seed", "1"] + extra)
        assert rc == 0
        outs.append((out / "traj_kf.txt").read_bytes())
    rows = np.loadtxt(tmp_path / "plain" / "traj_kf.txt")
    assert rows.ndim == 2 and rows.shape[0] >= 12, "the sequence must yield enough keyframes for two batched windows"
    assert outs[0] == outs[1]


def test_pinned_frames_every_frame_upload_counts_and_wraps():
    """dist.PinnedFrames(upload_every_frame=True): every frame nobody claims crosses the link once (runs of consecutive frames as one copy,
    cut at the end of the recording's period and at the scratch buffer's size), claimed frames are fetched by their consumer."""
    from cut3r_slam_amd.dist import PinnedFrames
    g = torch.Generator().manual_seed(0)
    rec = torch.randint(0, 256, (50, 12, 16, 3), generator=g, dtype=torch.uint8)
    pf = PinnedFrames(rec, virtual_len=500, device=DEV, ring=4, upload_every_frame=True)
    keep = lambda f: f % 10 == 0
    pf.upload_range(0, 137, keep)
    n_unclaimed = sum(1 for f in range(137) if not keep(f))
    assert pf.bytes_uploaded == n_unclaimed * rec[0].numel()
    pf.upload_range(100, 260, keep)                   # overlapping call: only the new frames 137..259
    n2 = sum(1 for f in range(137, 260) if not keep(f))
    assert pf.bytes_uploaded == (n_unclaimed + n2) * rec[0].numel()
    got = pf.fetch(120)                               # frame 120 of the virtual stream = frame 20 of the recording
    pf.event().synchronize()
    assert torch.equal(got[0].cpu(), rec[20])
    # the last run of a call lands in the scratch buffer intact (frames 251..259 -> recording 1..9)
    torch.cuda.synchronize()
    assert torch.equal(pf._scratch[:9].cpu(), rec[1:10])
