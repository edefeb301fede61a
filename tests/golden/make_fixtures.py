#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE itself on CPU.

Run only in the build container (needs /root/reference, which never travels):

    make -C oracle ref && python tests/golden/make_fixtures.py

What is pinned here:
  * model_medium_dpt.npz : the same for head widths 64 / 48 (`medium`)
  * model_tiny_{dpt,linear}.npz : reference `inference(views, ARCroco3DStereo, "cpu")` on a tiny config with
    seeded weights from cut3r_slam_amd.weights (weights are NOT stored: they are regenerated from the seed;
    this script asserts our key/shape schema equals the reference state_dict and loads with strict=True).
  * rope2d.npz   : reference CPU rope_2d (oracle/_ref/curope.so, compiled from the reference source) incl. position -1.
  * graph.npz    : reference FactorGraph.add / add_neighborhood_factors / cal_overlap_batch / cal_overlap_bi and
    util.utils.compute_patch_overlap_ratio / pose_vec_to_matrix on seeded synthetic poses + pointmaps.

  * motion_filter.npz / frontend.npz / loop.npz / backend.npz / handover.npz / terminate.npz : the reference's OWN MotionFilter.kfFilter,
    TrackFrontend.track / run / predict, TrackBackend.run (up to its optimiser call, which needs the absent lietorch), Hi2.run / call_gs /
    terminate (with a recording test double as Gaussian mapper) on seeded streams with the medium network: keyframe decisions, window
    schedule, stores, edge lists, loop candidates / NMS choice / re-tracked submap, hand-over packets and write-backs.
  * chol.npz : geom/chol.py block_solve / schur_solve / schur_solve_mono_prior.
  * gs_utils.npz / gaussian_model.npz : the pure-torch pieces of the GS backend (SSIM + gradient, projection matrix, se(3) exp, pose update,
    lr schedule) and GaussianModel + torch.optim.Adam through Adam steps, densify_and_prune, reset_opacity.
  * camera.npz also holds util.utils.umeyama_alignment (the ATE alignment).

Harness-side adapters (nothing in the reference is modified):
  * reference objects whose constructors allocate on "cuda" (KeyFrame, TrackFrontend, TrackBackend) are created without __init__ and given
    the attributes their methods read; `.to('cuda')` / `.cuda()` are redirected to the CPU for the duration of the calls; for GaussianModel,
    whose tensor factories name device="cuda", a TorchFunctionMode aliases that device to the CPU; hi2.py's methods are compiled from its
    AST (importing the file would pull the CUDA rasteriser and the GUI).  Stub modules serve import lines only -- none of them is called.
  * `curope` is pre-registered in sys.modules as the reference's own CPU op (oracle/_ref/curope.so) so the
    reference never tries its bundled CUDA binaries; half tensors (encoder q,k, croco/models/blocks.py:125-126)
    go through that fp32 CPU op via an up/down cast, which is exactly the CUDA kernel's contract
    (fp32 math, scalar_t I/O: kernels.cu:52-80).
  * empty stub modules for lietorch/cv2/open3d/torchvision so hislam2/factor_graph.py imports.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("CUT3R_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)

from cut3r_slam_amd.config import tiny_config, state_dict_schema, Cut3rConfig  # noqa: E402
from cut3r_slam_amd.weights import synth_state_dict  # noqa: E402
from oracle import ref_curope  # noqa: E402


class _RopeAdapter:
    def __init__(self, mod):
        self._m = mod

    def rope_2d(self, tokens, positions, base, fwd):
        if tokens.dtype == torch.float32:
            self._m.rope_2d(tokens, positions, base, fwd)
        else:
            t = tokens.float().contiguous()
            self._m.rope_2d(t, positions, base, fwd)
            tokens.copy_(t.to(tokens.dtype))


def import_reference_model():
    sys.modules["curope"] = ref_curope.load()
    sys.path[:0] = [REF, os.path.join(REF, "src")]
    from src.dust3r.model import ARCroco3DStereo, ARCroco3DStereoConfig  # noqa
    from src.dust3r.inference import inference  # noqa
    import models.curope.curope2d as c2d
    c2d._kernels = _RopeAdapter(ref_curope.load())
    import models.pos_embed as pe
    assert pe.RoPE2D is c2d.cuRoPE2D, "reference fell back to the slow RoPE2D"
    return ARCroco3DStereo, ARCroco3DStereoConfig, inference


def ref_config(ARCfg, cfg: Cut3rConfig):
    inf = float("inf")
    return ARCfg(
        state_size=cfg.state_size, local_mem_size=cfg.local_mem_size, pos_embed="RoPE100",
        rgb_head=cfg.rgb_head, pose_head=True, img_size=cfg.img_size, head_type=cfg.head_type,
        output_mode="pts3d+pose", depth_mode=("exp", -inf, inf), conf_mode=("exp", 1, inf),
        pose_mode=("exp", -inf, inf), enc_embed_dim=cfg.enc_embed_dim, enc_depth=cfg.enc_depth,
        enc_num_heads=cfg.enc_num_heads, dec_embed_dim=cfg.dec_embed_dim, dec_depth=cfg.dec_depth,
        dec_num_heads=cfg.dec_num_heads, state_dec_num_heads=cfg.state_dec_num_heads,
        ray_enc_depth=cfg.ray_enc_depth, landscape_only=False, patch_embed_cls="PatchEmbedDust3R")


def make_views(imgs_u8):
    """Same dict as hislam2/track_frontend.py:47-75 (device moves are done by inference())."""
    images = (imgs_u8.float() / 255.0 - 0.5) / 0.5
    views = []
    for i in range(len(images)):
        views.append({
            "img": images[i][None],
            "ray_map": torch.full((1, 6, images[i].shape[-2], images[i].shape[-1]), torch.nan),
            "true_shape": torch.from_numpy(np.int32([images[i].shape[-2], images[i].shape[-1]])),
            "idx": i, "instance": str(i),
            "camera_pose": torch.eye(4).unsqueeze(0),
            "img_mask": torch.tensor(True).unsqueeze(0), "ray_mask": torch.tensor(False).unsqueeze(0),
            "update": torch.tensor(True).unsqueeze(0), "reset": torch.tensor(False).unsqueeze(0)})
    return views


def gen_model_fixture(name, cfg: Cut3rConfig, seed, n_views):
    AR, ARCfg, inference = import_reference_model()
    torch.manual_seed(0)
    model = AR(ref_config(ARCfg, cfg)).eval()
    ref_sd = model.state_dict()
    schema = state_dict_schema(cfg)
    missing = [k for k in ref_sd if k not in schema]
    extra = [k for k in schema if k not in ref_sd]
    assert not missing and not extra, f"schema mismatch: missing={missing[:8]} extra={extra[:8]}"
    for k, v in ref_sd.items():
        assert tuple(v.shape) == tuple(schema[k]), (k, tuple(v.shape), schema[k])
    sd = synth_state_dict(cfg, seed)
    res = torch.nn.Module.load_state_dict(model, sd, strict=True)
    print(name, "load_state_dict:", res)

    H, W = cfg.img_size
    g = np.random.Generator(np.random.PCG64(1234 + seed))
    # smooth-ish images: low-res noise upsampled + per-view shift so views differ but overlap
    base = g.integers(0, 256, size=(3, H // 4 + 4, W // 4 + 4)).astype(np.float32)
    imgs = []
    for v in range(n_views):
        crop = base[:, v:v + H // 4, v:v + W // 4]
        up = np.kron(crop, np.ones((1, 4, 4), np.float32))
        up = up + g.normal(0, 6.0, size=up.shape)
        imgs.append(np.clip(np.round(up), 0, 255).astype(np.uint8))
    imgs = torch.from_numpy(np.stack(imgs))            # [V,3,H,W] u8

    # taps: encoder features via encode_image (fp32 path, as in a window), decoder outputs via hooks
    with torch.no_grad():
        feat, pos, _ = model.encode_image({"img": model.normalize(imgs[:1].float())})
        out, state_args = inference(make_views(imgs), model, "cpu")
    fx = {"imgs": imgs.numpy(), "seed": np.int64(seed), "enc_feat0": feat.numpy(), "enc_pos0": pos.numpy()}
    for i, pred in enumerate(out["pred"]):
        for k, v in pred.items():
            fx[f"pred{i}_{k}"] = v.detach().numpy()
    for i, st in enumerate(state_args):
        fx[f"state{i}_feat"] = st[0].detach().numpy()
        fx[f"state{i}_mem"] = st[3].detach().numpy()
    fx["state_pos"] = state_args[0][1].numpy()
    import json
    fx["config_json"] = np.frombuffer(json.dumps(cfg.to_dict()).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **fx)
    print("wrote", name, {k: v.shape for k, v in fx.items() if hasattr(v, "shape")})


def gen_rope_fixture():
    g = np.random.Generator(np.random.PCG64(7))
    fx = {}
    for D in (16, 48, 64):
        B, N, Hh = 2, 9, 3
        tok = g.standard_normal((B, N, Hh, D)).astype(np.float32)
        pos = g.integers(0, 40, size=(B, N, 2)).astype(np.int64)
        pos[0, 0] = (-1, -1)           # pose token position (model.py:769-771)
        pos[1, 3] = (0, 31)
        for F0 in (1.0, -1.0):
            t = torch.from_numpy(tok.copy())
            ref_curope.rope_2d_ref(t, torch.from_numpy(pos), 100.0, F0)
            fx[f"D{D}_F{int(F0)}_out"] = t.numpy()
        fx[f"D{D}_tok"] = tok
        fx[f"D{D}_pos"] = pos
    np.savez_compressed(os.path.join(HERE, "rope2d.npz"), **fx)
    print("wrote rope2d", list(fx))


def import_reference_graph():
    for m in ("lietorch", "cv2", "open3d", "torchvision", "torchvision.transforms"):
        if m not in sys.modules:
            sys.modules[m] = types.ModuleType(m)
    sys.modules["lietorch"].SE3 = object
    sys.path[:0] = [REF, os.path.join(REF, "hislam2")]
    from factor_graph import FactorGraph
    from util.utils import pose_vec_to_matrix, compute_patch_overlap_ratio, depth_to_pointmap
    return FactorGraph, pose_vec_to_matrix, compute_patch_overlap_ratio, depth_to_pointmap


def synth_trajectory(g, n, step=0.22, yaw_step=0.06):
    """Seeded camera path that revisits its start (so far-apart KFs overlap): poses as [t, q_xyzw]."""
    from scipy.spatial.transform import Rotation
    poses = []
    for i in range(n):
        ang = 2 * np.pi * i / (n - 2)            # slightly more than one lap of a lateral circle
        t = np.array([0.9 * np.cos(ang) - 0.9, 0.6 * np.sin(ang), 0.15 * np.sin(2 * ang)])
        yaw = 0.12 * np.sin(ang + 0.3) + g.normal(0, 0.01)
        q = Rotation.from_euler("yxz", [yaw, 0.05 * np.cos(ang), 0.02 * np.sin(ang)]).as_quat()
        poses.append(np.concatenate([t, q]))
    return np.asarray(poses, np.float32)


def gen_graph_fixture():
    FG, pose_vec_to_matrix, patch_overlap, depth_to_pointmap = import_reference_graph()
    g = np.random.Generator(np.random.PCG64(11))
    n, H, W = 24, 24, 32                       # down-sampled pointmap size (H/2, W/2 of a 48x64 frame)
    K = np.array([[30.0, 0, 15.5], [0, 30.0, 11.5], [0, 0, 1.0]])
    poses = torch.from_numpy(synth_trajectory(g, n))
    c2w = pose_vec_to_matrix(poses)
    depth = torch.from_numpy(g.uniform(2.0, 3.0, size=(n, H, W)).astype(np.float32))
    pm = depth_to_pointmap(depth, c2w, K[0, 0], K[1, 1], K[0, 2], K[1, 2])      # [n,H,W,3] world
    kf = types.SimpleNamespace()
    graph = FG(kf, device="cpu", max_factors=48)
    fx = {"poses": poses.numpy(), "pointmaps": pm.numpy(), "K": K, "c2w": c2w.numpy()}
    # replay TrackFrontend.track's graph calls (track_frontend.py:166-262) for init window + later KFs
    graph.add_neighborhood_factors(0, 3, r=3)
    for i in range(n):
        if i >= 6:
            graph.add_neighborhood_factors(i - 3, i + 1, r=3)
        if i > 2:
            graph.add(i, c2w[:i], pm[:i], c2w[i], pm[i], K)
        fx[f"ii_{i}"] = graph.ii.numpy().copy()
        fx[f"jj_{i}"] = graph.jj.numpy().copy()
        fx[f"age_{i}"] = graph.age.numpy().copy()
    i = n - 1
    fx["ovl_batch_last"] = graph.cal_overlap_batch(pm[i], c2w[:i], K).numpy()
    fx["ovl_bi_last"] = graph.cal_overlap_bi(pm[:i], c2w[i][None], K).numpy()
    fx["loop_last"] = np.asarray(graph.detect_loop(i, None, torch.zeros(n, 1)) if graph.detect_loop(
        i, None, torch.zeros(n, 1)) is not None else [], np.int64)
    # patch overlap on seeded features
    f0 = torch.from_numpy(g.standard_normal((48, 32)).astype(np.float32))
    ratios = []
    for a in (0.0, 0.5, 1.0, 2.0):
        f1 = f0[torch.from_numpy(g.permutation(48))] + a * torch.from_numpy(g.standard_normal((48, 32)).astype(np.float32))
        fx[f"feat1_{len(ratios)}"] = f1.numpy()
        ratios.append(patch_overlap(f0, f1))
    fx["feat0"] = f0.numpy()
    fx["patch_ratios"] = np.asarray(ratios, np.float64)
    np.savez_compressed(os.path.join(HERE, "graph.npz"), **fx)
    print("wrote graph: edges", len(fx[f"ii_{n-1}"]), "loop", fx["loop_last"], "ratios", ratios)


def gen_nms_fixture():
    """Loop-candidate scoring: the reference's FactorGraph.NMS (factor_graph.py:561-582) itself, run on the CPU.  Its body calls
    `.cuda()` on its arguments; the harness turns that into a no-op (torch.Tensor.cuda patched for this call only -- the
    reference file is untouched).  Also records compute_feature_overlap_batch (:328-341) and both cal_overlap_bi calls."""
    FG, pose_vec_to_matrix, patch_overlap, depth_to_pointmap = import_reference_graph()
    g = np.random.Generator(np.random.PCG64(23))
    n, H, W, Np, Cf = 14, 24, 32, 48, 32
    K = np.array([[30.0, 0, 15.5], [0, 30.0, 11.5], [0, 0, 1.0]])
    poses = torch.from_numpy(synth_trajectory(g, n))
    c2w = pose_vec_to_matrix(poses)
    depth = torch.from_numpy(g.uniform(2.0, 3.0, size=(n, H, W)).astype(np.float32))
    pm = depth_to_pointmap(depth, c2w, K[0, 0], K[1, 1], K[0, 2], K[1, 2])
    base = g.standard_normal((Np, Cf)).astype(np.float32)
    feats = np.stack([base[g.permutation(Np)] + a * g.standard_normal((Np, Cf)).astype(np.float32)
                      for a in np.linspace(0.0, 1.6, n)]).astype(np.float32)
    feats = torch.from_numpy(feats)
    graph = FG(types.SimpleNamespace(), device="cpu", max_factors=48)
    cur = n - 1
    fx = {"poses": poses.numpy(), "c2w": c2w.numpy(), "pointmaps": pm.numpy(), "K": K, "feats": feats.numpy(), "idx_current": np.int64(cur)}
    real_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        for name, ids in (("a", [0, 1, 2, 3]), ("b", [2, 5, 7]), ("c", [8, 9])):
            ids_t = torch.as_tensor(ids)
            mask, feat_sim = graph.compute_feature_overlap_batch(feats[cur], feats[ids_t], return_item=True)
            a2c = graph.cal_overlap_bi(pm[ids_t], c2w[cur][None], K).squeeze(-1)
            c2a = graph.cal_overlap_bi(pm[cur][None], c2w[ids_t], K).squeeze(0)
            scores = 0.8 * (a2c + c2a) / 2 + 0.2 * feat_sim
            for th in (0.4, 0.95):
                k = graph.NMS(pm[ids_t], feats[ids_t], c2w[ids_t], pm[cur], feats[cur], c2w[cur], K, th=th)
                fx[f"{name}_k_th{int(th * 100)}"] = np.int64(-1 if k is None else int(k))
            fx[f"{name}_ids"] = np.asarray(ids, np.int64)
            fx[f"{name}_feat_sim"] = feat_sim.numpy()
            fx[f"{name}_feat_mask"] = mask.numpy()
            fx[f"{name}_a2c"] = a2c.numpy()
            fx[f"{name}_c2a"] = c2a.numpy()
            fx[f"{name}_scores"] = scores.numpy()
    finally:
        torch.Tensor.cuda = real_cuda
    np.savez_compressed(os.path.join(HERE, "nms.npz"), **fx)
    print("wrote nms", {k: (v.tolist() if v.size < 6 else v.shape) for k, v in fx.items() if len(k) > 1 and k[1] == "_"})


def gen_camera_fixture():
    """pose_encoding_to_camera / quaternion_to_matrix (src/dust3r/utils/camera.py:364-420), geotrf (utils/geometry.py:49-115)
    and hislam2 pose_vec_to_matrix (util/utils.py:676-700) on seeded inputs -- the host pose helpers of the trackers."""
    sys.path[:0] = [REF, os.path.join(REF, "src")]
    from src.dust3r.utils.camera import pose_encoding_to_camera, quaternion_to_matrix
    from src.dust3r.utils.geometry import geotrf
    _, pose_vec_to_matrix, _, _ = import_reference_graph()
    g = np.random.Generator(np.random.PCG64(31))
    enc = g.standard_normal((9, 7)).astype(np.float32)
    enc[:, 3:] /= np.linalg.norm(enc[:, 3:], axis=1, keepdims=True)
    enc[4, 3:] *= 1.7                                        # non-unit quaternion: two_s = 2 / |q|^2 (camera.py:378)
    c2w = pose_encoding_to_camera(torch.from_numpy(enc))
    pts = torch.from_numpy(g.standard_normal((9, 5, 7, 3)).astype(np.float32))
    vec = g.standard_normal((9, 7)).astype(np.float32)       # (t, q_xyzw), un-normalised: pose_vec_to_matrix normalises
    fx = {"enc": enc, "c2w": c2w.numpy(), "R": quaternion_to_matrix(torch.from_numpy(enc[:, 3:])).numpy(),
          "pts": pts.numpy(), "geotrf": geotrf(c2w, pts).numpy(), "pose_vec": vec,
          "pose_vec_c2w": pose_vec_to_matrix(torch.from_numpy(vec)).numpy()}
    # the other branches of geotrf (geometry.py:85-115): one matrix on a point list, a batch of matrices on a batch of point lists, 3x3
    # matrices, the z = norm projection, numpy inputs  (round 4: pins cut3r_slam_amd/dust3r_utils.geotrf, the zero-edit model boundary)
    pl = torch.from_numpy(g.standard_normal((9, 11, 3)).astype(np.float32))
    pl[..., 2] = pl[..., 2].abs() + 0.5
    fx.update({"pts_list": pl.numpy(), "geotrf_single": geotrf(c2w[2], pl[0]).numpy(), "geotrf_batch": geotrf(c2w, pl).numpy(),
               "geotrf_rot3": geotrf(c2w[:, :3, :3].contiguous(), pl).numpy(), "geotrf_norm": geotrf(c2w[:, :3, :3].contiguous(), pl, norm=2.0, ncol=2).numpy(),
               "geotrf_numpy": geotrf(c2w[3].numpy(), pl[1].numpy()), "inv": __import__("src.dust3r.utils.geometry", fromlist=["inv"]).inv(c2w).numpy()})
    # util.utils.umeyama_alignment (:738-763): a similarity between two point sets, once a proper one and once through a reflection
    from util.utils import umeyama_alignment
    src = g.standard_normal((40, 3))
    from scipy.spatial.transform import Rotation
    Rt = Rotation.from_euler("xyz", [0.4, -0.7, 1.1]).as_matrix()
    for name, M in (("proper", Rt), ("reflected", Rt @ np.diag([1.0, 1.0, -1.0]))):
        dst = 1.7 * src @ M.T + np.array([0.3, -1.2, 2.0]) + 0.01 * g.standard_normal((40, 3))
        sc, R, t = umeyama_alignment(src, dst)
        fx.update({f"um_{name}_src": src, f"um_{name}_dst": dst, f"um_{name}_scale": np.float64(sc), f"um_{name}_R": R, f"um_{name}_t": t})
    np.savez_compressed(os.path.join(HERE, "camera.npz"), **fx)
    print("wrote camera", {k: v.shape for k, v in fx.items()})


def gen_frontend_fixture():
    """TrackFrontend.track (hislam2/track_frontend.py:166-262) ITSELF on the CPU: the initialisation window and two steady-state windows
    of the reference's tracker -- its own prepare_input / inference / prepare_output, log-depth scale chaining, pose composition, stride-2
    stores and FactorGraph calls -- over 16 keyframes of a seeded pan, with the reference model (medium config, seeded weights).
    Harness-side adapters (the reference file is untouched): the tracker object is created without its __init__ (which builds a
    FactorGraph on "cuda:0") and given the attributes track() reads; the keyframe store is a namespace with the tensors of
    keyframe.py:19-36 (its constructor allocates three of them on "cuda"); the one `.to('cuda')` of prepare_input (:48) is redirected
    to the CPU for the duration of the calls."""
    AR, ARCfg, inference = import_reference_model()          # (before the stub modules: transformers probes torchvision on import)
    import_reference_graph()
    from track_frontend import TrackFrontend
    from factor_graph import FactorGraph
    from cut3r_slam_amd import synth
    cfg = synth.medium_config()
    seed = 11
    sd = synth.tracking_state_dict(cfg, seed)
    torch.manual_seed(0)
    model = AR(ref_config(ARCfg, cfg)).eval()
    print("frontend: load_state_dict:", torch.nn.Module.load_state_dict(model, sd, strict=True))
    H, W = cfg.img_size
    n, buffer, ds = 16, 20, 2
    frames = synth.pan_stream(n, H, W, pool=5, num=6, den=1, seed=4)          # 6 px per keyframe: consecutive keyframes overlap
    intr = torch.tensor([80.0, 80.0, 47.5, 31.5])
    kf = types.SimpleNamespace(
        image=torch.zeros(buffer, 3, H, W, dtype=torch.uint8), intrinsic=torch.zeros(buffer, 4), pose=torch.zeros(buffer, 7),
        submap_ds=torch.ones(buffer // 5, 6, H // ds, W // ds, 3), conf_ds=torch.zeros(buffer // 5, 6, H // ds, W // ds),
        depth=torch.ones(buffer, H, W), tstamp=torch.zeros(buffer), mono_depth_alpha=None)
    kf.pose[:] = torch.as_tensor([0, 0, 0, 0, 0, 0, 1.0])
    kf.image[:n] = frames
    kf.intrinsic[:n] = intr
    graph = FactorGraph(kf, device="cpu", max_factors=48)
    tr = object.__new__(TrackFrontend)
    tr.device, tr.keyframes, tr.model, tr.graph = "cpu", kf, model, graph
    tr.verbose, tr.output_dir, tr.use_gt, tr.conf_th, tr.downsample_ratio, tr.t1 = False, None, False, 0.5, ds, 0
    real_to = torch.Tensor.to

    def to_cpu(self, *a, **k):
        a = tuple("cpu" if isinstance(x, str) and x.startswith("cuda") else x for x in a)
        return real_to(self, *a, **k)
    # (the frames are not stored: synth.pan_stream(16, 64, 96, pool=5, num=6, den=1, seed=4) regenerates them; their sum is)
    fx = {"frames_sum": np.int64(int(frames.long().sum())), "intrinsic": intr.numpy(), "seed": np.int64(seed),
          "windows": np.asarray([[0, 6, 1], [5, 11, 0], [10, 16, 0]], np.int64)}
    torch.Tensor.to = to_cpu
    try:
        with torch.no_grad():
            for w, (t0, t1, init) in enumerate(fx["windows"].tolist()):
                tr.track(t0, t1, init=bool(init))
                # what the window wrote: poses / depths of its keyframes t0..t1-1, its submap (index t0 // 5), the edge list after it
                fx[f"pose_{w}"] = kf.pose[t0:t1].numpy().copy()
                fx[f"depth_{w}"] = kf.depth[t0:t1].numpy().copy()
                fx[f"submap_ds_{w}"] = kf.submap_ds[t0 // 5].numpy().copy()
                fx[f"conf_ds_{w}"] = kf.conf_ds[t0 // 5].numpy().copy()
                fx[f"ii_{w}"], fx[f"jj_{w}"], fx[f"age_{w}"] = graph.ii.numpy().copy(), graph.jj.numpy().copy(), graph.age.numpy().copy()
            # TrackFrontend.predict (:102-162): a non-keyframe (here: frame 13 of the pan) relocalised against keyframe 9 of the tracked map
            p_pose, p_depth, p_pm, p_conf = tr.predict(frames[13], kf.image[9], kf.pose[9], kf.depth[9], kf.submap_ds[1, 4])
            fx.update(predict_pose=p_pose.numpy().copy(), predict_depth=p_depth.numpy().copy(), predict_pointmap=p_pm.numpy().copy(),
                      predict_conf=p_conf.numpy().copy(), predict_args=np.asarray([13, 9], np.int64))
    finally:
        torch.Tensor.to = real_to
    np.savez_compressed(os.path.join(HERE, "frontend.npz"), **fx)
    print("wrote frontend:", {k: v.shape for k, v in fx.items() if hasattr(v, "shape")})
    print("  edges after each window:", [len(fx[f"ii_{w}"]) for w in range(3)], "| log-depth scales of the chained windows:",
          [float(np.log(fx[f"depth_{w}"][0]).mean()) for w in (1, 2)])


def gen_loop_fixture():
    """The per-frame loop of Hi2.run without its backend / mapper calls (hislam2/hi2.py:101-111): MotionFilter.kfFilter followed by
    TrackFrontend.run for every frame of a seeded stream (fixed cadence kf_every = 2, 45 frames, the second-last / last frame flags of
    demo_s.py) -- the reference's own window scheduling (warm-up of 6, `t1 < counter - 5`, the closing window of the last frame), its
    return values (run_backend flag, keyframe range, submap index) and the final stores and edge list.  Adapters as in the two fixtures
    above (KeyFrame / TrackFrontend created without their __init__, `.to('cuda')` redirected)."""
    AR, ARCfg, inference = import_reference_model()
    import_reference_graph()
    import motion_filter as MF
    from keyframe import KeyFrame
    from track_frontend import TrackFrontend
    from factor_graph import FactorGraph
    from torch.multiprocessing import Value
    from cut3r_slam_amd import synth
    cfg = synth.medium_config()
    seed = 11
    sd = synth.tracking_state_dict(cfg, seed)
    torch.manual_seed(0)
    model = AR(ref_config(ARCfg, cfg)).eval()
    torch.nn.Module.load_state_dict(model, sd, strict=True)
    H, W = cfg.img_size
    n, buffer, ds = 45, 32, 2
    frames = synth.pan_stream(n, H, W, pool=5, num=3, den=1, seed=2)
    kf = object.__new__(KeyFrame)
    kf.counter, kf.ready, kf.is_initialized, kf.downsample_ratio = Value("i", 0), Value("i", 0), False, ds
    kf.tstamp = torch.zeros(buffer)
    kf.image = torch.zeros(buffer, 3, H, W, dtype=torch.uint8)
    kf.intrinsic, kf.pose, kf.depth = torch.zeros(buffer, 4), torch.zeros(buffer, 7), torch.ones(buffer, H, W)
    kf.pose[:] = torch.as_tensor([0, 0, 0, 0, 0, 0, 1.0])
    kf.submap_ds = torch.ones(buffer // 5, 6, H // ds, W // ds, 3)
    kf.conf_ds = torch.zeros(buffer // 5, 6, H // ds, W // ds)
    kf.featI = torch.zeros(buffer, (H // 16) * (W // 16), cfg.enc_embed_dim)
    kf.pos = torch.zeros(buffer, (H // 16) * (W // 16), 2, dtype=torch.int64)
    filt = MF.MotionFilter(model, kf, {"thresh": 0.9, "skip": 1, "kf_every": 2, "skip_blur": False}, device="cpu")
    graph = FactorGraph(kf, device="cpu", max_factors=48)
    tr = object.__new__(TrackFrontend)
    tr.device, tr.keyframes, tr.model, tr.graph = "cpu", kf, model, graph
    tr.verbose, tr.output_dir, tr.use_gt, tr.conf_th, tr.downsample_ratio, tr.t1, tr.warmup = False, None, False, 0.5, ds, 0, 6
    real_to = torch.Tensor.to

    def to_cpu(self, *a, **k):
        a = tuple("cpu" if isinstance(x, str) and x.startswith("cuda") else x for x in a)
        return real_to(self, *a, **k)
    intr = torch.tensor([80.0, 80.0, 47.5, 31.5])
    calls = []
    torch.Tensor.to = to_cpu
    try:
        with torch.no_grad():
            for t in range(n):
                filt.kfFilter(t, frames[t:t + 1], intrinsics=intr, second_last_frame=(t == n - 2), last_frame=(t == n - 1))
                flag, rng, sub = tr.run(t, last_frame=(t == n - 1))
                if rng is not None:
                    calls.append([t, int(bool(flag)), rng.start, rng.stop, int(sub)])
    finally:
        torch.Tensor.to = real_to
    k, t1 = kf.counter.value, tr.t1
    fx = {"seed": np.int64(seed), "frames_sum": np.int64(int(frames.long().sum())), "intrinsic": intr.numpy(), "calls": np.asarray(calls, np.int64),
          "keyframes": kf.tstamp[:k].numpy().astype(np.int64), "t1": np.int64(t1), "pose": kf.pose[:t1].numpy().copy(),
          "depth_mean": kf.depth[:t1].mean(dim=(1, 2)).numpy().copy(), "submap_ds": kf.submap_ds[:(t1 - 1) // 5 + 1].numpy().copy(),
          "conf_mean": kf.conf_ds[:(t1 - 1) // 5 + 1].mean(dim=(2, 3)).numpy().copy(),
          "ii": graph.ii.numpy().copy(), "jj": graph.jj.numpy().copy(), "age": graph.age.numpy().copy()}
    np.savez_compressed(os.path.join(HERE, "loop.npz"), **fx)
    print("wrote loop: keyframes", k, "tracked", t1, "calls (frame, run_backend, t0, t1, submap):", calls, "| edges", len(fx["ii"]))


def gen_backend_fixture(production=False):
    """TrackBackend.run (hislam2/track_backend.py:527-586) ITSELF on the CPU up to its optimiser call: the per-frame loop of Hi2.run
    (hi2.py:101-121: kfFilter, TrackFrontend.run, every other eligible window the backend) over a seeded stream with weights for which
    the loop detector fires (synth.loop_state_dict); at the first backend call the reference's own detect_loop scan, FactorGraph.NMS
    choice and 6-view re-tracking (TrackBackend.track, :137-217) run, and the arguments it hands to loop_closure_init are recorded --
    the optimiser itself needs lietorch, which the reference tree does not hold, so it is replaced by the recorder and the
    fixture ends there.  Adapters as above, plus `.cuda()` as a no-op (NMS) and an empty `lietorch` module for the import line."""
    AR, ARCfg, inference = import_reference_model()
    import_reference_graph()
    for m in ("tqdm",):
        if m not in sys.modules:
            try:
                __import__(m)
            except Exception:
                sys.modules[m] = types.ModuleType(m)
                sys.modules[m].tqdm = lambda x, *a, **k: x
    import motion_filter as MF
    from keyframe import KeyFrame
    from track_frontend import TrackFrontend
    from track_backend import TrackBackend
    from factor_graph import FactorGraph
    from torch.multiprocessing import Value
    from cut3r_slam_amd import synth
    cfg = synth.medium_config()
    seed = 11
    n, buffer, ds = 90, 64, 2
    if production:                       # (`backend_production`: the same at 384x512 with the ViT-L / DPT network; minutes of CPU, on request)
        from cut3r_slam_amd.config import production_config
        cfg, seed, n, buffer = production_config(), 0, 40, 24
    sd = synth.loop_state_dict(cfg, seed, enc_residual_gain=0.1) if production else synth.loop_state_dict(cfg, seed)
    torch.manual_seed(0)
    model = AR(ref_config(ARCfg, cfg)).eval()
    torch.nn.Module.load_state_dict(model, sd, strict=True)
    H, W = (384, 512) if production else cfg.img_size
    frames = synth.pan_stream(n, H, W, pool=9, num=6, den=1, seed=0) if production else synth.pan_stream(n, H, W, pool=5, num=2, den=1, seed=0)
    kf = object.__new__(KeyFrame)
    kf.counter, kf.ready, kf.is_initialized, kf.downsample_ratio = Value("i", 0), Value("i", 0), False, ds
    kf.tstamp = torch.zeros(buffer)
    kf.image = torch.zeros(buffer, 3, H, W, dtype=torch.uint8)
    kf.intrinsic, kf.pose, kf.depth = torch.zeros(buffer, 4), torch.zeros(buffer, 7), torch.ones(buffer, H, W)
    kf.pose[:] = torch.as_tensor([0, 0, 0, 0, 0, 0, 1.0])
    kf.submap_ds = torch.ones(buffer // 5, 6, H // ds, W // ds, 3)
    kf.conf_ds = torch.zeros(buffer // 5, 6, H // ds, W // ds)
    kf.featI = torch.zeros(buffer, (H // 16) * (W // 16), cfg.enc_embed_dim)
    kf.pos = torch.zeros(buffer, (H // 16) * (W // 16), 2, dtype=torch.int64)
    filt = MF.MotionFilter(model, kf, {"thresh": 0.9, "skip": 1, "kf_every": 2, "skip_blur": False}, device="cpu")
    graph = FactorGraph(kf, device="cpu", max_factors=48)
    tr = object.__new__(TrackFrontend)
    tr.device, tr.keyframes, tr.model, tr.graph = "cpu", kf, model, graph
    tr.verbose, tr.output_dir, tr.use_gt, tr.conf_th, tr.downsample_ratio, tr.t1, tr.warmup = False, None, False, 0.5, ds, 0, 6
    be = object.__new__(TrackBackend)
    be.device, be.keyframes, be.model, be.graph = "cpu", kf, model, graph
    be.verbose, be.output_dir, be.conf_th, be.downsample_ratio, be.lc_initialized = False, None, 0.05, ds, False
    be.closed_loop = {"idx_current": [], "idx_matched": [], "pointmaps_lc": []}
    rec = {}

    def recorder(pointmap_current_lc, idx_matched, idx_current):
        rec.update(pointmap_current_lc=pointmap_current_lc.detach().clone(), idx_matched=int(idx_matched), idx_current=int(idx_current))
        return None
    be.loop_closure_init = recorder
    real_track, real_detect, real_nms = be.track, graph.detect_loop, graph.NMS

    def track_spy(selected_idx, anchor_sub_num):
        out = real_track(selected_idx, anchor_sub_num)
        rec.update(selected_idx=selected_idx.numpy().copy(), anchor_sub_num=int(anchor_sub_num), pointmaps_lc=out[0].clone(), confs_lc=out[1].clone(), poses_lc=out[2].clone())
        return out

    def detect_spy(idx, *a, **k):
        r = real_detect(idx, *a, **k)
        rec.setdefault("scan", []).append((int(idx), None if r is None else np.asarray(r).copy()))
        return r

    def nms_spy(*a, **k):
        r = real_nms(*a, **k)
        rec["k_th"] = -1 if r is None else int(r)
        return r
    be.track, graph.detect_loop, graph.NMS = track_spy, detect_spy, nms_spy
    real_to, real_cuda = torch.Tensor.to, torch.Tensor.cuda

    def to_cpu(self, *a, **k):
        a = tuple("cpu" if isinstance(x, str) and x.startswith("cuda") else x for x in a)
        return real_to(self, *a, **k)
    intr = torch.tensor([600.0 * W / 1200.0, 600.0 * H / 680.0, 599.5 * W / 1200.0, 339.5 * H / 680.0]) if production else torch.tensor([80.0, 80.0, 47.5, 31.5])
    torch.Tensor.to, torch.Tensor.cuda = to_cpu, (lambda self, *a, **k: self)
    fired_at, freeze, windows = None, 0, []
    try:
        with torch.no_grad():
            for t in range(n):
                filt.kfFilter(t, frames[t:t + 1], intrinsics=intr, second_last_frame=(t == n - 2), last_frame=(t == n - 1))
                flag, rng, sub = tr.run(t, last_frame=(t == n - 1))
                if rng is not None:
                    windows.append([t, int(bool(flag)), rng.start, rng.stop])
                if flag:                                   # hi2.py:112-121
                    if freeze > 0:
                        rec.clear()
                        ok, _ = be.run()
                        freeze = 0
                        if ok:
                            fired_at = t
                            break
                    else:
                        freeze += 1
    finally:
        torch.Tensor.to, torch.Tensor.cuda = real_to, real_cuda
    assert fired_at is not None, "the reference backend never closed a loop on this stream"
    k, t1 = kf.counter.value, tr.t1
    cand = [c for _, c in rec["scan"] if c is not None][-1]
    fx = {"seed": np.int64(seed), "frames_sum": np.int64(int(frames.long().sum())), "intrinsic": intr.numpy(), "fired_at_frame": np.int64(fired_at),
          "windows": np.asarray(windows, np.int64), "keyframes": kf.tstamp[:k].numpy().astype(np.int64),
          "scan_idx": np.asarray([i for i, _ in rec["scan"]], np.int64), "candidates": np.asarray(cand, np.int64), "k_th": np.int64(rec["k_th"]),
          "idx_current": np.int64(rec["idx_current"]), "idx_matched": np.int64(rec["idx_matched"]), "selected_idx": rec["selected_idx"].astype(np.int64),
          "anchor_sub_num": np.int64(rec["anchor_sub_num"]), "pointmaps_lc": rec["pointmaps_lc"].numpy(), "confs_lc": rec["confs_lc"].numpy(),
          "poses_lc": rec["poses_lc"].numpy(), "pointmap_current_lc": rec["pointmap_current_lc"].numpy(),
          "pose_before": kf.pose[:t1].numpy().copy(), "ii": graph.ii.numpy().copy(), "jj": graph.jj.numpy().copy()}
    if production:                       # strided samples instead of the full re-tracked submap
        fx["pointmaps_lc"], fx["confs_lc"] = fx["pointmaps_lc"][:, 4::12, 4::16].copy(), fx["confs_lc"][:, 4::12, 4::16].copy()
        fx["pointmap_current_lc"] = fx["pointmap_current_lc"][:, 4::12, 4::16].copy()
    np.savez_compressed(os.path.join(HERE, "backend_production.npz" if production else "backend.npz"), **fx)
    print("wrote backend: fired at frame", fired_at, "keyframes", k, "| scan", fx["scan_idx"].tolist(), "candidates", fx["candidates"].tolist(), "k_th", int(fx["k_th"]),
          "-> matched", int(fx["idx_matched"]), "current", int(fx["idx_current"]), "| selected", fx["selected_idx"].tolist())


def gen_chol_fixture():
    """hislam2/geom/chol.py ITSELF on the CPU (pure torch; its `import geom.projective_ops` pulls `lietorch` names that only the
    projective functions use: the empty stub module of import_reference_graph() serves the import line): block_solve, schur_solve (full and
    sless) and schur_solve_mono_prior on seeded SPD systems with GENERAL (not block-diagonal) blocks, in the reference's fp32 and once more
    with fp64 inputs (same code) as the tight target.  Pins the reduced solves of the legacy dense-BA stack."""
    import_reference_graph()
    sys.modules["lietorch"].Sim3 = object
    sys.path[:0] = [os.path.join(REF, "hislam2")]
    from geom import chol
    g = torch.Generator().manual_seed(17)
    fx = {}
    # ---- schur_solve: P poses of 6 dof, M disparity maps of HW pixels
    P, M, D, HW = 4, 3, 6, 24
    A = torch.randn(P * D, P * D, generator=g, dtype=torch.float64)
    Hf = A @ A.T + 6.0 * torch.eye(P * D, dtype=torch.float64)
    H = Hf.reshape(P, D, P, D).permute(0, 2, 1, 3)[None].contiguous()                     # [1,P,P,D,D]
    E = torch.randn(1, P, M, D, HW, generator=g, dtype=torch.float64) * 0.25
    C = torch.rand(1, M, HW, generator=g, dtype=torch.float64) * 2 + 6.0
    v = torch.randn(1, P, D, generator=g, dtype=torch.float64)
    w = torch.randn(1, M, HW, generator=g, dtype=torch.float64)
    fx.update(ss_H=H.numpy(), ss_E=E.numpy(), ss_C=C.numpy(), ss_v=v.numpy(), ss_w=w.numpy())
    for tag, cast in (("f32", torch.float32), ("f64", torch.float64)):
        c = lambda t: t.to(cast)
        dx, dz, cov = chol.schur_solve(c(H), c(E), c(C), c(v), c(w))
        fx[f"ss_dx_{tag}"], fx[f"ss_dz_{tag}"], fx[f"ss_cov_{tag}"] = dx.double().numpy(), dz.double().numpy(), cov.double().numpy()
        fx[f"ss_dx_sless_{tag}"] = chol.schur_solve(c(H), c(E), c(C), c(v), c(w), sless=True).double().numpy()
        fx[f"bs_x_{tag}"] = chol.block_solve(c(H), c(v)).double().numpy()
    # ---- schur_solve_mono_prior: M frames, D = hs * ws scale-grid nodes each
    M2, D2, HW2 = 3, 6, 40
    A = torch.randn(M2 * D2, M2 * D2, generator=g, dtype=torch.float64)
    Hf = A @ A.T + 4.0 * torch.eye(M2 * D2, dtype=torch.float64)
    Hs = Hf.reshape(M2, D2, M2, D2).permute(0, 2, 1, 3)[None].contiguous()
    Es = torch.randn(1, M2, M2, D2, HW2, generator=g, dtype=torch.float64) * 0.3
    vs = torch.randn(1, M2, D2, generator=g, dtype=torch.float64)
    C2 = torch.rand(1, M2, HW2, generator=g, dtype=torch.float64) * 2 + 6.0
    w2 = torch.randn(1, M2, HW2, generator=g, dtype=torch.float64)
    fx.update(mp_Hs=Hs.numpy(), mp_Es=Es.numpy(), mp_vs=vs.numpy(), mp_C=C2.numpy(), mp_w=w2.numpy())
    for tag, cast in (("f32", torch.float32), ("f64", torch.float64)):
        c = lambda t: t.to(cast)
        dso, dz, cov = chol.schur_solve_mono_prior(c(C2), c(w2), c(Hs), c(Es), c(vs), dzcov=True)
        fx[f"mp_dso_{tag}"], fx[f"mp_dz_{tag}"], fx[f"mp_cov_{tag}"] = dso.double().numpy(), dz.double().numpy(), cov.double().numpy()
    np.savez_compressed(os.path.join(HERE, "chol.npz"), **fx)
    rel = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())
    print("wrote chol: reference fp32 vs its fp64 run:", {k: f"{rel(fx[k + '_f32'], fx[k + '_f64']):.1e}" for k in ("ss_dx", "ss_dz", "ss_cov", "bs_x", "mp_dso", "mp_dz", "mp_cov")})


def gen_gs_utils_fixture():
    """The pure-torch pieces of the reference's Gaussian-splatting backend, run on the CPU (the rasteriser itself is a CUDA extension and
    cannot run here): loss_utils.ssim with its autograd gradient, graphics_utils.getProjectionMatrix2 / getWorld2View2, slam_utils.SE3_exp
    (small and large angles), slam_utils.update_pose / get_pose on a camera namespace, project2world, general_utils.helper (the position
    learning-rate schedule) and inverse_sigmoid.  Stub modules for cv2 / matplotlib serve the import lines only."""
    for m in ("cv2", "matplotlib", "matplotlib.cm"):
        if m not in sys.modules:
            sys.modules[m] = types.ModuleType(m)
    sys.modules["matplotlib"].cm = sys.modules["matplotlib.cm"]
    sys.path[:0] = [os.path.join(REF, "hislam2")]
    from gaussian.utils import loss_utils, graphics_utils, slam_utils, general_utils
    g = torch.Generator().manual_seed(29)
    fx = {}
    # ---- SSIM value + gradient (the mapper's 0.2 (1 - ssim) term)
    a = torch.rand(3, 40, 56, generator=g, dtype=torch.float64)
    b = (a + 0.15 * torch.randn(3, 40, 56, generator=g, dtype=torch.float64)).clamp(0, 1)
    ar = a.clone().requires_grad_(True)
    val = loss_utils.ssim(ar, b)
    val.backward()
    fx.update(ssim_a=a.numpy(), ssim_b=b.numpy(), ssim_value=np.float64(val.item()), ssim_grad_a=ar.grad.numpy(),
              l1_value=np.float64(loss_utils.l1_loss(a, b).item()))
    # ---- projection / view matrices
    cams = [(0.01, 100.0, 31.5, 23.5, 40.0, 42.0, 64, 48), (0.01, 100.0, 250.3, 190.9, 440.0, 441.5, 512, 384)]
    fx["proj_args"] = np.asarray(cams, np.float64)
    fx["proj"] = np.stack([graphics_utils.getProjectionMatrix2(*c).numpy() for c in cams])
    Rm = torch.linalg.qr(torch.randn(3, 3, generator=g))[0]
    tv = torch.randn(3, generator=g)
    fx.update(w2v_R=Rm.numpy(), w2v_t=tv.numpy(), w2v=graphics_utils.getWorld2View2(Rm, tv).numpy())
    # ---- SE3_exp and the pose update
    taus = torch.cat([torch.randn(6, 6, generator=g, dtype=torch.float64) * 0.3, torch.randn(2, 6, generator=g, dtype=torch.float64) * 1e-7,
                      torch.tensor([[0.1, -0.2, 0.3, 0.0, 0.0, 0.0], [0.0, 0.0, 0.0, 2.5, -1.0, 0.7]], dtype=torch.float64)], 0)
    fx["tau"] = taus.numpy()
    fx["se3_exp"] = np.stack([slam_utils.SE3_exp(t).numpy() for t in taus])
    cam = types.SimpleNamespace(R=torch.linalg.qr(torch.randn(3, 3, generator=g))[0], T=torch.randn(3, generator=g),
                                cam_trans_delta=torch.tensor([0.02, -0.01, 0.03]), cam_rot_delta=torch.tensor([0.01, 0.02, -0.015]))

    def update_RT(R, t):
        cam.R, cam.T = R.clone(), t.clone()
    cam.update_RT = update_RT
    fx.update(cam_R=cam.R.numpy().copy(), cam_T=cam.T.numpy().copy(), cam_trans_delta=cam.cam_trans_delta.numpy().copy(),
              cam_rot_delta=cam.cam_rot_delta.numpy().copy(), get_pose=slam_utils.get_pose(cam).numpy())
    slam_utils.update_pose(cam)
    fx.update(updated_R=cam.R.numpy().copy(), updated_T=cam.T.numpy().copy())
    # ---- project2world, learning-rate schedule, inverse sigmoid
    c2w = torch.eye(4)[None].repeat(2, 1, 1)
    c2w[:, :3, :3] = torch.linalg.qr(torch.randn(2, 3, 3, generator=g))[0]
    c2w[:, :3, 3] = torch.randn(2, 3, generator=g)
    dep = torch.rand(2, 12, 16, generator=g) * 3 + 0.5
    fx.update(p2w_c2w=c2w.numpy(), p2w_depth=dep.numpy(), p2w=slam_utils.project2world(c2w, dep, 20.0, 21.0, 7.5, 5.5).numpy())
    steps = np.asarray([0, 1, 10, 100, 999, 1000, 5000, 29999, 30000, 40000], np.int64)
    fx["lr_steps"] = steps
    fx["lr"] = np.asarray([general_utils.helper(int(st), 0.00016, 0.0000016, 0, 0.01, 30000) for st in steps], np.float64)
    fx["lr_delay"] = np.asarray([general_utils.helper(int(st), 0.00016, 0.0000016, 500, 0.01, 30000) for st in steps], np.float64)
    x = torch.tensor([0.01, 0.1, 0.5, 0.9, 0.99])
    fx.update(isig_x=x.numpy(), isig=general_utils.inverse_sigmoid(x).numpy())
    np.savez_compressed(os.path.join(HERE, "gs_utils.npz"), **fx)
    print("wrote gs_utils: ssim", float(val), "| proj[0]", fx["proj"][0].round(4).tolist())


class _CudaIsCpu(torch.overrides.TorchFunctionMode):
    """harness-side device alias for reference code that writes device="cuda" into its tensor factories: inside this mode every `device`
    argument naming cuda becomes the CPU.  Nothing of the reference's arithmetic is replaced."""

    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = dict(kwargs or {})
        d = kwargs.get("device")
        if d is not None and "cuda" in str(d):
            kwargs["device"] = "cpu"
        return func(*args, **kwargs)


def gen_gaussian_model_fixture():
    """The optimiser and densification book-keeping of the reference's GaussianModel (hislam2/gaussian/scene/gaussian_model.py) ITSELF on
    the CPU: training_setup (torch.optim.Adam, one group per attribute, eps 1e-15), Adam steps on seeded gradients, update_learning_rate,
    add_densification_stats (incl. the absolute-gradient statistic), densify_and_prune (clone / split by the gradient OR quantile rule,
    pruning incl. the < 5e-4 rule) and reset_opacity with their optimiser-state surgery (cat / prune / replace), then more steps.
    Recorded after every phase: all parameters, both Adam moments and the step count; the standard-normal draws of densify_and_split.
    Adapters: stub modules for open3d / plyfile / simple_knn / cv2 / matplotlib (import lines only), device="cuda" aliased to the CPU by
    the TorchFunctionMode above, `.cuda()` as a no-op."""
    for m in ("open3d", "plyfile", "simple_knn", "simple_knn._C", "cv2", "matplotlib", "matplotlib.cm"):
        if m not in sys.modules:
            sys.modules[m] = types.ModuleType(m)
    sys.modules["plyfile"].PlyData = sys.modules["plyfile"].PlyElement = object
    sys.modules["simple_knn._C"].distCUDA2 = None
    sys.modules["matplotlib"].cm = sys.modules["matplotlib.cm"]
    sys.path[:0] = [os.path.join(REF, "hislam2")]
    from gaussian.scene.gaussian_model import GaussianModel
    g = torch.Generator().manual_seed(41)
    torch.manual_seed(43)                      # the global generator feeds densify_and_split's torch.normal
    P = 240
    opt = types.SimpleNamespace(position_lr_init=0.0005, position_lr_final=0.000005, position_lr_max_steps=2000, feature_lr=0.005, opacity_lr=0.05,
                                scaling_lr=0.001, rotation_lr=0.001, percent_dense=0.01)
    extent, max_grad, min_opacity, size_threshold = 1.0, 0.0005, 0.05, 20
    init = {"xyz": torch.randn(P, 3, generator=g), "f_dc": torch.randn(P, 1, 3, generator=g) * 0.5, "f_rest": torch.zeros(P, 0, 3),
            "opacity": torch.randn(P, 1, generator=g) * 1.5 - 0.5,
            # log scales: a third above percent_dense * extent = 0.01 (split candidates), a few below 5e-4 (pruned as degenerate)
            "scaling": torch.log(torch.cat([torch.rand(P // 3, 3, generator=g) * 0.03 + 0.012, torch.rand(P - P // 3 - 6, 3, generator=g) * 0.006 + 0.001,
                                            torch.rand(6, 3, generator=g) * 3e-4 + 1e-4], 0)),
            "rotation": torch.nn.functional.normalize(torch.randn(P, 4, generator=g), dim=-1)}
    fx = {f"init_{k}": v.numpy().copy() for k, v in init.items() if k != "f_rest"}
    fx.update(opt=np.asarray([opt.position_lr_init, opt.position_lr_final, opt.position_lr_max_steps, opt.feature_lr, opt.opacity_lr, opt.scaling_lr,
                              opt.rotation_lr, opt.percent_dense, extent, max_grad, min_opacity, size_threshold], np.float64))
    real_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    real_normal = torch.normal
    draws = []

    def normal_spy(*a, **k):
        out = real_normal(*a, **k)
        std = k.get("std", a[1] if len(a) > 1 else None)
        draws.append((out / std).detach().clone())
        return out
    names = ("xyz", "f_dc", "opacity", "scaling", "rotation")

    def snapshot(gm, tag):
        for grp in gm.optimizer.param_groups:
            if grp["name"] not in names:
                continue
            prm = grp["params"][0]
            st = gm.optimizer.state.get(prm, {})
            fx[f"{tag}_{grp['name']}"] = prm.detach().reshape(prm.shape[0], -1).numpy().copy()
            if st:
                fx[f"{tag}_{grp['name']}_m"] = st["exp_avg"].reshape(prm.shape[0], -1).numpy().copy()
                fx[f"{tag}_{grp['name']}_v"] = st["exp_avg_sq"].reshape(prm.shape[0], -1).numpy().copy()
                fx[f"{tag}_step"] = np.float64(float(st["step"]))
        fx[f"{tag}_lr_xyz"] = np.float64([grp["lr"] for grp in gm.optimizer.param_groups if grp["name"] == "xyz"][0])

    def step(gm, it, gseed, between=None):
        """one iteration; `between` runs between the statistics and optimizer.step(), where the reference's loops densify / reset
        (gs_backend_per_frame.py:425-438, 1025-1041)"""
        gg = torch.Generator().manual_seed(gseed)
        n = gm._xyz.shape[0]
        grads = {"xyz": torch.randn(n, 3, generator=gg) * 1e-3, "f_dc": torch.randn(n, 1, 3, generator=gg) * 1e-2, "opacity": torch.randn(n, 1, generator=gg) * 1e-2,
                 "scaling": torch.randn(n, 3, generator=gg) * 1e-3, "rotation": torch.randn(n, 4, generator=gg) * 1e-3}
        for grp in gm.optimizer.param_groups:
            prm = grp["params"][0]
            prm.grad = grads[grp["name"]].clone() if grp["name"] in grads else torch.zeros_like(prm)
        fx[f"grad_{it}"] = torch.cat([grads[k].reshape(n, -1) for k in names], 1).numpy().copy()
        # screen-space gradient of one rendered view: (x, y, absolute statistic), visible subset
        vs = types.SimpleNamespace(grad=torch.cat([torch.randn(n, 2, generator=gg) * 3.2e-4, torch.rand(n, 1, generator=gg) * 2e-3], 1))
        vis = torch.rand(n, generator=gg) < 0.8
        fx[f"vs_{it}"], fx[f"vis_{it}"] = vs.grad.numpy().copy(), vis.numpy().copy()
        gm.max_radii2D[vis] = torch.max(gm.max_radii2D[vis], torch.randint(1, 30, (int(vis.sum()),), generator=gg).float())
        fx[f"radii_{it}"] = gm.max_radii2D.numpy().copy()
        gm.add_densification_stats(vs, vis)
        if between is not None:
            between()
        gm.optimizer.step()
        gm.optimizer.zero_grad(set_to_none=True)
        gm.update_learning_rate(it)
    torch.normal = normal_spy
    try:
        with _CudaIsCpu():
            gm = GaussianModel(sh_degree=0, config=None)
            gm.init_lr(1.0)
            # (extend_from_pcd appends through the optimiser, which training_setup creates from existing tensors: set them directly)
            gm._xyz = torch.nn.Parameter(init["xyz"].clone())
            gm._features_dc = torch.nn.Parameter(init["f_dc"].clone())
            gm._features_rest = torch.nn.Parameter(init["f_rest"].clone())
            gm._opacity = torch.nn.Parameter(init["opacity"].clone())
            gm._scaling = torch.nn.Parameter(init["scaling"].clone())
            gm._rotation = torch.nn.Parameter(init["rotation"].clone())
            gm.max_radii2D = torch.zeros(P)
            gm.unique_kfIDs, gm.n_obs = torch.zeros(P).int(), torch.zeros(P).int()
            gm.training_setup(opt)
            for it in range(4):
                step(gm, it, 100 + it)
            snapshot(gm, "a")                                             # after 4 steps
            gm.densify_and_prune(max_grad, min_opacity, extent, size_threshold)
            snapshot(gm, "b")                                             # after clone / split / prune
            fx["n_after_densify"] = np.int64(gm._xyz.shape[0])
            for it in range(4, 7):
                step(gm, it, 100 + it)
            snapshot(gm, "c")
            gm.reset_opacity()
            snapshot(gm, "d")
            step(gm, 7, 107)
            gm.densify_and_prune(max_grad, min_opacity, extent, None)     # second round without the screen-size rule
            step(gm, 8, 108)
            snapshot(gm, "e")
            # round 4: the ORDER inside the reference's training loops -- backward, statistics, densify_and_prune, THEN
            # optimizer.step(): the re-created parameters have no .grad, so that step changes nothing (no update, no moment update, no step
            # count); likewise reset_opacity before the step: the opacity group is skipped, the others step
            step(gm, 9, 109, between=lambda: gm.densify_and_prune(max_grad, min_opacity, extent, None))
            snapshot(gm, "f")
            fx["f_steps_by_group"] = np.asarray([float(gm.optimizer.state[g_["params"][0]]["step"]) for g_ in gm.optimizer.param_groups if g_["name"] in names])
            step(gm, 10, 110, between=gm.reset_opacity)
            snapshot(gm, "g")
            fx["g_steps_by_group"] = np.asarray([float(gm.optimizer.state[g_["params"][0]]["step"]) for g_ in gm.optimizer.param_groups if g_["name"] in names])
            fx["group_names"] = np.asarray(names)
    finally:
        torch.Tensor.cuda, torch.normal = real_cuda, real_normal
    for i, d in enumerate(draws):
        fx[f"split_draws_{i}"] = d.numpy().copy()
    np.savez_compressed(os.path.join(HERE, "gaussian_model.npz"), **fx)
    print("wrote gaussian_model: P", P, "->", int(fx["n_after_densify"]), "->", fx["e_xyz"].shape[0], "| split draws", [d.shape[0] for d in draws],
          "| steps", [float(fx[f"{t}_step"]) for t in "abcdefg"], "| lr_xyz", [float(fx[f"{t}_lr_xyz"]) for t in "abcdefg"],
          "| steps by group after densify-then-step", fx["f_steps_by_group"], "after reset-then-step", fx["g_steps_by_group"])


def gen_tf32_budgets():
    """tests/golden/tf32_budgets.json: the TF32 halves of the production-shape precision budgets (tests/tf32_budget.py: what they are and
    why they are cached).  These are outputs of the CPU RESTATEMENT (oracle/), not of the reference -- the restatement itself is pinned
    to the reference by the fixtures above; ~5 minutes of CPU on 8 cores; on request only."""
    import json
    root = os.path.dirname(os.path.dirname(HERE))
    if root not in sys.path:
        sys.path.insert(0, root)
    from tests import tf32_budget as TB
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    out = {"six_view_window": TB.six_view_window(False), "six_view_window_outliers": TB.six_view_window(True), "e2e_production_33": TB.e2e_production()}
    out["_generated_with"] = {"threads": torch.get_num_threads(), "torch": torch.__version__}
    with open(TB.PATH, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote tf32_budgets", {k: (v if k.startswith("six") else "...") for k, v in out.items()})


def handover_mapper_update(data):
    """the deterministic 'mapper' of the hand-over fixture (a test double for the GS backend's interface, shared with the tests): refined
    poses, depths with a hole, full-resolution pointmaps -- simple functions of the packet it was given"""
    poses = data["poses"].double().clone()
    poses[:, :3] += 0.01 * torch.arange(1, poses.shape[0] + 1, dtype=torch.float64, device=poses.device)[:, None]
    depths = data["depths"].clone() * 1.02
    depths[:, :8, :8] = 0
    pm = data["depths"][..., None] * torch.tensor([0.5, -0.25, 1.0], device=depths.device)
    return {"poses": poses, "depths": depths, "pointmaps": pm}, list(data["viz_idx"])


def gen_handover_fixture():
    """Hi2.run and Hi2.call_gs (hislam2/hi2.py:56-133) THEMSELVES on the CPU with a recording test double in place of the Gaussian mapper:
    the packet the tracker hands to `mapper.run` after every window (viz_idx, submap_idx, tstamp, poses, images, pointmaps[:n], confs[:n],
    depths, intrinsics) and what call_gs writes back (poses; the masked depth assignment on an advanced-indexing copy, i.e. nothing;
    stride-2 pointmaps; the overlap row of the previous submaps).  hi2.py cannot be imported here (its import chain pulls the CUDA
    rasteriser, the GUI, ...): the two method definitions are compiled from the file's AST into a bare class; KeyFrame / MotionFilter /
    TrackFrontend are the reference's objects as in the fixtures above."""
    import ast
    AR, ARCfg, inference = import_reference_model()
    import_reference_graph()
    import motion_filter as MF
    from keyframe import KeyFrame
    from track_frontend import TrackFrontend
    from factor_graph import FactorGraph
    from torch.multiprocessing import Value
    from cut3r_slam_amd import synth
    tree = ast.parse(open(os.path.join(REF, "hislam2", "hi2.py")).read())
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "Hi2"][0]
    fns = {n.name: n for n in cls.body if isinstance(n, ast.FunctionDef)}
    ns = {"torch": torch, "np": np, "os": os, "viz_pcd": lambda *a, **k: None}
    exec(compile(ast.Module(body=[fns["call_gs"], fns["run"]], type_ignores=[]), os.path.join(REF, "hislam2", "hi2.py"), "exec"), ns)
    Hi2 = type("Hi2", (), {"call_gs": ns["call_gs"], "run": ns["run"]})
    cfg = synth.medium_config()
    seed = 11
    sd = synth.tracking_state_dict(cfg, seed)
    torch.manual_seed(0)
    model = AR(ref_config(ARCfg, cfg)).eval()
    torch.nn.Module.load_state_dict(model, sd, strict=True)
    H, W = cfg.img_size
    n, buffer, ds = 45, 32, 2
    frames = synth.pan_stream(n, H, W, pool=5, num=3, den=1, seed=2)
    kf = object.__new__(KeyFrame)
    kf.counter, kf.ready, kf.is_initialized, kf.downsample_ratio = Value("i", 0), Value("i", 0), False, ds
    kf.tstamp = torch.zeros(buffer)
    kf.image = torch.zeros(buffer, 3, H, W, dtype=torch.uint8)
    kf.intrinsic, kf.pose, kf.depth = torch.zeros(buffer, 4), torch.zeros(buffer, 7), torch.ones(buffer, H, W)
    kf.pose[:] = torch.as_tensor([0, 0, 0, 0, 0, 0, 1.0])
    kf.submap_ds = torch.ones(buffer // 5, 6, H // ds, W // ds, 3)
    kf.conf_ds = torch.zeros(buffer // 5, 6, H // ds, W // ds)
    kf.featI = torch.zeros(buffer, (H // 16) * (W // 16), cfg.enc_embed_dim)
    kf.pos = torch.zeros(buffer, (H // 16) * (W // 16), 2, dtype=torch.int64)
    graph = FactorGraph(kf, device="cpu", max_factors=48)
    tr = object.__new__(TrackFrontend)
    tr.device, tr.keyframes, tr.model, tr.graph = "cpu", kf, model, graph
    tr.verbose, tr.output_dir, tr.use_gt, tr.conf_th, tr.downsample_ratio, tr.t1, tr.warmup = False, None, False, 0.5, ds, 0, 6
    packets = []

    class Recorder:
        def run(self, data, iterations):
            packets.append({k: (v.clone() if torch.is_tensor(v) else (list(v) if isinstance(v, range) else v)) for k, v in data.items()})
            packets[-1]["iterations"] = iterations
            return handover_mapper_update(data)
    slam = Hi2()
    slam.images, slam.keyframes, slam.tracker, slam.backend, slam.do_lc, slam.freeze_counter = {}, kf, tr, None, False, 0
    slam.filterx = MF.MotionFilter(model, kf, {"thresh": 0.9, "skip": 1, "kf_every": 2, "skip_blur": False}, device="cpu")
    slam.mapper, slam.gs_iter_num, slam.verbose, slam.downsample_ratio, slam.output_dir = Recorder(), 7, False, ds, None
    real_to = torch.Tensor.to

    def to_cpu(self, *a, **k):
        a = tuple("cpu" if isinstance(x, str) and x.startswith("cuda") else x for x in a)
        return real_to(self, *a, **k)
    intr = torch.tensor([[80.0, 80.0, 47.5, 31.5]])
    fx = {"seed": np.int64(seed), "frames_sum": np.int64(int(frames.long().sum())), "intrinsic": intr[0].numpy()}
    torch.Tensor.to = to_cpu
    try:
        for t in range(n):
            before = len(packets)
            slam.run(t, frames[t:t + 1], intr, frames[t:t + 1], intr[0], second_last_frame=(t == n - 2), last_frame=(t == n - 1))
            if len(packets) != before:
                k = len(packets) - 1
                pk = packets[-1]
                viz, sub = list(pk["viz_idx"]), int(pk["submap_idx"])
                fx.update({f"pk{k}_frame": np.int64(t), f"pk{k}_viz_idx": np.asarray(viz, np.int64), f"pk{k}_submap_idx": np.int64(sub), f"pk{k}_iterations": np.int64(pk["iterations"]),
                           f"pk{k}_tstamp": pk["tstamp"].numpy(), f"pk{k}_poses": pk["poses"].numpy(), f"pk{k}_pointmaps_mean": pk["pointmaps"].double().mean(dim=(1, 2)).numpy(),
                           f"pk{k}_confs_mean": pk["confs"].double().mean(dim=(1, 2)).numpy(), f"pk{k}_depths_mean": pk["depths"].mean(dim=(1, 2)).numpy(), f"pk{k}_intrinsics": pk["intrinsics"].numpy(),
                           f"pk{k}_images_shape": np.asarray(pk["images"].shape, np.int64), f"pk{k}_images_sum": np.int64(int(pk["images"].long().sum())),
                           f"pk{k}_pose_after": kf.pose[viz].numpy().copy(), f"pk{k}_depth_after_mean": kf.depth[viz].mean(dim=(1, 2)).numpy().copy(),
                           f"pk{k}_depth_after_corner": kf.depth[viz][:, :8, :8].mean(dim=(1, 2)).numpy().copy(),
                           f"pk{k}_submaps_after_mean": kf.submap_ds[max(0, sub - 1):sub + 2].double().mean(dim=(2, 3)).numpy().copy()})        # the rows call_gs can touch: this submap, the overlap row of the one before, the next
    finally:
        torch.Tensor.to = real_to
    fx["n_packets"] = np.int64(len(packets))
    fx["keys"] = np.frombuffer(",".join(sorted(k for k in packets[0] if k != "iterations")).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "handover.npz"), **fx)
    print("wrote handover:", len(packets), "packets at frames", [int(fx[f"pk{k}_frame"]) for k in range(len(packets))], "| viz_idx", [fx[f"pk{k}_viz_idx"].tolist() for k in range(len(packets))],
          "| packet keys", bytes(fx["keys"]).decode(), "| depth corner after write-back (0 would mean the masked write took effect)", fx["pk0_depth_after_corner"].round(3).tolist())


def gen_terminate_fixture():
    """Hi2.terminate(add_kf=True) (hislam2/hi2.py:152-229) ITSELF on the CPU after the reference loop ran over a stream whose keyframes lie
    32 frames apart: which in-between frames it relocalises (interval rule :186-196), the TrackFrontend.predict call on each, what it
    hands to `mapper.add_new_view`, and the write-back of `mapper.finalize()`'s poses.  `mapper` is a recording test double; hi2.py's
    methods are compiled from its AST as in gen_handover_fixture; `F` of its namespace is torch.nn.functional."""
    import ast
    AR, ARCfg, inference = import_reference_model()
    import_reference_graph()
    import motion_filter as MF
    from keyframe import KeyFrame
    from track_frontend import TrackFrontend
    from factor_graph import FactorGraph
    from torch.multiprocessing import Value
    from cut3r_slam_amd import synth
    tree = ast.parse(open(os.path.join(REF, "hislam2", "hi2.py")).read())
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "Hi2"][0]
    fns = {n.name: n for n in cls.body if isinstance(n, ast.FunctionDef)}
    ns = {"torch": torch, "np": np, "os": os, "F": torch.nn.functional, "viz_pcd": lambda *a, **k: None}
    exec(compile(ast.Module(body=[fns["call_gs"], fns["run"], fns["terminate"]], type_ignores=[]), os.path.join(REF, "hislam2", "hi2.py"), "exec"), ns)
    Hi2 = type("Hi2", (), {"call_gs": ns["call_gs"], "run": ns["run"], "terminate": ns["terminate"]})
    cfg = synth.medium_config()
    seed = 11
    sd = synth.tracking_state_dict(cfg, seed)
    torch.manual_seed(0)
    model = AR(ref_config(ARCfg, cfg)).eval()
    torch.nn.Module.load_state_dict(model, sd, strict=True)
    H, W = cfg.img_size
    n, buffer, ds = 262, 16, 2
    frames = synth.pan_stream(n, H, W, pool=5, num=1, den=2, seed=6)
    kf = object.__new__(KeyFrame)
    kf.counter, kf.ready, kf.is_initialized, kf.downsample_ratio = Value("i", 0), Value("i", 0), False, ds
    kf.tstamp = torch.zeros(buffer)
    kf.image = torch.zeros(buffer, 3, H, W, dtype=torch.uint8)
    kf.intrinsic, kf.pose, kf.depth = torch.zeros(buffer, 4), torch.zeros(buffer, 7), torch.ones(buffer, H, W)
    kf.pose[:] = torch.as_tensor([0, 0, 0, 0, 0, 0, 1.0])
    kf.submap_ds = torch.ones(buffer // 5, 6, H // ds, W // ds, 3)
    kf.conf_ds = torch.zeros(buffer // 5, 6, H // ds, W // ds)
    kf.featI = torch.zeros(buffer, (H // 16) * (W // 16), cfg.enc_embed_dim)
    kf.pos = torch.zeros(buffer, (H // 16) * (W // 16), 2, dtype=torch.int64)
    graph = FactorGraph(kf, device="cpu", max_factors=48)
    tr = object.__new__(TrackFrontend)
    tr.device, tr.keyframes, tr.model, tr.graph = "cpu", kf, model, graph
    tr.verbose, tr.output_dir, tr.use_gt, tr.conf_th, tr.downsample_ratio, tr.t1, tr.warmup = False, None, False, 0.5, ds, 0, 6
    added = []

    class Recorder:
        def run(self, data, iterations):
            return {"poses": data["poses"].double(), "depths": data["depths"], "pointmaps": torch.zeros(len(list(data["viz_idx"])), H, W, 3)
                    + data["depths"][..., None] * torch.tensor([0.5, -0.25, 1.0])}, list(data["viz_idx"])

        def add_new_view(self, new_img, new_pose, new_depth, new_pointmap, new_conf, new_kf_tstamp, kf_sub_idx):
            added.append({"img_shape": tuple(new_img.shape), "img_sum": int(new_img.long().sum()), "pose": new_pose.clone(), "depth": new_depth.clone(),
                          "pointmap": new_pointmap.clone(), "conf": new_conf.clone(), "tstamp": int(new_kf_tstamp), "sub": int(kf_sub_idx)})

        def finalize(self):
            p = kf.pose[:kf.counter.value].double().numpy().copy()
            p[:, :3] += 0.005 * np.arange(1, p.shape[0] + 1)[:, None]
            return p
    slam = Hi2()
    slam.images, slam.keyframes, slam.tracker, slam.backend, slam.do_lc, slam.freeze_counter = {}, kf, tr, None, False, 0
    slam.filterx = MF.MotionFilter(model, kf, {"thresh": 0.9, "skip": 1, "kf_every": 32, "skip_blur": False}, device="cpu")
    slam.mapper, slam.gs_iter_num, slam.verbose, slam.downsample_ratio, slam.output_dir = Recorder(), 1, False, ds, None
    real_to = torch.Tensor.to

    def to_cpu(self, *a, **k):
        a = tuple("cpu" if isinstance(x, str) and x.startswith("cuda") else x for x in a)
        return real_to(self, *a, **k)
    intr = torch.tensor([[80.0, 80.0, 47.5, 31.5]])
    torch.Tensor.to = to_cpu
    try:
        for t in range(n):
            slam.run(t, frames[t:t + 1], intr, frames[t:t + 1], intr[0], second_last_frame=(t == n - 2), last_frame=(t == n - 1))
        with torch.no_grad():
            traj = slam.terminate(n - 1, fill=False, eval_render=False, gaussian_retrain=False, add_kf=True)
    finally:
        torch.Tensor.to = real_to
    k = kf.counter.value
    fx = {"seed": np.int64(seed), "frames_sum": np.int64(int(frames.long().sum())), "intrinsic": intr[0].numpy(), "keyframes": kf.tstamp[:k].numpy().astype(np.int64),
          "tracked": np.int64(tr.t1), "added_tstamp": np.asarray([a["tstamp"] for a in added], np.int64), "added_sub": np.asarray([a["sub"] for a in added], np.int64),
          "added_img_sum": np.asarray([a["img_sum"] for a in added], np.int64), "added_img_shape": np.asarray([a["img_shape"] for a in added], np.int64),
          "added_pose": torch.cat([a["pose"] for a in added]).numpy(), "added_depth_mean": np.asarray([float(a["depth"].mean()) for a in added]),
          "added_pointmap_mean": torch.stack([a["pointmap"][0].double().mean(dim=(0, 1)) for a in added]).numpy(),
          "added_conf_mean": np.asarray([float(a["conf"].mean()) for a in added]), "traj": np.asarray(traj, np.float64)[:k]}
    np.savez_compressed(os.path.join(HERE, "terminate.npz"), **fx)
    print("wrote terminate: keyframes", fx["keyframes"].tolist(), "tracked", int(tr.t1), "| extra views at", fx["added_tstamp"].tolist(), "submaps", fx["added_sub"].tolist(),
          "| image handed over", fx["added_img_shape"][0].tolist())


def gen_loop_production_fixture():
    """The same per-frame loop (kfFilter + TrackFrontend.run of the reference, hi2.py:101-111) AT PRODUCTION SHAPE: the reference's own
    ViT-L / 768-d dual decoder / DPT model (cut3r_slam_amd.config.production_config, seeded weights) on 384x512 frames, 33 frames at
    kf_every = 2 -> 18 keyframes, three six-view windows and the closing window -- the stream of tests/test_e2e_production_gpu.py.
    Takes a few minutes of CPU: NOT in the default list, run `python tests/golden/make_fixtures.py loop_production`.  Stored: keyframes, window calls, final poses, edge list, and strided samples of the depths and submaps."""
    AR, ARCfg, inference = import_reference_model()
    import_reference_graph()
    import motion_filter as MF
    from keyframe import KeyFrame
    from track_frontend import TrackFrontend
    from factor_graph import FactorGraph
    from torch.multiprocessing import Value
    from cut3r_slam_amd import synth
    from cut3r_slam_amd.config import production_config
    cfg = production_config()
    sd = synth.tracking_state_dict(cfg, 0, enc_residual_gain=0.1)
    torch.manual_seed(0)
    model = AR(ref_config(ARCfg, cfg)).eval()
    print("production: load_state_dict:", torch.nn.Module.load_state_dict(model, sd, strict=True))
    H, W = 384, 512
    n, buffer, ds = 33, 40, 2
    frames = synth.pan_stream(n, H, W, pool=9, num=6, den=1, seed=0)
    intr = torch.tensor([600.0 * W / 1200.0, 600.0 * H / 680.0, 599.5 * W / 1200.0, 339.5 * H / 680.0])
    kf = object.__new__(KeyFrame)
    kf.counter, kf.ready, kf.is_initialized, kf.downsample_ratio = Value("i", 0), Value("i", 0), False, ds
    kf.tstamp = torch.zeros(buffer)
    kf.image = torch.zeros(buffer, 3, H, W, dtype=torch.uint8)
    kf.intrinsic, kf.pose, kf.depth = torch.zeros(buffer, 4), torch.zeros(buffer, 7), torch.ones(buffer, H, W)
    kf.pose[:] = torch.as_tensor([0, 0, 0, 0, 0, 0, 1.0])
    kf.submap_ds = torch.ones(buffer // 5, 6, H // ds, W // ds, 3)
    kf.conf_ds = torch.zeros(buffer // 5, 6, H // ds, W // ds)
    kf.featI = torch.zeros(buffer, (H // 16) * (W // 16), cfg.enc_embed_dim)
    kf.pos = torch.zeros(buffer, (H // 16) * (W // 16), 2, dtype=torch.int64)
    filt = MF.MotionFilter(model, kf, {"thresh": 0.9, "skip": 1, "kf_every": 2, "skip_blur": False}, device="cpu")
    graph = FactorGraph(kf, device="cpu", max_factors=48)
    tr = object.__new__(TrackFrontend)
    tr.device, tr.keyframes, tr.model, tr.graph = "cpu", kf, model, graph
    tr.verbose, tr.output_dir, tr.use_gt, tr.conf_th, tr.downsample_ratio, tr.t1, tr.warmup = False, None, False, 0.5, ds, 0, 6
    real_to = torch.Tensor.to

    def to_cpu(self, *a, **k):
        a = tuple("cpu" if isinstance(x, str) and x.startswith("cuda") else x for x in a)
        return real_to(self, *a, **k)
    calls = []
    torch.Tensor.to = to_cpu
    import time
    t_start = time.time()
    try:
        with torch.no_grad():
            for t in range(n):
                filt.kfFilter(t, frames[t:t + 1], intrinsics=intr, second_last_frame=(t == n - 2), last_frame=(t == n - 1))
                flag, rng, sub = tr.run(t, last_frame=(t == n - 1))
                if rng is not None:
                    calls.append([t, int(bool(flag)), rng.start, rng.stop, int(sub)])
                    print(f"  window {calls[-1]} after {time.time() - t_start:.0f} s", flush=True)
    finally:
        torch.Tensor.to = real_to
    k, t1 = kf.counter.value, tr.t1
    nsub = (t1 - 1) // 5 + 1
    fx = {"frames_sum": np.int64(int(frames.long().sum())), "intrinsic": intr.numpy(), "calls": np.asarray(calls, np.int64),
          "keyframes": kf.tstamp[:k].numpy().astype(np.int64), "t1": np.int64(t1), "pose": kf.pose[:t1].numpy().copy(),
          "depth_samples": kf.depth[:t1, 8::24, 8::32].numpy().copy(), "depth_mean": kf.depth[:t1].double().mean(dim=(1, 2)).numpy(),
          "submap_samples": kf.submap_ds[:nsub, :, 4::12, 4::16].numpy().copy(), "conf_mean": kf.conf_ds[:nsub].double().mean(dim=(2, 3)).numpy(),
          "ii": graph.ii.numpy().copy(), "jj": graph.jj.numpy().copy()}
    np.savez_compressed(os.path.join(HERE, "loop_production.npz"), **fx)
    print("wrote loop_production: keyframes", k, "tracked", t1, "calls", calls, "| edges", len(fx["ii"]), f"| {time.time() - t_start:.0f} s")


def gen_motion_filter_fixture():
    """MotionFilter.kfFilter (hislam2/motion_filter.py:70-135) ITSELF on the CPU over two seeded streams: overlap mode (kf_every = -1,
    skip = 2, thresh = 0.9: a slideshow whose content changes every 4 frames) and fixed cadence (kf_every = 3), both with the
    second-last / last frame flags of demo_s.py.  Recorded: the time stamps that became keyframes and every overlap ratio the filter
    computed (a spy around the module's compute_patch_overlap_ratio).  Adapters: the keyframe store is the reference's KeyFrame class
    created without its __init__ (which allocates on "cuda") and given CPU tensors; `.to('cuda')` is redirected to the CPU."""
    AR, ARCfg, inference = import_reference_model()
    import_reference_graph()
    import motion_filter as MF
    from keyframe import KeyFrame
    from torch.multiprocessing import Value
    from cut3r_slam_amd import synth
    cfg = synth.medium_config()
    seed = 11
    sd = synth.tracking_state_dict(cfg, seed)
    torch.manual_seed(0)
    model = AR(ref_config(ARCfg, cfg)).eval()
    torch.nn.Module.load_state_dict(model, sd, strict=True)
    H, W = cfg.img_size
    real_to, real_ratio = torch.Tensor.to, MF.compute_patch_overlap_ratio

    def to_cpu(self, *a, **k):
        a = tuple("cpu" if isinstance(x, str) and x.startswith("cuda") else x for x in a)
        return real_to(self, *a, **k)
    fx = {"seed": np.int64(seed)}
    torch.Tensor.to = to_cpu
    try:
        for name, frames, mf in (("overlap", synth.slideshow_stream(41, H, W, hold=4, seed=3), {"thresh": 0.9, "skip": 2, "kf_every": -1}),
                                 ("blend", synth.blend_stream(40, H, W, period=12, seed=5), {"thresh": 0.9, "skip": 1, "kf_every": -1}),
                                 ("cadence", synth.pan_stream(20, H, W, pool=5, num=2, den=1, seed=0), {"thresh": 0.9, "skip": 1, "kf_every": 3})):
            n, buffer = frames.shape[0], 48
            kf = object.__new__(KeyFrame)
            kf.counter, kf.ready = Value("i", 0), Value("i", 0)
            kf.tstamp = torch.zeros(buffer)
            kf.image = torch.zeros(buffer, 3, H, W, dtype=torch.uint8)
            kf.intrinsic, kf.pose, kf.depth = torch.zeros(buffer, 4), torch.zeros(buffer, 7), torch.ones(buffer, H, W)
            kf.featI = torch.zeros(buffer, (H // 16) * (W // 16), cfg.enc_embed_dim)
            kf.pos = torch.zeros(buffer, (H // 16) * (W // 16), 2, dtype=torch.int64)
            filt = MF.MotionFilter(model, kf, dict(mf, skip_blur=False), device="cpu")
            ratios = []

            def spy(f0, f1, _r=ratios, **k):
                r = real_ratio(f0, f1, **k)
                _r.append(float(r))
                return r
            MF.compute_patch_overlap_ratio = spy
            intr = torch.tensor([80.0, 80.0, 47.5, 31.5])
            with torch.no_grad():
                for t in range(n):
                    filt.kfFilter(t, frames[t:t + 1], intrinsics=intr, second_last_frame=(t == n - 2), last_frame=(t == n - 1))
            k = kf.counter.value
            fx[f"{name}_frames_sum"] = np.int64(int(frames.long().sum()))
            fx[f"{name}_keyframes"] = kf.tstamp[:k].numpy().astype(np.int64)
            fx[f"{name}_ratios"] = np.asarray(ratios, np.float64)
            fx[f"{name}_feat_last"] = kf.featI[k - 1].numpy().copy()
            print(name, "keyframes", fx[f"{name}_keyframes"].tolist(), "ratios", np.round(fx[f"{name}_ratios"], 3).tolist())
    finally:
        torch.Tensor.to = real_to
        MF.compute_patch_overlap_ratio = real_ratio
    np.savez_compressed(os.path.join(HERE, "motion_filter.npz"), **fx)


if __name__ == "__main__":
    what = sys.argv[1:] or ["rope", "graph", "dpt", "linear", "medium", "nms", "camera", "frontend", "motion_filter", "loop", "backend", "chol", "gs_utils", "gaussian_model", "handover", "terminate"]
    if "motion_filter" in what:
        gen_motion_filter_fixture()
    if "loop" in what:
        gen_loop_fixture()
    if "backend" in what:
        gen_backend_fixture()
    if "chol" in what:
        gen_chol_fixture()
    if "gs_utils" in what:
        gen_gs_utils_fixture()
    if "gaussian_model" in what:
        gen_gaussian_model_fixture()
    if "handover" in what:
        gen_handover_fixture()
    if "terminate" in what:
        gen_terminate_fixture()
    if "loop_production" in what:
        gen_loop_production_fixture()
    if "backend_production" in what:
        gen_backend_fixture(production=True)
    if "frontend" in what:
        gen_frontend_fixture()
    if "nms" in what:
        gen_nms_fixture()
    if "camera" in what:
        gen_camera_fixture()
    if "tf32_budgets" in what:
        gen_tf32_budgets()
    if "rope" in what:
        gen_rope_fixture()
    if "graph" in what:
        gen_graph_fixture()
    if "dpt" in what:
        gen_model_fixture("model_tiny_dpt", tiny_config("dpt"), seed=3, n_views=3)
    if "linear" in what:
        gen_model_fixture("model_tiny_linear", tiny_config("linear"), seed=5, n_views=2)
    if "medium" in what:
        # the production head widths (64-wide encoder / image-side heads, 48-wide state-side heads) at a size the reference runs in
        # seconds on the CPU: pins the oracle and the HIP path for those widths against the reference itself
        gen_model_fixture("model_medium_dpt", Cut3rConfig(img_size=(64, 96), enc_embed_dim=256, enc_depth=3, enc_num_heads=4, dec_embed_dim=192,
                                                          dec_depth=4, dec_num_heads=3, state_dec_num_heads=4, state_size=30, local_mem_size=16,
                                                          ray_enc_depth=1, head_type="dpt", rgb_head=True), seed=11, n_views=3)
