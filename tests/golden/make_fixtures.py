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

Harness-side adapters (nothing in the reference is modified):
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
    np.savez_compressed(os.path.join(HERE, "camera.npz"), **fx)
    print("wrote camera", {k: v.shape for k, v in fx.items()})


if __name__ == "__main__":
    what = sys.argv[1:] or ["rope", "graph", "dpt", "linear", "medium", "nms", "camera"]
    if "nms" in what:
        gen_nms_fixture()
    if "camera" in what:
        gen_camera_fixture()
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
