"""Seeded synthetic inputs for tests and bench (`"data": "synthetic"`): image streams and tracking-friendly random weights.

No checkpoint and no dataset exist in the reference tree, so the end-to-end runs need inputs for which the
tracking loop is well defined with RANDOM weights:
  * depth must be positive (the tracker takes log(depth), hislam2/track_frontend.py:216): the z-bias of the self-view DPT head
    is raised and its last 1x1 convolution damped, so pointmaps sit at ~3.5 m with small relief;
  * in overlap mode the keyframe test thresholds patch-feature cosine similarity at 0.7 (hislam2/util/utils.py:726-736); a
    deep random-init ViT collapses every token onto a common direction (similarity ~1 between unrelated images), so
    `enc_residual_gain` damps the residual branches of the encoder blocks: features then stay content dependent, like a
    trained encoder's, and the test separates "same texture shifted by 1 px" (ratio ~1) from "shifted by 2 px" (ratio ~0).
The arithmetic cost of the network does not depend on these values.
"""
from __future__ import annotations

import torch

from .config import Cut3rConfig
from .weights import synth_state_dict


def medium_config(head_type: str = "dpt") -> Cut3rConfig:
    """production head widths (64 / 48 / 128-wide heads) at a size the CPU oracle runs in a second per window"""
    return Cut3rConfig(img_size=(64, 96), enc_embed_dim=256, enc_depth=3, enc_num_heads=4, dec_embed_dim=192, dec_depth=4,
                       dec_num_heads=3, state_dec_num_heads=4, state_size=30, local_mem_size=16, ray_enc_depth=1,
                       head_type=head_type, rgb_head=True)


def tracking_state_dict(cfg: Cut3rConfig, seed: int = 0, enc_residual_gain: float | None = None, depth_logit: float = 1.5,
                        head_gain: float = 0.3, depth_relief: float = 1.0):
    """depth_relief < 1 damps the z logit's dependence on the features (its bias stays): a random-weight network predicts the SAME
    image's depth a constant factor apart as view 0 and as view 5 of a window (the recurrent state differs), which a trained network
    does not; chained over hundreds of windows of an uncut stream that factor compounds (log s grows linearly, see
    track_frontend._check_scale).  With a small relief the per-window ratio averages ~1 and an endless synthetic stream stays in
    the fp32 range.  The arithmetic cost does not depend on it."""
    sd = synth_state_dict(cfg, seed)
    if cfg.head_type == "dpt":
        k = "downstream_head.dpt_self.head.4."
        w = sd[k + "weight"] * head_gain
        if depth_relief != 1.0:
            w = w.clone()
            w[2] = w[2] * depth_relief
        sd[k + "weight"] = w
        b = sd[k + "bias"].clone()
        b[2] = depth_logit
        sd[k + "bias"] = b
    if enc_residual_gain is not None:
        for key in list(sd):
            if key.startswith("enc_blocks.") and key.rsplit(".", 1)[0].endswith(("attn.proj", "mlp.fc2")):
                sd[key] = sd[key] * enc_residual_gain
    return sd


def outlier_state_dict(cfg: Cut3rConfig, seed: int = 0, fc1_gain: float = 6.0e3, fc2_weight: float = 0.1, norm_gain: float = 8.0, **kw):
    """Random weights with injected MASSIVE ACTIVATIONS, the feature of trained ViT checkpoints that random initialisation lacks (no
    checkpoint exists in the reference tree and none may be fetched): in a few blocks one hidden unit h of the MLP gets its `fc1` row
    multiplied by `fc1_gain` (hidden activations ~1e4 at the tokens that excite it) and feeds one residual channel c through
    `fc2.weight[c, h] = fc2_weight` (the fp32 residual stream reaches ~1e3 in that channel, token dependent, and keeps it through
    every later block); one LayerNorm channel per affected stack gets its gain multiplied by `norm_gain`.  Like a trained head, the
    self-view DPT adapter does not read the massive decoder channels (its 1x1 input convolutions' columns for them are zero) --
    otherwise the exp() of the output activation turns the outlier into astronomically large coordinates, in any arithmetic.
    Returns (state_dict, plan) with plan = [(block prefix, hidden unit, residual channel)]."""
    sd = tracking_state_dict(cfg, seed, **kw)
    plan = []

    def inject(prefix, h, c):
        w1 = sd[prefix + ".mlp.fc1.weight"].clone()
        w1[h] = w1[h] * fc1_gain
        sd[prefix + ".mlp.fc1.weight"] = w1
        w2 = sd[prefix + ".mlp.fc2.weight"].clone()
        w2[:, h] = 0                                      # the massive unit feeds ONE channel (its fan-out is concentrated, as trained)
        w2[c, h] = fc2_weight
        sd[prefix + ".mlp.fc2.weight"] = w2
        plan.append((prefix, h, c))

    E, D = cfg.enc_embed_dim, cfg.dec_embed_dim
    inject(f"enc_blocks.{max(1, cfg.enc_depth // 6)}", 7 % (4 * E), 5 % E)
    inject(f"enc_blocks.{max(1, (2 * cfg.enc_depth) // 3)}", 100 % (4 * E), 321 % E)
    c_img, c_state = 11 % D, 200 % D
    inject(f"dec_blocks.{min(1, cfg.dec_depth - 1)}", 9 % (4 * D), c_img)
    inject(f"dec_blocks_state.{min(2, cfg.dec_depth - 1)}", 33 % (4 * D), c_state)
    for key, c in ((f"enc_blocks.{cfg.enc_depth - 1}.norm1.weight", 5 % E), (f"dec_blocks.{cfg.dec_depth - 1}.norm1.weight", c_img),
                   (f"dec_blocks_state.{cfg.dec_depth - 1}.norm1.weight", c_state)):
        g = sd[key].clone()
        g[c] = g[c] * norm_gain
        sd[key] = g
    if cfg.head_type == "dpt":
        for i in (1, 2):                                # hooks dec6 / dec9: raw residual streams of the image-side decoder
            k = f"downstream_head.dpt_self.act_postprocess.{i}.0.weight"
            w = sd[k].clone()
            w[:, c_img] = 0
            sd[k] = w
    return sd, plan


def pan_stream(n: int, H: int, W: int, pool: int = 5, num: int = 2, den: int = 1, seed: int = 0, device="cpu") -> torch.Tensor:
    """u8 [n,3,H,W]: a seeded random texture (box-filtered with `pool`, stretched to 0..255) seen through a window that pans
    `num` pixels every `den` frames horizontally and half of that vertically."""
    g = torch.Generator().manual_seed(seed)
    span = (n * num) // den + 2
    base = torch.rand(1, 3, H + span // 2 + 2, W + span, generator=g)
    if pool > 1:
        base = torch.nn.functional.avg_pool2d(base, pool, 1, pool // 2)
    base = ((base - base.min()) / (base.max() - base.min()) * 255).round().to(torch.uint8)[0].to(device)
    out = torch.empty(n, 3, H, W, dtype=torch.uint8, device=device)
    for t in range(n):
        dx = (t * num) // den
        out[t] = base[:, dx // 2:dx // 2 + H, dx:dx + W]
    return out


def slideshow_stream(n: int, H: int, W: int, hold: int = 10, seed: int = 0, device="cpu") -> torch.Tensor:
    """u8 [n,3,H,W] for the overlap-mode keyframe test: an unrelated white-noise texture every `hold` frames (patch overlap
    with the previous keyframe ~0 -> keyframe), and inside a hold the same texture with a brightness drift of one grey level
    per frame (patch overlap ~1 -> no keyframe).  The decisions are driven by the image content and are far from the
    thresholds (0.7 on the cosine, `thresh` on the ratio), so they do not depend on the arithmetic precision of the encoder."""
    g = torch.Generator().manual_seed(seed)
    out = torch.empty(n, 3, H, W, dtype=torch.uint8, device=device)
    tex = None
    for t in range(n):
        if t % hold == 0:
            tex = torch.randint(8, 240, (3, H, W), generator=g, dtype=torch.int16).to(device)
        out[t] = (tex + (t % hold)).to(torch.uint8)
    return out


def blend_stream(n: int, H: int, W: int, period: int = 12, seed: int = 0, device="cpu") -> torch.Tensor:
    """u8 [n,3,H,W]: cross-fades between unrelated white-noise textures, one fade per `period` frames: the patch-overlap ratio against
    the last keyframe falls gradually from 1 to 0 along a fade, so an overlap-mode motion filter passes through INTERMEDIATE ratios
    (the slideshow stream only produces 0 and 1)."""
    g = torch.Generator().manual_seed(seed)
    out = torch.empty(n, 3, H, W, dtype=torch.uint8, device=device)
    a = torch.randint(8, 240, (3, H, W), generator=g).float()
    b = torch.randint(8, 240, (3, H, W), generator=g).float()
    for t in range(n):
        k = t % period
        if t > 0 and k == 0:
            a, b = b, torch.randint(8, 240, (3, H, W), generator=g).float()
        w = k / period
        out[t] = ((1 - w) * a + w * b).round().to(torch.uint8).to(device)
    return out


def gs_wall_window(H: int = 384, W: int = 512, focal: float = 440.0, n_views: int = 6, device="cuda:0"):
    """A synthetic keyframe window for the GS mapper: a textured, gently curved wall modelled by one ground-truth Gaussian per pixel,
    rendered through the HIP rasteriser from `n_views` poses.  Returns (packet for GSMapper.run, images u8 [n,3,H,W], config dict)."""
    from . import gs_mapper as GM
    from .lietorch import SE3
    cfg = {"Training": {"lambda_depth": 10.0, "lambda_normal": 0.1, "lambda_iso": 10.0, "gaussian_th": 0.05, "gaussian_extent": 1.0, "size_threshold": 20,
                        "window_size": 10},
           "opt_params": {"pose_lr": 0.0001, "position_lr_init": 0.0005, "feature_lr": 0.005, "opacity_lr": 0.05, "scaling_lr": 0.001,
                          "rotation_lr": 0.001, "percent_dense": 0.01, "densify_grad_threshold": 0.0005}}
    ys, xs = torch.meshgrid(torch.linspace(-1.7, 1.7, H), torch.linspace(-2.3, 2.3, W), indexing="ij")
    z = 3.0 + 0.2 * torch.sin(xs) * torch.cos(1.3 * ys)
    col = torch.stack([0.5 + 0.4 * torch.sin(3 * xs), 0.5 + 0.4 * torch.cos(2.5 * ys), 0.5 + 0.4 * torch.sin(2 * xs + 3 * ys)], -1)
    truth = GM.GaussianMap(cfg["opt_params"], device)
    truth.extend_from_pcd_seq(0, rgb=col.reshape(-1, 3), pointmap=torch.stack([xs, ys, z], -1).reshape(-1, 3))
    with torch.no_grad():
        truth.p["opacity"].fill_(2.2)
        truth.p["scaling"] += 0.26
    poses = [SE3.exp(torch.tensor([[0.05 * k, 0.01 * (k % 2), 0.0, 0.0, -0.01 * k, 0.0]], device=device)).data[0].cpu() for k in range(n_views)]
    imgs, depths, pms = [], [], []
    yy, xx = torch.meshgrid(torch.arange(H, device=device).float(), torch.arange(W, device=device).float(), indexing="ij")
    for p in poses:
        T = GM.pose_vec_to_matrix(p[None].to(device))[0]
        cam = GM.Camera(0, torch.zeros(3, H, W), torch.ones(H, W), torch.inverse(T), focal, focal, W / 2, H / 2, device=device)
        with torch.no_grad():
            pkg = GM.render(cam, truth, torch.zeros(3, device=device))
        d = pkg["depth"][0]
        imgs.append((pkg["render"].clamp(0, 1) * 255).round().to(torch.uint8))
        depths.append(d)
        pc = torch.stack([(xx - W / 2) / focal * d, (yy - H / 2) / focal * d, d], -1)
        pms.append((pc @ T[:3, :3].T + T[:3, 3])[::2, ::2])
    packet = {"viz_idx": list(range(n_views)), "submap_idx": 0, "tstamp": torch.arange(n_views).float(), "poses": torch.stack(poses),
              "images": torch.stack(imgs), "pointmaps": torch.stack(pms), "confs": torch.ones(n_views, H // 2, W // 2, device=device),
              "depths": torch.stack(depths), "intrinsics": torch.tensor([focal, focal, W / 2, H / 2])}
    return packet, torch.stack(imgs), cfg


def gs_mapper_window_leg(H: int = 384, W: int = 512, device="cuda:0", use_graphs: bool = False, fused: bool = True):
    """one synthetic 6-keyframe window through GSMapper.run with the reference's iteration counts (gs_backend_per_frame.py:776-862: 100
    initial, per new keyframe 50 pose-refine + 20 window + 50 single-view, 10 per view global) -> timing / quality figures"""
    import time
    from . import gs_mapper as GM
    packet, imgs, cfg = gs_wall_window(H, W, device=device)
    n = len(packet["viz_idx"])
    times = []
    for _ in range(2):                                # the first pass pays the one-time costs of the process (code objects, allocator pools)
        mapper = GM.GSMapper(cfg, float(packet["intrinsics"][0]), float(packet["intrinsics"][1]), W / 2, H / 2, downsample_ratio=2, device=device)
        mapper.use_graphs, mapper.graph_min_iters = use_graphs, 12
        mapper.fused = fused
        torch.cuda.synchronize()
        t0 = time.time()
        with torch.enable_grad():
            mapper.run(packet, iterations=100)
        torch.cuda.synchronize()
        times.append(time.time() - t0)
    dt = times[1]
    with torch.no_grad():
        ps = []
        for k in range(n):
            r = GM.render(mapper.viewpoints[k], mapper.gaussians, torch.zeros(3, device=device))["render"]
            ps.append(float(-10 * torch.log10(((r - imgs[k].float() / 255) ** 2).mean())))
    renders = 100 + (n - 1) * (50 + 50) + sum(20 * min(k + 1, 10) for k in range(1, n)) + 10 * n
    return {"config": f"synthetic wall, {n} keyframes at {W}x{H}, one Gaussian per stride-2 pixel of the first keyframe, the reference's iteration counts"
                      + (", iterations without densification replayed from a captured hipGraph" if use_graphs else ""),
            "trainer": "tape-free (gs_step.FusedTrainer: direct C-ABI calls)" if fused else "tensor-op formulation with autograd",
            "seconds": round(dt, 3), "seconds_first_pass_in_process": round(times[0], 3), "ms_per_keyframe": round(1e3 * dt / n, 1),
            "realtime_budget_ms_per_keyframe_at_30fps_kf_every_10": 333.3, "render_iterations": renders,
            "ms_per_render_iteration": round(1e3 * dt / renders, 3), "gaussians": len(mapper.gaussians), "psnr_db": round(sum(ps) / n, 2)}


def loop_state_dict(cfg: Cut3rConfig, seed: int = 0, pose_gain: float = 0.05, **kw):
    """Random weights for which the loop-closure backend fires by itself: the pose head's last layer is damped by `pose_gain`, so the
    predicted cameras of a window sit within centimetres of each other and every keyframe sees the same stretch of the (random)
    pointmaps.  The covisibility graph then links keyframes more than 8 apart (`FactorGraph.detect_loop`), the NMS score passes 0.4
    and `TrackBackend.run` closes a loop every other eligible window -- detect -> NMS -> re-track -> optimise -> rewrite, nothing
    forced.  (A random-weight network carries no geometry: its poses do not depend on where the camera really is, so a stream that
    "revisits" a place would not be recognised; this variant makes EVERY place the same one.)"""
    sd = tracking_state_dict(cfg, seed, **kw)
    k = "downstream_head.pose_head.mlp.fc2.weight"
    sd[k] = sd[k] * pose_gain
    return sd
