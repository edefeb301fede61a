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
                        head_gain: float = 0.3):
    sd = synth_state_dict(cfg, seed)
    if cfg.head_type == "dpt":
        k = "downstream_head.dpt_self.head.4."
        sd[k + "weight"] = sd[k + "weight"] * head_gain
        b = sd[k + "bias"].clone()
        b[2] = depth_logit
        sd[k + "bias"] = b
    if enc_residual_gain is not None:
        for key in list(sd):
            if key.startswith("enc_blocks.") and key.rsplit(".", 1)[0].endswith(("attn.proj", "mlp.fc2")):
                sd[key] = sd[key] * enc_residual_gain
    return sd


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
