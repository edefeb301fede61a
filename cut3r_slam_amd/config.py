"""Model configuration + reference state_dict schema for the CUT3R pointmap network.

The schema mirrors the parameter names/shapes of the reference module tree
(/root/reference/src/dust3r/model.py:225-303 `ARCroco3DStereo.__init__`,
 /root/reference/src/croco/models/croco.py:59-145 `CroCoNet.__init__`,
 /root/reference/src/dust3r/heads/dpt_head.py:138-211, linear_head.py:246-297,
 /root/reference/src/croco/models/dpt_block.py:281-480) so that a reference checkpoint's
``ckpt["model"]`` loads into this runtime unchanged.  tests/golden/make_fixtures.py asserts this
schema equals ``ARCroco3DStereo(cfg).state_dict()`` key-for-key, shape-for-shape.
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass, asdict, field
from typing import Dict, Tuple


@dataclass
class Cut3rConfig:
    img_size: Tuple[int, int] = (384, 512)      # (H, W) the head was built for; inference takes any multiple of 16
    patch_size: int = 16
    enc_embed_dim: int = 1024
    enc_depth: int = 24
    enc_num_heads: int = 16
    dec_embed_dim: int = 768
    dec_depth: int = 12
    dec_num_heads: int = 12
    state_dec_num_heads: int = 16                # model.py:112
    state_size: int = 768
    local_mem_size: int = 256
    ray_enc_depth: int = 2
    ray_enc_num_heads: int = 16                  # hard-coded 16 at model.py:245
    mlp_ratio: int = 4
    rope_freq: float = 100.0                     # pos_embed="RoPE100"
    head_type: str = "dpt"                       # "dpt" | "linear"
    rgb_head: bool = True
    pose_head: bool = True
    ln_eps: float = 1e-6
    # DPT constants (dpt_head.py:153-156, dpt_block.py:300-303)
    dpt_layer_dims: Tuple[int, int, int, int] = (96, 192, 384, 768)
    dpt_feature_dim: int = 256
    dpt_last_dim: int = 128

    def to_dict(self):
        return asdict(self)

    @staticmethod
    def from_dict(d):
        d = dict(d)
        for k in ("img_size", "dpt_layer_dims"):
            if k in d:
                d[k] = tuple(d[k])
        return Cut3rConfig(**d)

    @property
    def state_width(self) -> int:
        """2-D state-token grid width (model.py:553-555)."""
        w = int(self.state_size ** 0.5)
        return w + 1 if w % 2 == 1 else w


def production_config() -> Cut3rConfig:
    """Assumed config of ./checkpoints/cut3r_512_dpt_4_64.pth."""
    return Cut3rConfig()


def config1_224() -> Cut3rConfig:
    """BASELINE config 1: the reference's only in-tree config (model.py:1120-1137)."""
    return Cut3rConfig(img_size=(224, 224), state_size=256, head_type="linear", rgb_head=True)


def tiny_config(head_type: str = "dpt") -> Cut3rConfig:
    """Small config used for golden fixtures (DPT conv widths are fixed by the head class)."""
    return Cut3rConfig(img_size=(32, 48), enc_embed_dim=64, enc_depth=2, enc_num_heads=4,
                       dec_embed_dim=48, dec_depth=4, dec_num_heads=3, state_dec_num_heads=3,
                       state_size=12, local_mem_size=8, ray_enc_depth=1, ray_enc_num_heads=16,
                       head_type=head_type, rgb_head=True)


# ------------------------------------------------------------------------------------------------
def _lin(s, name, n_out, n_in, bias=True):
    s[name + ".weight"] = (n_out, n_in)
    if bias:
        s[name + ".bias"] = (n_out,)


def _ln(s, name, d):
    s[name + ".weight"] = (d,)
    s[name + ".bias"] = (d,)


def _enc_block(s, p, d, r):
    _ln(s, p + ".norm1", d)
    _lin(s, p + ".attn.qkv", 3 * d, d)
    _lin(s, p + ".attn.proj", d, d)
    _ln(s, p + ".norm2", d)
    _lin(s, p + ".mlp.fc1", r * d, d)
    _lin(s, p + ".mlp.fc2", d, r * d)


def _dec_block(s, p, d, r):
    # module registration order of DecoderBlock (dust3r/blocks.py:262-290)
    _ln(s, p + ".norm1", d)
    _lin(s, p + ".attn.qkv", 3 * d, d)
    _lin(s, p + ".attn.proj", d, d)
    _lin(s, p + ".cross_attn.projq", d, d)
    _lin(s, p + ".cross_attn.projk", d, d)
    _lin(s, p + ".cross_attn.projv", d, d)
    _lin(s, p + ".cross_attn.proj", d, d)
    _ln(s, p + ".norm2", d)
    _ln(s, p + ".norm3", d)
    _lin(s, p + ".mlp.fc1", r * d, d)
    _lin(s, p + ".mlp.fc2", d, r * d)
    _ln(s, p + ".norm_y", d)


def _modln_block(s, p, d, r):
    # ConditionModulationBlock (dust3r/blocks.py:382-420)
    _ln(s, f"{p}.norm1.norm", d)
    _lin(s, f"{p}.norm1.mlp.1", 2 * d, d)
    _lin(s, f"{p}.attn.qkv", 3 * d, d)
    _lin(s, f"{p}.attn.proj", d, d)
    _ln(s, f"{p}.norm2.norm", d)
    _lin(s, f"{p}.norm2.mlp.1", 2 * d, d)
    _lin(s, f"{p}.mlp.fc1", r * d, d)
    _lin(s, f"{p}.mlp.fc2", d, r * d)


def _conv(s, name, cout, cin, k, bias=True):
    s[name + ".weight"] = (cout, cin, k, k)
    if bias:
        s[name + ".bias"] = (cout,)


def _dpt(s, p, cfg: Cut3rConfig, nch: int):
    E, D = cfg.enc_embed_dim, cfg.dec_embed_dim
    ld, F, L = cfg.dpt_layer_dims, cfg.dpt_feature_dim, cfg.dpt_last_dim
    toks = (E, D, D, D)
    for i in range(4):
        _conv(s, f"{p}.scratch.layer{i+1}_rn", F, ld[i], 3, bias=False)
    for i in range(4):   # layer_rn ModuleList aliases the same tensors (both names are in state_dict)
        _conv(s, f"{p}.scratch.layer_rn.{i}", F, ld[i], 3, bias=False)
    for r in (1, 2, 3, 4):
        q = f"{p}.scratch.refinenet{r}"
        _conv(s, q + ".out_conv", F, F, 1)
        for u in ("resConfUnit1", "resConfUnit2"):
            _conv(s, f"{q}.{u}.conv1", F, F, 3)
            _conv(s, f"{q}.{u}.conv2", F, F, 3)
    _conv(s, f"{p}.head.0", F // 2, F, 3)
    _conv(s, f"{p}.head.2", L, F // 2, 3)
    _conv(s, f"{p}.head.4", nch, L, 1)
    # act_postprocess (ModuleList; act_{1..4}_postprocess aliases are deleted by DPTOutputAdapter_fix.init)
    a = f"{p}.act_postprocess"
    _conv(s, f"{a}.0.0", ld[0], toks[0], 1)
    s[f"{a}.0.1.weight"] = (ld[0], ld[0], 4, 4)      # ConvTranspose2d weight is (Cin, Cout, k, k)
    s[f"{a}.0.1.bias"] = (ld[0],)
    _conv(s, f"{a}.1.0", ld[1], toks[1], 1)
    s[f"{a}.1.1.weight"] = (ld[1], ld[1], 2, 2)
    s[f"{a}.1.1.bias"] = (ld[1],)
    _conv(s, f"{a}.2.0", ld[2], toks[2], 1)
    _conv(s, f"{a}.3.0", ld[3], toks[3], 1)
    _conv(s, f"{a}.3.1", ld[3], ld[3], 3)


def state_dict_schema(cfg: Cut3rConfig) -> "OrderedDict[str, Tuple[int, ...]]":
    """key -> shape for every tensor in the reference ``state_dict()`` of this config."""
    E, D, r, P = cfg.enc_embed_dim, cfg.dec_embed_dim, cfg.mlp_ratio, cfg.patch_size
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    if cfg.pose_head:
        s["pose_token"] = (1, 1, D)
    s["masked_img_token"] = (1, E)
    s["masked_ray_map_token"] = (1, E)
    _conv(s, "patch_embed.proj", E, 3, P)
    _conv(s, "patch_embed_ray_map.proj", E, 6, P)
    for i in range(cfg.enc_depth):
        _enc_block(s, f"enc_blocks.{i}", E, r)
    _ln(s, "enc_norm", E)
    _lin(s, "decoder_embed", D, E)
    for i in range(cfg.dec_depth):
        _dec_block(s, f"dec_blocks.{i}", D, r)
    _ln(s, "dec_norm", D)
    for i in range(cfg.ray_enc_depth):
        _enc_block(s, f"enc_blocks_ray_map.{i}", E, 4)
    _ln(s, "enc_norm_ray_map", E)
    if cfg.pose_head:
        s["pose_retriever.masked_token"] = (1, 1, D)
        s["pose_retriever.mem"] = (1, cfg.local_mem_size, 2 * D)
        _lin(s, "pose_retriever.proj_q", D, E)
        for i in range(2):
            _dec_block(s, f"pose_retriever.write_blocks.{i}", 2 * D, 4)
        for i in range(2):
            _dec_block(s, f"pose_retriever.read_blocks.{i}", 2 * D, 4)
    s["register_tokens.weight"] = (cfg.state_size, E)
    _lin(s, "decoder_embed_state", D, E)
    for i in range(cfg.dec_depth):
        _dec_block(s, f"dec_blocks_state.{i}", D, r)
    _ln(s, "dec_norm_state", D)
    h = "downstream_head"
    if cfg.head_type == "dpt":
        _dpt(s, f"{h}.dpt_self", cfg, 4)
        for i in range(2):
            _modln_block(s, f"{h}.final_transform.{i}", D, 4)
        _dpt(s, f"{h}.dpt_cross", cfg, 4)
        if cfg.rgb_head:
            _dpt(s, f"{h}.dpt_rgb", cfg, 3)
        _lin(s, f"{h}.pose_head.mlp.fc1", 4 * D, D)
        _lin(s, f"{h}.pose_head.mlp.fc2", 7, 4 * D)
    elif cfg.head_type == "linear":
        _lin(s, f"{h}.proj.fc1", 4 * D, D)
        _lin(s, f"{h}.proj.fc2", 4 * P * P, 4 * D)
        if cfg.rgb_head:
            _lin(s, f"{h}.rgb_proj.fc1", 4 * D, D)
            _lin(s, f"{h}.rgb_proj.fc2", 3 * P * P, 4 * D)
        _lin(s, f"{h}.pose_head.mlp.fc1", 4 * D, D)
        _lin(s, f"{h}.pose_head.mlp.fc2", 7, 4 * D)
        for i in range(2):
            _modln_block(s, f"{h}.final_transform.{i}", D, 4)
        _lin(s, f"{h}.cross_proj.fc1", 4 * D, D)
        _lin(s, f"{h}.cross_proj.fc2", 4 * P * P, 4 * D)
    else:
        raise ValueError(cfg.head_type)
    return s
