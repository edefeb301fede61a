"""TEST INFRASTRUCTURE ONLY -- CPU (PyTorch fp32) restatement of the reference CUT3R forward pass.

This is the checker for the HIP path (tests/, __graft_entry__.smoke(), bench.py cpu_baseline).  It is a
from-scratch functional restatement (no nn.Module tree) pinned against tests/golden/model_tiny_*.npz,
which were produced by the reference itself (tests/golden/make_fixtures.py).  The product never imports it.

Reference lines followed:
  encoder            src/dust3r/model.py:516-525, src/dust3r/patch_embed.py:18-32, src/croco/models/blocks.py:96-190
  state init         src/dust3r/model.py:538-568, 705-711
  dual decoder       src/dust3r/model.py:660-698, src/dust3r/blocks.py:87-133, 178-243, 246-297
  pose memory        src/dust3r/model.py:140-222
  rollout            src/dust3r/model.py:816-892
  DPT head           src/dust3r/heads/dpt_head.py:40-72, 213-260, src/croco/models/dpt_block.py:84-232, 281-513
  linear head        src/dust3r/heads/linear_head.py:246-346
  adaLN blocks       src/dust3r/blocks.py:356-420
  activations        src/dust3r/heads/postprocess.py:11-63, 113-167
  RoPE-2D            src/croco/models/curope/kernels.cu:17-82 (fp32 math, I/O in the token dtype)
"""
from __future__ import annotations

import math
from typing import Dict, List

import torch
import torch.nn.functional as F


# ------------------------------------------------------------------------------------------------ RoPE
def rope2d(tokens: torch.Tensor, pos: torch.Tensor, base: float = 100.0, f0: float = 1.0) -> torch.Tensor:
    """tokens [B,H,N,D], pos [B,N,2] (y,x; may be -1).  fp32 math; result cast back to tokens.dtype.

    kernels.cu:43-44,52-54: inv_freq[q] = F0 / powf(base, q/Q); freq = pos * inv_freq.
    """
    B, H, N, D = tokens.shape
    Q = D // 4
    q = torch.arange(Q, dtype=torch.float32)
    inv = (torch.tensor(f0, dtype=torch.float32) / torch.pow(torch.tensor(base, dtype=torch.float32), q / float(Q)))
    t = tokens.float()
    out = torch.empty_like(t)
    for X in range(2):
        ang = pos[:, :, X].to(torch.float32)[:, None, :, None] * inv[None, None, None, :]   # [B,1,N,Q]
        c, s = torch.cos(ang), torch.sin(ang)
        u = t[..., X * 2 * Q: X * 2 * Q + Q]
        v = t[..., X * 2 * Q + Q: X * 2 * Q + 2 * Q]
        out[..., X * 2 * Q: X * 2 * Q + Q] = u * c - v * s
        out[..., X * 2 * Q + Q: X * 2 * Q + 2 * Q] = v * c + u * s
    return out.to(tokens.dtype)


# ------------------------------------------------------------------------------------------------ operand precision
# The reference runs its matmuls/convolutions in TF32 on NVIDIA (src/croco/models/croco.py:13 `allow_tf32 = True`): operands
# rounded to a 10-bit mantissa, fp32 accumulation.  `matmul_precision("tf32")` emulates that on the CPU (operands of every
# Linear / Conv / attention product rounded to 10 mantissa bits, round-to-nearest-even, products and sums in fp32), and
# "fp16" rounds them to IEEE half (what the MI355X path feeds its MFMAs).  Default "fp32" = exact fp32, the pinned mode.
_PRECISION = ["fp32"]


class matmul_precision:
    def __init__(self, mode: str):
        assert mode in ("fp32", "tf32", "fp16"), mode
        self.mode = mode

    def __enter__(self):
        self.prev = _PRECISION[0]
        _PRECISION[0] = self.mode

    def __exit__(self, *exc):
        _PRECISION[0] = self.prev


def round_tf32(x: torch.Tensor) -> torch.Tensor:
    """fp32 -> nearest value with a 10-bit mantissa (ties to even), exponent range of fp32"""
    i = x.contiguous().view(torch.int32)
    lsb = (i >> 13) & 1
    i = (i + 0x0FFF + lsb) & ~0x1FFF
    return i.view(torch.float32)


def _r(x):
    m = _PRECISION[0]
    if m == "fp32" or x is None:
        return x
    if m == "tf32":
        return round_tf32(x.float())
    return x.to(torch.float16).float()


def _sdpa(q, k, v, scale):
    if _PRECISION[0] == "fp32":
        return F.scaled_dot_product_attention(q, k, v, scale=scale)
    s = torch.softmax((_r(q) @ _r(k).transpose(-1, -2)) * scale, dim=-1)
    return _r(s) @ _r(v)


# ------------------------------------------------------------------------------------------------ blocks
def _ln(x, sd, p, eps=1e-6):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], eps)


def _lin(x, sd, p):
    return F.linear(_r(x), _r(sd[p + ".weight"]), sd.get(p + ".bias"))


def _mlp(x, sd, p):
    return _lin(F.gelu(_lin(x, sd, p + ".fc1")), sd, p + ".fc2")


def _self_attn(x, pos, sd, p, heads, rope: bool, fp16_qk: bool):
    B, N, C = x.shape
    d = C // heads
    qkv = _lin(x, sd, p + ".qkv").reshape(B, N, 3, heads, d).transpose(1, 3)    # [B,H,3,N,d]
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    if rope:
        if fp16_qk:      # croco/models/blocks.py:125-131 (encoder): q,k rounded to fp16 around RoPE
            q = rope2d(q.to(torch.float16), pos).float()
            k = rope2d(k.to(torch.float16), pos).float()
        else:            # dust3r/blocks.py:114-121 (decoder): fp32 throughout
            q = rope2d(q, pos)
            k = rope2d(k, pos)
    o = _sdpa(q, k, v, d ** -0.5)
    return _lin(o.transpose(1, 2).reshape(B, N, C), sd, p + ".proj")


def _cross_attn(xq, y, qpos, kpos, sd, p, heads, rope: bool):
    B, Nq, C = xq.shape
    Nk = y.shape[1]
    d = C // heads
    q = _lin(xq, sd, p + ".projq").reshape(B, Nq, heads, d).permute(0, 2, 1, 3)
    k = _lin(y, sd, p + ".projk").reshape(B, Nk, heads, d).permute(0, 2, 1, 3)
    v = _lin(y, sd, p + ".projv").reshape(B, Nk, heads, d).permute(0, 2, 1, 3)
    if rope:
        if qpos is not None:
            q = rope2d(q, qpos)
        if kpos is not None:
            k = rope2d(k, kpos)
    o = _sdpa(q, k, v, d ** -0.5)
    return _lin(o.transpose(1, 2).reshape(B, Nq, C), sd, p + ".proj")


def enc_block(x, pos, sd, p, heads):
    x = x + _self_attn(_ln(x, sd, p + ".norm1"), pos, sd, p + ".attn", heads, True, True)
    return x + _mlp(_ln(x, sd, p + ".norm2"), sd, p + ".mlp")


def dec_block(x, y, xpos, ypos, sd, p, heads, rope: bool):
    x = x + _self_attn(_ln(x, sd, p + ".norm1"), xpos, sd, p + ".attn", heads, rope, False)
    y_ = _ln(y, sd, p + ".norm_y")
    x = x + _cross_attn(_ln(x, sd, p + ".norm2"), y_, xpos, ypos, sd, p + ".cross_attn", heads, rope)
    return x + _mlp(_ln(x, sd, p + ".norm3"), sd, p + ".mlp")


def _modln(x, mod, sd, p):
    sh, sc = _lin(F.silu(mod), sd, p + ".mlp.1").chunk(2, dim=-1)
    return _ln(x, sd, p + ".norm") * (1 + sc.unsqueeze(1)) + sh.unsqueeze(1)


def modln_block(x, mod, pos, sd, p, heads):
    x = x + _self_attn(_modln(x, mod, sd, p + ".norm1"), pos, sd, p + ".attn", heads, True, False)
    return x + _mlp(_modln(x, mod, sd, p + ".norm2"), sd, p + ".mlp")


# ------------------------------------------------------------------------------------------------ encoder
def patch_positions(B, h, w):
    y, x = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    return torch.stack([y.reshape(-1), x.reshape(-1)], -1)[None].expand(B, -1, -1).contiguous()


def encode_image(cfg, sd, img):
    """img [B,3,H,W] float (already normalised) -> (feat [B,N,E], pos [B,N,2] int64)."""
    B, _, H, W = img.shape
    P = cfg.patch_size
    x = F.conv2d(_r(img), _r(sd["patch_embed.proj.weight"]), sd["patch_embed.proj.bias"], stride=P)
    x = x.flatten(2).transpose(1, 2)
    pos = patch_positions(B, H // P, W // P)
    for i in range(cfg.enc_depth):
        x = enc_block(x, pos, sd, f"enc_blocks.{i}", cfg.enc_num_heads)
    return _ln(x, sd, "enc_norm"), pos


def normalize(img_u8):
    return (img_u8.float() / 255.0 - 0.5) / 0.5


# ------------------------------------------------------------------------------------------------ heads
def reg_dense_depth_exp(xyz, pos_z=False):
    if pos_z:
        xyz = xyz * torch.sign(xyz[..., -1:])
    d = xyz.norm(dim=-1, keepdim=True)
    return xyz / d.clip(min=1e-8) * torch.expm1(d)


def postprocess_pose(out):
    trans, quats = out[..., 0:3], out[..., 3:7]
    d = trans.norm(dim=-1, keepdim=True)
    trans = trans * (torch.expm1(d) / d.clip(min=1e-8))
    quats = F.normalize(quats, p=2, dim=-1)
    quats = torch.where(quats[..., 0:1] < 0, -quats, quats)
    return torch.cat([trans, quats], dim=-1)


def _conv(x, sd, p, **kw):
    return F.conv2d(_r(x), _r(sd[p + ".weight"]), sd.get(p + ".bias"), **kw)


def _rcu(x, sd, p):
    out = _conv(F.relu(x), sd, p + ".conv1", padding=1)
    out = _conv(F.relu(out), sd, p + ".conv2", padding=1)
    return out + x


def _fusion(sd, p, x0, x1=None):
    out = x0
    if x1 is not None:
        out = out + _rcu(x1, sd, p + ".resConfUnit1")
    out = _rcu(out, sd, p + ".resConfUnit2")
    out = F.interpolate(out, scale_factor=2, mode="bilinear", align_corners=True)
    return _conv(out, sd, p + ".out_conv")


def dpt_adapter(cfg, sd, p, toks: List[torch.Tensor], H, W):
    P = cfg.patch_size
    nh, nw = H // P, W // P
    L = [t.transpose(1, 2).reshape(t.shape[0], t.shape[2], nh, nw) for t in toks]
    a = p + ".act_postprocess"
    l0 = F.conv_transpose2d(_r(_conv(L[0], sd, a + ".0.0")), _r(sd[a + ".0.1.weight"]), sd[a + ".0.1.bias"], stride=4)
    l1 = F.conv_transpose2d(_r(_conv(L[1], sd, a + ".1.0")), _r(sd[a + ".1.1.weight"]), sd[a + ".1.1.bias"], stride=2)
    l2 = _conv(L[2], sd, a + ".2.0")
    l3 = _conv(_conv(L[3], sd, a + ".3.0"), sd, a + ".3.1", stride=2, padding=1)
    L = [l0, l1, l2, l3]
    L = [_conv(l, sd, f"{p}.scratch.layer_rn.{i}", padding=1) for i, l in enumerate(L)]
    p4 = _fusion(sd, p + ".scratch.refinenet4", L[3])[:, :, : L[2].shape[2], : L[2].shape[3]]
    p3 = _fusion(sd, p + ".scratch.refinenet3", p4, L[2])
    p2 = _fusion(sd, p + ".scratch.refinenet2", p3, L[1])
    p1 = _fusion(sd, p + ".scratch.refinenet1", p2, L[0])
    out = _conv(p1, sd, p + ".head.0", padding=1)
    out = F.interpolate(out, scale_factor=2, mode="bilinear", align_corners=True)
    out = F.relu(_conv(out, sd, p + ".head.2", padding=1))
    return _conv(out, sd, p + ".head.4")


def _pts_conf(fmap_bchw, pos_z=False):
    f = fmap_bchw.permute(0, 2, 3, 1)
    return reg_dense_depth_exp(f[..., 0:3], pos_z), 1 + f[..., 3].exp()


def _rgb(fmap_bchw, eps=1e-6):
    f = fmap_bchw.permute(0, 2, 3, 1)
    return (torch.sigmoid(f) * (1 - 2 * eps) + eps - 0.5) * 2


def head_forward(cfg, sd, head_in: List[torch.Tensor], H, W, pos, minimal=False) -> Dict[str, torch.Tensor]:
    h = "downstream_head"
    pose_token = head_in[-1][:, 0]
    token = head_in[-1][:, 1:]
    res = {}
    res["camera_pose"] = postprocess_pose(_mlp(pose_token, sd, h + ".pose_head.mlp"))
    if cfg.head_type == "dpt":
        x = head_in[:-1] + [token]
        res["pts3d_in_self_view"], res["conf_self"] = _pts_conf(dpt_adapter(cfg, sd, h + ".dpt_self", x, H, W))
        if minimal:
            return res
        tc = token
        for i in range(2):
            tc = modln_block(tc, pose_token, pos, sd, f"{h}.final_transform.{i}", cfg.dec_num_heads)
        if cfg.rgb_head:
            res["rgb"] = _rgb(dpt_adapter(cfg, sd, h + ".dpt_rgb", x, H, W))
        res["pts3d_in_other_view"], res["conf"] = _pts_conf(
            dpt_adapter(cfg, sd, h + ".dpt_cross", head_in[:-1] + [tc], H, W))
    else:
        P = cfg.patch_size
        B = token.shape[0]

        def shuffle(f):
            return F.pixel_shuffle(f.transpose(-1, -2).reshape(B, -1, H // P, W // P), P)
        res["pts3d_in_self_view"], res["conf_self"] = _pts_conf(shuffle(_mlp(token, sd, h + ".proj")), pos_z=True)
        if minimal:
            return res
        tc = token
        for i in range(2):
            tc = modln_block(tc, pose_token, pos, sd, f"{h}.final_transform.{i}", cfg.dec_num_heads)
        if cfg.rgb_head:
            res["rgb"] = _rgb(shuffle(_mlp(token, sd, h + ".rgb_proj")))
        res["pts3d_in_other_view"], res["conf"] = _pts_conf(shuffle(_mlp(tc, sd, h + ".cross_proj")))
    return res


# ------------------------------------------------------------------------------------------------ forward
def state_positions(cfg, B=1):
    w = cfg.state_width
    i = torch.arange(cfg.state_size)
    return torch.stack([i // w, i % w], -1)[None].expand(B, -1, -1).contiguous()


def decoder(cfg, sd, f_state, pos_state, feat, pos_img, f_pose):
    """model.py:660-698.  Returns list `dec` (len dec_depth+1) of image-side outputs and the new state."""
    f_img = torch.cat([f_pose, _lin(feat, sd, "decoder_embed")], dim=1)
    pos_img = torch.cat([-torch.ones(feat.shape[0], 1, 2, dtype=pos_img.dtype), pos_img], dim=1)
    outs = [(f_state, feat)]
    s, im = f_state, f_img
    for i in range(cfg.dec_depth):
        s_new = dec_block(s, im, pos_state, pos_img, sd, f"dec_blocks_state.{i}", cfg.state_dec_num_heads, True)
        im_new = dec_block(im, s, pos_img, pos_state, sd, f"dec_blocks.{i}", cfg.dec_num_heads, True)
        s, im = s_new, im_new
        outs.append((s, im))
    outs[-1] = (_ln(s, sd, "dec_norm_state"), _ln(im, sd, "dec_norm"))
    return [o[1] for o in outs], outs[-1][0]


def mem_inquire(cfg, sd, gfeat, mem):
    x = _lin(gfeat, sd, "pose_retriever.proj_q")
    x = torch.cat([x, sd["pose_retriever.masked_token"].expand(x.shape[0], -1, -1)], dim=-1)
    for i in range(2):
        x = dec_block(x, mem, None, None, sd, f"pose_retriever.read_blocks.{i}", cfg.dec_num_heads, False)
    return x[..., -cfg.dec_embed_dim:]


def mem_update(cfg, sd, mem, gfeat, pose_out):
    f = torch.cat([_lin(gfeat, sd, "pose_retriever.proj_q"), pose_out], dim=-1)
    for i in range(2):
        mem = dec_block(mem, f, None, None, sd, f"pose_retriever.write_blocks.{i}", cfg.dec_num_heads, False)
    return mem


@torch.no_grad()
def forward_views(cfg, sd, imgs: torch.Tensor, minimal: bool = False, return_states: bool = False):
    """imgs [V,3,H,W] normalised float.  Window inference with img_mask=True, ray_mask=False, update=True,
    reset=False for every view (the only mode the SLAM trackers use, track_frontend.py:51-72).
    The dummy zero ray-map encode (model.py:644-653) contributes exactly 0.0 for finite activations and is skipped.
    """
    V, _, H, W = imgs.shape
    feats, pos = encode_image(cfg, sd, imgs)                         # one batched encoder pass (model.py:611)
    state = _lin(sd["register_tokens.weight"][None], sd, "decoder_embed_state")
    spos = state_positions(cfg)
    mem = sd["pose_retriever.mem"]
    preds, states = [], [(state, mem)]
    for i in range(V):
        f_i, p_i = feats[i:i + 1], pos[i:i + 1]
        g = f_i.mean(dim=1, keepdim=True)
        pose_feat = sd["pose_token"] if i == 0 else mem_inquire(cfg, sd, g, mem)
        dec, new_state = decoder(cfg, sd, state, spos, f_i, p_i, pose_feat)
        new_mem = mem_update(cfg, sd, mem, g, dec[-1][:, 0:1])
        L = cfg.dec_depth
        head_in = [dec[0], dec[L * 2 // 4][:, 1:], dec[L * 3 // 4][:, 1:], dec[L]]
        preds.append(head_forward(cfg, sd, head_in, H, W, p_i, minimal))
        state, mem = new_state, new_mem
        states.append((state, mem))
    return (preds, states) if return_states else preds


# ------------------------------------------------------------------------------------------------ FLOPs
def flops_per_view(cfg, H, W, minimal=True) -> float:
    """Algorithmic FLOPs (2*MAC) of one view through the model -- same accounting as SURVEY.md section 8(d)."""
    P = cfg.patch_size
    N = (H // P) * (W // P)
    E, D, r = cfg.enc_embed_dim, cfg.dec_embed_dim, cfg.mlp_ratio
    S = cfg.state_size
    de = E // cfg.enc_num_heads
    f = 2 * N * (3 * P * P) * E
    f += cfg.enc_depth * (2 * N * E * (3 * E + E + 2 * r * E) + 4 * N * N * E)
    f += 2 * N * E * D
    Ni = N + 1
    for (nq, nk) in ((S, Ni), (Ni, S)):
        f += cfg.dec_depth * (2 * nq * D * (3 * D + D + D + D + 2 * r * D) + 2 * nk * D * 2 * D
                              + 4 * nq * nq * D + 4 * nq * nk * D)
    M2, Msz = 2 * D, cfg.local_mem_size
    f += 2 * (2 * Msz * M2 * (3 * M2 + M2 + M2 + M2 + 8 * M2) + 2 * 1 * M2 * 2 * M2 + 4 * Msz * Msz * M2 + 4 * Msz * M2)
    f += 2 * (2 * 1 * M2 * (3 * M2 + M2 + M2 + M2 + 8 * M2) + 2 * Msz * M2 * 2 * M2 + 4 * M2 + 4 * Msz * M2)
    if cfg.head_type == "dpt":
        ld, Fd, Ld = cfg.dpt_layer_dims, cfg.dpt_feature_dim, cfg.dpt_last_dim
        nh, nw = H // P, W // P
        toks = (E, D, D, D)

        def dpt(nch):
            g = 0
            g += 2 * nh * nw * (toks[0] * ld[0] + ld[0] * ld[0] * 16)
            g += 2 * nh * nw * (toks[1] * ld[1] + ld[1] * ld[1] * 4)
            g += 2 * nh * nw * toks[2] * ld[2]
            g += 2 * nh * nw * toks[3] * ld[3] + 2 * (nh // 2) * (nw // 2) * 9 * ld[3] * ld[3]
            sizes = [(4 * nh, 4 * nw), (2 * nh, 2 * nw), (nh, nw), (nh // 2, nw // 2)]
            for i, (a, b) in enumerate(sizes):
                g += 2 * a * b * 9 * ld[i] * Fd
            rcu = lambda a, b: 2 * 2 * a * b * 9 * Fd * Fd
            a, b = sizes[3]
            g += rcu(a, b) + 2 * (2 * a) * (2 * b) * Fd * Fd
            for i in (2, 1, 0):
                a, b = sizes[i]
                g += 2 * rcu(a, b) + 2 * (2 * a) * (2 * b) * Fd * Fd
            a, b = 8 * nh, 8 * nw
            g += 2 * a * b * 9 * Fd * (Fd // 2) + 2 * (2 * a) * (2 * b) * (9 * (Fd // 2) * Ld + Ld * nch)
            return g
        f += dpt(4)
        if not minimal:
            f += dpt(4) + dpt(3)
            f += 2 * (2 * N * D * (3 * D + D + 8 * D) + 4 * N * N * D)
    else:
        n_heads = 1 if minimal else 3
        f += n_heads * 2 * N * D * 4 * D * 2
    return float(f)
