"""MI355X runtime of the CUT3R pointmap network behind the reference's model interface.

Mirrors (same names / argument meaning / outputs) the surface the SLAM trackers use:
    ARCroco3DStereo.from_pretrained(path)          /root/reference/src/dust3r/model.py:305-318
    .normalize(img)                                 :1111-1114
    .encode_image(view) -> (feat, pos, shape)       :1102-1109
    .forward(views, ret_state) / __call__           :894-900, 816-892
and loads the reference state_dict schema (cut3r_slam_amd.config.state_dict_schema).

Execution model: weights are converted once to fp16 [N,K] panels resident in HBM; every operator is a hand-written
gfx950 kernel launched through the C ABI (cut3r_slam_amd.ops); the residual streams stay fp32, GEMM/attention
operands are fp16 with fp32 accumulation (the MI355X has no TF32: fp16-in/fp32-acc MFMA carries the same 10-bit
mantissa the reference's TF32 matmuls do).  The six encoder passes of a window run as ONE batched pass, the DPT
head runs once per window over all views (it does not feed the recurrence), and the dummy zero ray-map encode
(model.py:644-653, multiplied by 0.0) is skipped.
"""
from __future__ import annotations

import gc

import math
import re
from dataclasses import dataclass
from typing import Dict, List, Optional

import torch

from . import ops
from .config import Cut3rConfig, production_config, state_dict_schema

F16, F32 = torch.float16, torch.float32


@dataclass
class ARCroco3DStereoOutput:          # model.py:50-56 (ModelOutput with ress, views)
    ress: Optional[List[dict]] = None
    views: Optional[List[dict]] = None


class _Lin:
    """fp16 [N,K] weight panel + fp32 bias, N padded up to a multiple of 4 for the vector epilogue."""

    def __init__(self, w: torch.Tensor, b: Optional[torch.Tensor], dev):
        n = w.shape[0]
        self.n = n
        pad = (-n) % 4
        w2 = w.reshape(n, -1).to(F32)
        if pad:
            w2 = torch.cat([w2, torch.zeros(pad, w2.shape[1])], 0)
            if b is not None:
                b = torch.cat([b.to(F32), torch.zeros(pad)], 0)
        self.w = w2.to(device=dev, dtype=F16).contiguous()
        self.b = b.to(device=dev, dtype=F32).contiguous() if b is not None else None
        self.npad = n + pad
        self.k = self.w.shape[1]


class Cut3rModel:
    def __init__(self, cfg: Cut3rConfig, state_dict: Dict[str, torch.Tensor], device="cuda:0", minimal: bool = False):
        """minimal=True computes only what the SLAM trackers consume (pts3d_in_self_view, conf_self, camera_pose:
        hislam2/track_frontend.py:81-100) and skips dpt_cross / dpt_rgb / final_transform."""
        self.cfg = cfg
        self.device = torch.device(device)
        self.minimal = minimal
        schema = state_dict_schema(cfg)
        missing = [k for k in schema if k not in state_dict]
        if missing:
            raise KeyError(f"state_dict is missing {len(missing)} tensors, e.g. {missing[:4]}")
        for k, shp in schema.items():
            if tuple(state_dict[k].shape) != tuple(shp):
                raise ValueError(f"{k}: shape {tuple(state_dict[k].shape)} != schema {tuple(shp)}")
        self._buf: Dict[tuple, torch.Tensor] = {}
        self._graphs: Dict[tuple, tuple] = {}
        import os as _os
        self.use_graphs = _os.environ.get("CUT3R_GRAPHS", "1") != "0"
        # the state-side and image-side decoder blocks of a layer are independent (both read the previous layer's
        # pair, model.py:669-692): issue them on two streams so the captured graph has two parallel branches
        self.dual_stream = _os.environ.get("CUT3R_DUAL_STREAM", "1") != "0"
        self.dual_ln = _os.environ.get("CUT3R_DUAL_LN", "1") != "0"            # shared-statistics LayerNorm of the decoder inputs
        # the seven projections of a decoder layer as pair launches (state + image problem in one grid: cut3r_gemm_f16_pair).
        # OFF by default: measured -5 % end to end (97.4 -> 102.8 ms per 400-frame step, same box, interleaved runs): the pair
        # launches are 30-40 % faster than the two launches they replace, but they join the two blocks at every projection, and
        # the step loses the overlap of one block's MFMA-bound GEMMs with the other block's VALU-bound attention / LayerNorm
        self.pair_gemm = _os.environ.get("CUT3R_PAIR_GEMM", "0") != "0"
        # The one-window schedule (round 4) was measured with a layer as pair launches on the 64 x 64 pair kernels (`_dec_layer_pair`, with the
        # LayerNorm fold and the fused RoPE): 24.5 vs 23.6 ms per window -- slower as well.  A one-window launch costs ~6.5 us before its
        # first K-tile whatever its size (a + b K fit of the kernel trace), so fewer, fuller launches only help if the chain gets shorter,
        # and the pair form joins the two blocks seven times per layer.  `pair_rows` > 0 turns the pair form on below that many rows.
        self.pair_rows = int(_os.environ.get("CUT3R_PAIR_ROWS", "0"))
        # RoPE in the q/k projection epilogue: 0 off, 1 heads of 64 (default), 2 also heads of 48.  Bit-identical to the stand-alone
        # kernel.  Through the run-time epilogue of round 1 it lost 2.5 % end to end; as a compile-time epilogue of the 256x256 kernel
        # (a wave's 64-column slab is one head) it gains 1.0 % (5703 -> 5761 frames/s, interleaved runs); 48-wide heads straddle the
        # slabs and keep the stand-alone launch
        self.fused_rope = int(_os.environ.get("CUT3R_FUSED_ROPE", "1"))      # RoPE in the q/k projection epilogue (D = 64)
        self.rope48_rows = int(_os.environ.get("CUT3R_ROPE48_ROWS", "0"))     # 48-wide heads: fused below this many GEMM rows (0: never -- measured round 4 at one window: 28.4 vs 23.7 ms per window, the 128 x 192 tile puts 72 workgroups on 256 CUs)
        # DPT head of view i (all windows) on a third stream while the recurrent decoder works on view i+1: the decoder's
        # mid-size kernels leave matrix and memory pipes idle that the head's large convolutions can use
        self.head_overlap = _os.environ.get("CUT3R_HEAD_OVERLAP", "1") != "0"
        # windows per DPT-head pass of a view (the head's convolutions at the coarse pyramid levels have few output tiles)
        self.head_chunk = max(1, int(_os.environ.get("CUT3R_HEAD_CHUNK", "28")))      # measured at 28 windows: 8 -> 4806, 14 -> 4867, 28 -> 4932 frames/s
        # LayerNorm folded into the GEMMs (round 4): the fp32 + residual projections (attn.proj, cross_attn.proj, mlp.fc2) also write an fp16
        # copy of the residual stream and its per-row slab statistics; qkv / projq / projk|projv / fc1 read that copy through gamma-folded
        # weights and normalise in their epilogue (include/cut3r_hip.h, cut3r_gemm_desc).  The LayerNorm launches in front of them
        # disappear, except where the input has no producer GEMM (first encoder block, first decoder layer of a view, pose memory).
        # Economics, measured (tools/bench_lnfold.py, bench probe passes): the producers pay for the extra fp16 stream
        # (+7.5 us per tile round at N = 1024, HBM), the consumers for the row parameters (+1.2 us per tile).  Per DECODER block (28 windows)
        # that is 35 us against 58 us of LayerNorm launches -- a net win, and at one window every removed launch is latency (24.4 -> 23.6 ms
        # per window); per ENCODER block 200 us against 2 x 109 us -- nothing, while the encoder GEMMs (the dominant kernel) run 5-6 % slower.
        # So: 1 (default) = decoder only, 2 = encoder as well, 0 = off.
        self.ln_fold = int(_os.environ.get("CUT3R_LN_FOLD", "1"))
        self._head_stream = None
        self._side = None
        self._head_side = None
        # one-window schedule: key / value branches of a decoder layer on their own capture streams (CUT3R_KV_FORK; rows <= CUT3R_KV_FORK_ROWS).
        # OFF: measured round 4 (tools/bench_wb1.py, profiles/r04/kvfork.log) 28.2 ms per window against 23.5 ms without -- every edge between
        # two capture streams becomes a barrier packet pair in the hipGraph and 4 more of them per layer cost more than the overlap returns
        self.kv_fork = _os.environ.get("CUT3R_KV_FORK", "0") != "0"
        self.kv_fork_rows = int(_os.environ.get("CUT3R_KV_FORK_ROWS", "2048"))
        self._kv_streams = None
        self._prep(state_dict)

    # ------------------------------------------------------------------ reference-compatible constructors
    @classmethod
    def from_state_dict(cls, cfg, sd, device="cuda:0", **kw):
        return cls(cfg, sd, device, **kw)

    @classmethod
    def from_pretrained(cls, path: str, config: Optional[Cut3rConfig] = None, device="cuda:0", **kw):
        """Load a reference checkpoint (ckpt['model'] = state_dict; ckpt['args'].model = constructor string,
        model.py:72-92).  Only weights_only loading is used; the constructor string is parsed, never eval'd."""
        import argparse
        import os
        if not os.path.isfile(path):
            raise FileNotFoundError(f"{path}: no checkpoint (this runtime never fetches from a hub)")
        with torch.serialization.safe_globals([argparse.Namespace]):
            ckpt = torch.load(path, map_location="cpu", weights_only=True)
        sd = ckpt["model"] if "model" in ckpt else ckpt
        sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
        if not any(k.startswith("dec_blocks_state") for k in sd):       # model.py:390-393
            for k in list(sd):
                if k.startswith("dec_blocks"):
                    sd[k.replace("dec_blocks", "dec_blocks_state")] = sd[k]
        if config is None:
            config = production_config()
            args = getattr(ckpt.get("args", None), "model", "") if isinstance(ckpt, dict) else ""
            config = _config_from_ctor_string(args, config)
        return cls(config, sd, device, **kw)

    def to(self, device):
        return self

    def eval(self):
        return self

    # ------------------------------------------------------------------ weight preparation
    def _prep(self, sd):
        cfg, dev = self.cfg, self.device
        f32 = lambda k: sd[k].to(device=dev, dtype=F32).contiguous()
        self.w: Dict[str, object] = {}

        def lin(name, key=None):
            key = key or name
            self.w[name] = _Lin(sd[key + ".weight"], sd.get(key + ".bias"), dev)

        def ln(name):
            self.w[name] = (f32(name + ".weight"), f32(name + ".bias"))

        def fold(name, w, b, norm):
            """LayerNorm `norm` folded into the Linear (w, b) that consumes it: y = LN(x) W^T + b = rstd (x (gamma.W)^T - mu c) + d with the
            panel fp16(gamma . W), c_n = the row sums of THAT panel (so the constant part cancels exactly) and d = W beta + b"""
            if not self.ln_fold or w.shape[1] % 64 or w.shape[0] % 4:
                return
            gam, bet = sd[norm + ".weight"].double(), sd[norm + ".bias"].double()
            wd = w.double()
            L = _Lin((wd * gam[None, :]).float(), None, dev)
            L.b = (wd @ bet + b.double()).to(device=dev, dtype=F32).contiguous()
            L.c = L.w.double().sum(1).to(F32).contiguous()
            self.w[name + "@ln"] = L

        def enc_block(p):
            ln(p + ".norm1"); lin(p + ".attn.qkv"); lin(p + ".attn.proj"); ln(p + ".norm2")
            lin(p + ".mlp.fc1"); lin(p + ".mlp.fc2")
            if self.ln_fold >= 2:
                fold(p + ".attn.qkv", sd[p + ".attn.qkv.weight"], sd[p + ".attn.qkv.bias"], p + ".norm1")
                fold(p + ".mlp.fc1", sd[p + ".mlp.fc1.weight"], sd[p + ".mlp.fc1.bias"], p + ".norm2")

        def dec_block(p, folded=True):
            ln(p + ".norm1"); lin(p + ".attn.qkv"); lin(p + ".attn.proj")
            ln(p + ".norm2"); ln(p + ".norm3"); ln(p + ".norm_y")
            lin(p + ".cross_attn.projq"); lin(p + ".cross_attn.proj")
            wkv = torch.cat([sd[p + ".cross_attn.projk.weight"], sd[p + ".cross_attn.projv.weight"]], 0)
            bkv = torch.cat([sd[p + ".cross_attn.projk.bias"], sd[p + ".cross_attn.projv.bias"]], 0)
            self.w[p + ".cross_attn.projkv"] = _Lin(wkv, bkv, dev)
            lin(p + ".mlp.fc1"); lin(p + ".mlp.fc2")
            if folded:
                fold(p + ".attn.qkv", sd[p + ".attn.qkv.weight"], sd[p + ".attn.qkv.bias"], p + ".norm1")
                fold(p + ".cross_attn.projq", sd[p + ".cross_attn.projq.weight"], sd[p + ".cross_attn.projq.bias"], p + ".norm2")
                fold(p + ".cross_attn.projkv", wkv, bkv, p + ".norm_y")
                fold(p + ".mlp.fc1", sd[p + ".mlp.fc1.weight"], sd[p + ".mlp.fc1.bias"], p + ".norm3")

        self.w["patch_embed"] = _Lin(sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], dev)
        for i in range(cfg.enc_depth):
            enc_block(f"enc_blocks.{i}")
        ln("enc_norm")
        lin("decoder_embed"); lin("decoder_embed_state")
        for i in range(cfg.dec_depth):
            dec_block(f"dec_blocks.{i}"); dec_block(f"dec_blocks_state.{i}")
        ln("dec_norm"); ln("dec_norm_state")
        lin("pose_retriever.proj_q")
        for i in range(2):
            dec_block(f"pose_retriever.write_blocks.{i}", False); dec_block(f"pose_retriever.read_blocks.{i}", False)
        self.pose_token = f32("pose_token").reshape(1, -1)
        self.masked_token = f32("pose_retriever.masked_token").reshape(1, -1)
        self.mem0 = f32("pose_retriever.mem").reshape(cfg.local_mem_size, -1)
        self.register_tokens16 = sd["register_tokens.weight"].to(device=dev, dtype=F16).contiguous()
        w = cfg.state_width
        i = torch.arange(cfg.state_size)
        self.state_pos = torch.stack([i // w, i % w], -1)[None].to(dev).contiguous()       # int64 [1,S,2]
        for hd in {cfg.enc_embed_dim // cfg.enc_num_heads, cfg.dec_embed_dim // cfg.dec_num_heads, cfg.dec_embed_dim // cfg.state_dec_num_heads}:
            if hd % 16 == 0:
                ops.rope_table(dev, cfg.rope_freq, 1.0, hd)        # cos|sin tables of the RoPE launches: filled before any graph capture
        h = "downstream_head"
        lin(h + ".pose_head.mlp.fc1"); lin(h + ".pose_head.mlp.fc2")
        if cfg.head_type == "dpt":
            self._prep_dpt(sd, h + ".dpt_self", 4)
            if not self.minimal:
                self._prep_dpt(sd, h + ".dpt_cross", 4)
                if cfg.rgb_head:
                    self._prep_dpt(sd, h + ".dpt_rgb", 3)
        else:
            lin(h + ".proj.fc1"); lin(h + ".proj.fc2")
            if not self.minimal:
                lin(h + ".cross_proj.fc1"); lin(h + ".cross_proj.fc2")
                if cfg.rgb_head:
                    lin(h + ".rgb_proj.fc1"); lin(h + ".rgb_proj.fc2")
        if not self.minimal:
            for i in range(2):
                p = f"{h}.final_transform.{i}"
                for n in ("norm1", "norm2"):
                    ln(f"{p}.{n}.norm"); lin(f"{p}.{n}.mlp.1")
                lin(p + ".attn.qkv"); lin(p + ".attn.proj"); lin(p + ".mlp.fc1"); lin(p + ".mlp.fc2")

    def _prep_dpt(self, sd, p, nch):
        dev = self.device

        def c3(name):
            w = sd[name + ".weight"]                                  # [Cout,Cin,3,3] -> [Cout,(ky,kx,ci)]
            self.w[name] = _Lin(w.permute(0, 2, 3, 1).reshape(w.shape[0], -1), sd.get(name + ".bias"), dev)

        def c1(name):
            w = sd[name + ".weight"]
            self.w[name] = _Lin(w.reshape(w.shape[0], -1), sd.get(name + ".bias"), dev)

        def ct(name):
            w = sd[name + ".weight"]                                  # [Cin,Cout,k,k] -> [(i,j,co), ci]
            k = w.shape[2]
            l = _Lin(w.permute(2, 3, 1, 0).reshape(k * k * w.shape[1], w.shape[0]), None, dev)
            l.b = sd[name + ".bias"].to(device=dev, dtype=F32).contiguous()
            l.s = k
            self.w[name] = l

        a = p + ".act_postprocess"
        c1(a + ".0.0"); ct(a + ".0.1"); c1(a + ".1.0"); ct(a + ".1.1"); c1(a + ".2.0"); c1(a + ".3.0"); c3(a + ".3.1")
        for i in range(4):
            c3(f"{p}.scratch.layer_rn.{i}")
        for r in (1, 2, 3, 4):
            q = f"{p}.scratch.refinenet{r}"
            c1(q + ".out_conv")
            for u in ("resConfUnit1", "resConfUnit2"):
                c3(f"{q}.{u}.conv1"); c3(f"{q}.{u}.conv2")
        c3(p + ".head.0"); c3(p + ".head.2")
        self.w[p + ".head.4.w"] = sd[p + ".head.4.weight"].reshape(nch, -1).to(device=dev, dtype=F32).contiguous()
        self.w[p + ".head.4.b"] = sd[p + ".head.4.bias"].to(device=dev, dtype=F32).contiguous()

    # ------------------------------------------------------------------ buffers
    def buf(self, name, shape, dtype):
        key = (name, tuple(shape), dtype)
        t = self._buf.get(key)
        if t is None:
            t = torch.empty(shape, dtype=dtype, device=self.device)
            self._buf[key] = t
        return t

    # ------------------------------------------------------------------ primitives
    def _linear(self, x16, name, out, act=0, res1=None, res2=None, skinny=False, rope=None, ln=None, emit=None):
        """skinny: the operand has ONE row per independent sequence (pose token of a tracking window): weight-streaming
        kernel whose per-row result does not depend on how many windows are batched.
        rope = (positions [B,N,2], cols): RoPE of the first `cols` output columns fused into the GEMM epilogue.
        ln = slab statistics of the rows of `x16` (then x16 is the fp16 copy of the UN-normalised residual stream and the LayerNorm in front
        of this Linear runs inside its epilogue through the folded panel `name@ln`); emit = (stats, x16) this GEMM writes for the next one."""
        L = self.w[name + "@ln"] if ln is not None else self.w[name]
        if rope is not None:
            rope = (rope[0], rope[1], self.cfg.rope_freq, rope[2])
        return ops.linear(x16, L.w, out, L.b, act, res1, res2, tile=16 if skinny else 0, rope=rope,
                          ln=(ln, L.c, self.cfg.ln_eps) if ln is not None else None, emit=emit)

    def _folded(self, name):
        return self.ln_fold and (name + "@ln") in self.w

    def _xs(self, tag, x):
        """the fp16 copy + slab statistics that belong to the fp32 residual buffer `x` ([M,C], C % 64 == 0): (x16, stats [C/64, M, 2])"""
        M, Cc = x.shape
        return self.buf(tag + ".x16", (M, Cc), F16), self.buf(tag + ".xst", (Cc // 64, M, 2), F32)

    def _linear_pair(self, x0, name0, out0, x1, name1, out1, act=0, res0=None, res1=None, ex0=None, ex1=None):
        """the same projection of the state-side and the image-side decoder block in ONE launch (ops.linear_pair); ex = per-problem extras:
        rope = (pos, cols, D) fused RoPE, ln = slab statistics (the operand is then the un-normalised fp16 copy, folded panel `name@ln`),
        emit = (stats, fp16 copy) the projection writes"""
        def prob(x, name, out, res, ex):
            ex = dict(ex or {})
            L = self.w[name + "@ln"] if ex.get("ln") is not None else self.w[name]
            if ex.get("ln") is not None:
                ex["ln"] = (ex["ln"], L.c, self.cfg.ln_eps)
            if ex.get("rope") is not None:
                r = ex["rope"]
                ex["rope"] = (r[0], r[1], self.cfg.rope_freq, r[2])
            return (x, L.w, out, L.b, res, {k: v for k, v in ex.items() if v is not None})
        ops.linear_pair(prob(x0, name0, out0, res0, ex0), prob(x1, name1, out1, res1, ex1), act)

    def _fuse_rope(self, pos, D, rows):
        """the GEMM-fused RoPE covers head dimension 64 with one position row per GEMM row; 48-wide heads (the state side of the decoder)
        need the 128 x 192 tile, which loses to the 256 x 256 kernel + a RoPE launch on large batches (round 3: -3 % end to end at 28
        windows) but shortens the launch chain of the one-window schedule, where every launch is latency: fused up to `rope48_rows` rows"""
        dims = (64, 48) if (self.fused_rope == 2 or (self.fused_rope and rows <= self.rope48_rows)) else (64,)
        return self.fused_rope and pos is not None and D in dims and pos.is_contiguous() and pos.numel() == 2 * rows

    def _ln(self, x, name, out16=None, out32=None, mod=None):
        g, b = self.w[name]
        ops.layernorm(x, g, b, self.cfg.ln_eps, out16, out32, mod[0] if mod else None, mod[1] if mod else None)

    def _rope(self, t, pos):
        ops.rope_2d_pair(t, pos, None, None, self.cfg.rope_freq, 1.0)

    def _self_attn(self, tag, x_ln16, B, N, heads, pos, p, out, res, ln=None, emit=None):
        """x_ln16 fp16 [B*N,C] -> out(fp32) = res + proj(attn(qkv(x))).  ln: x_ln16 is the UN-normalised fp16 copy and `ln` its slab
        statistics (norm1 folded into qkv); emit: the projection also writes (stats, fp16 copy) of `out`."""
        Cc = x_ln16.shape[1]
        D = Cc // heads
        sk = N == 1
        qkv = self.buf(tag + ".qkv", (B * N, 3 * Cc), F16)
        fuse = (not sk) and self._fuse_rope(pos, D, B * N)
        self._linear(x_ln16, p + ".qkv", qkv, skinny=sk, rope=(pos, 2 * Cc, D) if fuse else None, ln=ln)
        v5 = qkv.view(B, N, 3, heads, D)
        q, k, v = v5[:, :, 0], v5[:, :, 1], v5[:, :, 2]
        if pos is not None and not fuse:
            ops.rope_2d_pair(q, pos, k, pos, self.cfg.rope_freq, 1.0)
        a = self.buf(tag + ".attn", (B, N, heads, D), F16)
        ops.attention(q, k, v, a, D ** -0.5)
        self._linear(a.view(B * N, Cc), p + ".proj", out, res1=res, skinny=sk, emit=emit)

    def _mlp(self, tag, x_ln16, p, out, res, skinny=False, ln=None, emit=None):
        M = x_ln16.shape[0]
        hdim = self.w[p + ".fc1"].npad
        h = self.buf(tag + ".mlp_h", (M, hdim), F16)
        self._linear(x_ln16, p + ".fc1", h, act=1, skinny=skinny, ln=ln)
        self._linear(h, p + ".fc2", out, res1=res, skinny=skinny, emit=emit)

    # ------------------------------------------------------------------ encoder
    def _encode(self, img: torch.Tensor):
        """img [B,3,H,W] (fp32 normalised, or uint8 -> normalisation fused).  Returns fp32 feat [B,N,E], fp16 copy."""
        cfg = self.cfg
        B, _, H, W = img.shape
        P, E = cfg.patch_size, cfg.enc_embed_dim
        nh, nw = H // P, W // P
        N = nh * nw
        M = B * N
        tag = f"enc{B}x{N}"
        patches = self.buf(tag + ".patches", (M, 3 * P * P), F16)
        ops.im2col_patch(img.contiguous(), P, patches)
        x = self.buf(tag + ".x", (M, E), F32)
        self._linear(patches, "patch_embed", x)
        y, xx = torch.meshgrid(torch.arange(nh, device=self.device), torch.arange(nw, device=self.device), indexing="ij")
        pos = torch.stack([y.reshape(-1), xx.reshape(-1)], -1)[None].expand(B, -1, -1).contiguous()
        ln16 = self.buf(tag + ".ln16", (M, E), F16)
        fold = self.ln_fold >= 2 and self._folded("enc_blocks.0.attn.qkv") and E % 64 == 0
        x16, xst = self._xs(tag, x) if fold else (None, None)
        em = (xst, x16) if fold else None
        for i in range(cfg.enc_depth):
            p = f"enc_blocks.{i}"
            if fold and i > 0:           # norm1 runs inside the qkv epilogue, on the copy + statistics the previous block's fc2 wrote
                self._self_attn(tag, x16, B, N, cfg.enc_num_heads, pos, p + ".attn", x, x, ln=xst, emit=em)
            else:
                self._ln(x, p + ".norm1", out16=ln16)
                self._self_attn(tag, ln16, B, N, cfg.enc_num_heads, pos, p + ".attn", x, x, emit=em)
            if fold:                     # norm2 inside the fc1 epilogue
                self._mlp(tag, x16, p + ".mlp", x, x, ln=xst, emit=em if i + 1 < cfg.enc_depth else None)
            else:
                self._ln(x, p + ".norm2", out16=ln16)
                self._mlp(tag, ln16, p + ".mlp", x, x)
        feat = torch.empty((B, N, E), dtype=F32, device=self.device)
        feat16 = torch.empty((B, N, E), dtype=F16, device=self.device)
        self._ln(x, "enc_norm", out16=feat16.view(M, E), out32=feat.view(M, E))
        return feat, feat16, pos

    # ------------------------------------------------------------------ hipGraph capture
    def _graphed(self, kind, fn, inp):
        """Capture `fn(static_input)` once per input signature into a hipGraph and replay it: a window is ~4000 kernel
        launches, which the Python/ctypes host path cannot issue as fast as the GPU retires them.  Inputs are copied
        into a static buffer; outputs are static tensors that the caller must consume before the next replay."""
        key = (kind, tuple(inp.shape), inp.dtype)      # `kind` may itself be a tuple (e.g. image size)
        ent = self._graphs.get(key)
        if ent is None:
            static_in = inp.clone()
            cur = torch.cuda.current_stream()
            side = torch.cuda.Stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                for _ in range(2):          # warm-up: creates every persistent workspace outside the capture
                    fn(static_in)
            cur.wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            # no garbage collection while the stream is capturing: a collected tensor / graph of a dead model would be
            # released (hipFree, hipGraphExecDestroy) in the middle of the capture
            gc_was_on = gc.isenabled()
            gc.collect()
            gc.disable()
            try:
                with torch.cuda.graph(graph):
                    out = fn(static_in)
            finally:
                if gc_was_on:
                    gc.enable()
            ent = (graph, static_in, out)
            self._graphs[key] = ent
        graph, static_in, out = ent
        static_in.copy_(inp, non_blocking=True)
        graph.replay()
        return out

    def normalize(self, img_tensor):
        return (img_tensor / 255.0 - 0.5) / 0.5

    def encode_image(self, view):
        img = view["img"]
        if not img.is_cuda:
            img = img.to(self.device)
        B = img.shape[0]
        im_shape = view.get("true_shape", torch.tensor(img.shape[-2:])[None].repeat(B, 1))
        img = img.to(F32) if img.dtype != torch.uint8 else img
        if self.use_graphs:
            feat, _, pos = self._graphed("enc", self._encode, img.contiguous())
            feat = feat.clone()            # callers keep encoder features (keyframe store)
        else:
            feat, _, pos = self._encode(img)
        return feat, pos, im_shape

    # ------------------------------------------------------------------ decoder block
    def _dec_block(self, tag, p, x, y, xpos, ypos, heads, out, B=1, pre_ln=False, xs=None, ys=None, os_=None, kv_stream=None, kv_only=False):
        """x fp32 [B*Nx,C], y fp32 [B*Ny,C] -> out fp32 [B*Nx,C]   (dust3r/blocks.py:292-297).  B = independent
        sequences (tracking windows batched through the decoder).  pre_ln: norm1(x) and norm_y(y) are already in this
        block's `.ln16` / `.y16` buffers (`_dual_norms`).
        LayerNorm fold: xs / ys = (fp16 copy, slab statistics) of x / y as the GEMMs that produced them wrote them (norm1 and norm_y then run
        inside the qkv / projk|projv epilogues); os_ = the pair that belongs to `out`: the three residual projections of this block write
        it (norm2 -> projq and norm3 -> fc1 read it here, the next layer's norm1 / norm_y read what fc2 leaves)."""
        Cc = x.shape[1]
        Nx, Ny = x.shape[0] // B, y.shape[0] // B
        D = Cc // heads
        ln16 = self.buf(tag + ".ln16", (B * Nx, Cc), F16)
        y16 = self.buf(tag + ".y16", (B * Ny, Cc), F16)
        kv = self.buf(tag + ".kv", (B * Ny, 2 * Cc), F16)
        kv4 = kv.view(B, Ny, 2, heads, D)
        k, v = kv4[:, :, 0], kv4[:, :, 1]
        if os_ is not None and not (Nx > 1 and self._folded(p + ".mlp.fc1")):
            os_ = None
        em = (os_[1], os_[0]) if os_ is not None else None          # (stats, fp16 copy) the residual projections write
        if xs is not None and not (Nx > 1 and self._folded(p + ".attn.qkv")):
            xs = None
        if ys is not None and not (Ny > 1 and self._folded(p + ".cross_attn.projkv")):
            ys = None

        def kv_branch():          # depends only on y (the other stream's previous layer): norm_y -> projk|projv -> RoPE(k)
            fuse_k = Ny > 1 and self._fuse_rope(ypos, D, B * Ny)
            if ys is not None:
                self._linear(ys[0], p + ".cross_attn.projkv", kv, rope=(ypos, Cc, D) if fuse_k else None, ln=ys[1])
            else:
                if not pre_ln:
                    self._ln(y, p + ".norm_y", out16=y16)
                self._linear(y16, p + ".cross_attn.projkv", kv, skinny=(Ny == 1), rope=(ypos, Cc, D) if fuse_k else None)
            if ypos is not None and not fuse_k:
                self._rope(k, ypos)

        if kv_only:               # the key / value branch alone (it depends on y only): run by the caller on its own capture stream
            kv_branch()
            return None
        if xs is not None:
            self._self_attn(tag, xs[0], B, Nx, heads, xpos, p + ".attn", out, x, ln=xs[1], emit=em)
        else:
            if not pre_ln:
                self._ln(x, p + ".norm1", out16=ln16)
            self._self_attn(tag, ln16, B, Nx, heads, xpos, p + ".attn", out, x, emit=em)
        q = self.buf(tag + ".q", (B, Nx, heads, D), F16)
        fuse_q = Nx > 1 and self._fuse_rope(xpos, D, B * Nx)
        if os_ is not None:
            self._linear(os_[0], p + ".cross_attn.projq", q.view(B * Nx, Cc), rope=(xpos, Cc, D) if fuse_q else None, ln=os_[1])
        else:
            self._ln(out, p + ".norm2", out16=ln16)
            self._linear(ln16, p + ".cross_attn.projq", q.view(B * Nx, Cc), skinny=(Nx == 1), rope=(xpos, Cc, D) if fuse_q else None)
        if xpos is not None and not fuse_q:
            self._rope(q, xpos)
        if kv_stream is not None:
            torch.cuda.current_stream().wait_stream(kv_stream)      # the branch ran beside the self-attention half (forked by the caller)
        else:
            kv_branch()      # (forking it from INSIDE this block's stream was tried: nested forks crash hipGraph capture_end on ROCm 7.2)
        a = self.buf(tag + ".cattn", (B, Nx, heads, D), F16)
        ops.attention(q, k, v, a, D ** -0.5)
        self._linear(a.view(B * Nx, Cc), p + ".cross_attn.proj", out, res1=out, skinny=(Nx == 1), emit=em)
        if os_ is not None:
            self._mlp(tag, os_[0], p + ".mlp", out, out, ln=os_[1], emit=em)
        else:
            self._ln(out, p + ".norm3", out16=ln16)
            self._mlp(tag, ln16, p + ".mlp", out, out, skinny=(Nx == 1))
        return out

    def _dec_layer_pair(self, l, a, s_a, b, s_b, pos_img, pos_state, Wn, fork=False, xs=None, os_=None):
        """Decoder layer l for BOTH streams: image block (a, s_a) -> b and state block (s_a, a) -> s_b (model.py:669-692, both
        read the previous layer's pair).  The seven projections of a block run as seven PAIR launches (state + image problem in
        one grid, ops.linear_pair); RoPE of the 48-wide state heads / attention stay per side (different token counts and head widths: 16 x 48
        state heads, 12 x 64 image heads) and, under graph capture with `fork`, on two streams.  Same kernels and row arithmetic as
        `_dec_block`: bit-identical results.
        This is the ONE-WINDOW schedule's form of a layer (round 4): at M = 768 / 769 rows every launch is latency, the two capture streams of
        `_dec_block` overlap poorly (kernel trace: one kernel running 59 % of the time, two 34 %), and one launch of 2 x 156 tiles of 64 x 64
        fills the 256 CUs where each problem alone leaves 100 idle.  xs = ((a16, a_stats), (s16, s_stats)) | None and os_ likewise for
        (b, s_b): the LayerNorm fold (norm1 / norm_y / norm2 / norm3 inside the projections' epilogues)."""
        cfg = self.cfg
        Cc = a.shape[1]
        ps, pi = f"dec_blocks_state.{l}", f"dec_blocks.{l}"
        Ni, Ns = a.shape[0] // Wn, s_a.shape[0] // Wn
        hs, hi = cfg.state_dec_num_heads, cfg.dec_num_heads
        Ds, Di = Cc // hs, Cc // hi
        S = dict(tag="decs", p=ps, x=s_a, y=a, xpos=pos_state, ypos=pos_img, heads=hs, D=Ds, out=s_b, Nx=Ns, Ny=Ni)
        I = dict(tag="deci", p=pi, x=a, y=s_a, xpos=pos_img, ypos=pos_state, heads=hi, D=Di, out=b, Nx=Ni, Ny=Ns)
        for sd in (S, I):
            t, Nx, Ny, h, D = sd["tag"], sd["Nx"], sd["Ny"], sd["heads"], sd["D"]
            sd["ln16"] = self.buf(t + ".ln16", (Wn * Nx, Cc), F16)
            sd["y16"] = self.buf(t + ".y16", (Wn * Ny, Cc), F16)
            sd["qkv"] = self.buf(t + ".qkv", (Wn * Nx, 3 * Cc), F16)
            sd["att"] = self.buf(t + ".attn", (Wn, Nx, h, D), F16)
            sd["q"] = self.buf(t + ".q", (Wn, Nx, h, D), F16)
            sd["kv"] = self.buf(t + ".kv", (Wn * Ny, 2 * Cc), F16)
            sd["catt"] = self.buf(t + ".cattn", (Wn, Nx, h, D), F16)
            sd["h"] = self.buf(t + ".mlp_h", (Wn * Nx, self.w[sd["p"] + ".mlp.fc1"].npad), F16)
        fold_in = xs is not None and self._folded(pi + ".attn.qkv") and self._folded(ps + ".attn.qkv")
        fold_out = os_ is not None and self._folded(pi + ".mlp.fc1") and self._folded(ps + ".mlp.fc1")
        if fold_in:
            (I["x16"], I["xst"]), (S["x16"], S["xst"]) = xs          # image tokens a, state tokens s_a
        if fold_out:
            (I["o16"], I["ost"]), (S["o16"], S["ost"]) = os_
        em = (lambda sd: {"emit": (sd["ost"], sd["o16"])}) if fold_out else (lambda sd: None)
        # fused RoPE where the head width is 64 and the tokens have a position row each
        small = Wn * max(Ni, Ns) <= self.pair_rows          # the 64 x 64 pair kernels carry the fused RoPE / the fold; large batches run plain pairs
        fr = lambda pos, D, rows: bool(small and self.fused_rope and pos is not None and D == 64 and pos.is_contiguous() and pos.numel() == 2 * rows)

        def both(fn):
            """fn(side) for the state and the image side; on two capture streams when forking"""
            if fork:
                cur = torch.cuda.current_stream()
                self._side.wait_stream(cur)
                with torch.cuda.stream(self._side):
                    fn(S)
                fn(I)
                cur.wait_stream(self._side)
            else:
                fn(S)
                fn(I)

        if not fold_in:
            if self.dual_ln and Cc in (768, 1024, 1536):
                self._dual_norms(l, a, s_a)
            else:
                both(lambda sd: (self._ln(sd["x"], sd["p"] + ".norm1", out16=sd["ln16"]), self._ln(sd["y"], sd["p"] + ".norm_y", out16=sd["y16"])))
        # ---- self attention
        for sd in (S, I):
            sd["fq"] = fr(sd["xpos"], sd["D"], Wn * sd["Nx"])
            sd["fk"] = fr(sd["ypos"], sd["D"], Wn * sd["Ny"])
        ex = lambda sd: {"rope": (sd["xpos"], 2 * Cc, sd["D"]) if sd["fq"] else None, "ln": sd["xst"] if fold_in else None}
        self._linear_pair(S["x16"] if fold_in else S["ln16"], ps + ".attn.qkv", S["qkv"], I["x16"] if fold_in else I["ln16"], pi + ".attn.qkv", I["qkv"],
                          ex0=ex(S), ex1=ex(I))

        def self_attn(sd):
            v5 = sd["qkv"].view(Wn, sd["Nx"], 3, sd["heads"], sd["D"])
            q, k, v = v5[:, :, 0], v5[:, :, 1], v5[:, :, 2]
            if sd["xpos"] is not None and not sd["fq"]:
                ops.rope_2d_pair(q, sd["xpos"], k, sd["xpos"], cfg.rope_freq, 1.0)
            ops.attention(q, k, v, sd["att"], sd["D"] ** -0.5)
        both(self_attn)
        self._linear_pair(S["att"].view(Wn * Ns, Cc), ps + ".attn.proj", S["out"], I["att"].view(Wn * Ni, Cc), pi + ".attn.proj", I["out"],
                          res0=S["x"], res1=I["x"], ex0=em(S), ex1=em(I))
        # ---- cross attention
        if not fold_out:
            both(lambda sd: self._ln(sd["out"], sd["p"] + ".norm2", out16=sd["ln16"]))
        ex = lambda sd: {"rope": (sd["xpos"], Cc, sd["D"]) if sd["fq"] else None, "ln": sd["ost"] if fold_out else None}
        self._linear_pair(S["o16"] if fold_out else S["ln16"], ps + ".cross_attn.projq", S["q"].view(Wn * Ns, Cc),
                          I["o16"] if fold_out else I["ln16"], pi + ".cross_attn.projq", I["q"].view(Wn * Ni, Cc), ex0=ex(S), ex1=ex(I))
        # projk|projv of the state block reads the image tokens (a), that of the image block the state tokens (s_a): different row counts,
        # same N and K -- one pair launch; the key half gets the fused RoPE where the heads are 64 wide
        exk = lambda sd, other: {"rope": (sd["ypos"], Cc, sd["D"]) if sd["fk"] else None, "ln": other["xst"] if fold_in else None}
        self._linear_pair(I["x16"] if fold_in else S["y16"], ps + ".cross_attn.projkv", S["kv"], S["x16"] if fold_in else I["y16"], pi + ".cross_attn.projkv", I["kv"],
                          ex0=exk(S, I), ex1=exk(I, S))

        def cross_attn(sd):
            kv4 = sd["kv"].view(Wn, sd["Ny"], 2, sd["heads"], sd["D"])
            k, v = kv4[:, :, 0], kv4[:, :, 1]
            if not sd["fq"] or not sd["fk"]:
                qq = sd["q"] if not sd["fq"] else None
                kk = k if not sd["fk"] else None
                if qq is not None and kk is not None:
                    ops.rope_2d_pair(qq, sd["xpos"], kk, sd["ypos"], cfg.rope_freq, 1.0)
                elif qq is not None:
                    self._rope(qq, sd["xpos"])
                elif kk is not None:
                    self._rope(kk, sd["ypos"])
            ops.attention(sd["q"], k, v, sd["catt"], sd["D"] ** -0.5)
        both(cross_attn)
        self._linear_pair(S["catt"].view(Wn * Ns, Cc), ps + ".cross_attn.proj", S["out"], I["catt"].view(Wn * Ni, Cc), pi + ".cross_attn.proj", I["out"],
                          res0=S["out"], res1=I["out"], ex0=em(S), ex1=em(I))
        # ---- MLP
        if not fold_out:
            both(lambda sd: self._ln(sd["out"], sd["p"] + ".norm3", out16=sd["ln16"]))
        ex = lambda sd: {"ln": sd["ost"] if fold_out else None}
        self._linear_pair(S["o16"] if fold_out else S["ln16"], ps + ".mlp.fc1", S["h"], I["o16"] if fold_out else I["ln16"], pi + ".mlp.fc1", I["h"], act=1,
                          ex0=ex(S), ex1=ex(I))
        self._linear_pair(S["h"], ps + ".mlp.fc2", S["out"], I["h"], pi + ".mlp.fc2", I["out"], res0=S["out"], res1=I["out"], ex0=em(S), ex1=em(I))

    def _dual_norms(self, l, a, s_a):
        """The four input norms of decoder layer l in two launches: the image tokens `a` feed norm1 of the image block and
        norm_y of the state block, the state tokens `s_a` feed norm1 of the state block and norm_y of the image block; each
        tensor is read once and its row statistics are shared (identical outputs to four separate LayerNorms)."""
        pi, ps = f"dec_blocks.{l}", f"dec_blocks_state.{l}"
        Ci = a.shape[1]
        ops.layernorm_dual(a, *self.w[pi + ".norm1"], self.buf("deci.ln16", (a.shape[0], Ci), F16),
                           *self.w[ps + ".norm_y"], self.buf("decs.y16", (a.shape[0], Ci), F16), self.cfg.ln_eps)
        ops.layernorm_dual(s_a, *self.w[ps + ".norm1"], self.buf("decs.ln16", (s_a.shape[0], Ci), F16),
                           *self.w[pi + ".norm_y"], self.buf("deci.y16", (s_a.shape[0], Ci), F16), self.cfg.ln_eps)

    # ------------------------------------------------------------------ pose memory
    def _mem_inquire(self, gfeat16, mem, B=1):
        """gfeat16 [B,E] fp16, mem [B*size, 2D] -> pose feature [B, D]  (model.py:217-222)"""
        cfg = self.cfg
        D = cfg.dec_embed_dim
        x = self.buf("memr.x", (B, 2 * D), F32)
        self._linear(gfeat16, "pose_retriever.proj_q", x[:, :D], skinny=True)
        x[:, D:] = self.masked_token
        a, b = x, self.buf("memr.x2", (B, 2 * D), F32)
        for i in range(2):
            self._dec_block("memr", f"pose_retriever.read_blocks.{i}", a, mem, None, None, cfg.dec_num_heads, b, B)
            a, b = b, a
        return a[:, D:]

    def _mem_update(self, mem, gfeat16, pose_out, out, B=1):
        """mem [B*size, 2D], pose_out [B, D] -> out (new memory)  (model.py:204-215)"""
        cfg = self.cfg
        D = cfg.dec_embed_dim
        f = self.buf("memw.f", (B, 2 * D), F32)
        self._linear(gfeat16, "pose_retriever.proj_q", f[:, :D], skinny=True)
        f[:, D:] = pose_out
        tmp = self.buf("memw.tmp", tuple(mem.shape), F32)
        self._dec_block("memw", "pose_retriever.write_blocks.0", mem, f, None, None, cfg.dec_num_heads, tmp, B)
        self._dec_block("memw", "pose_retriever.write_blocks.1", tmp, f, None, None, cfg.dec_num_heads, out, B)
        return out

    # ------------------------------------------------------------------ heads
    def _conv3(self, x, name, stride=1, relu_in=False, act=0, res1=None, res2=None, tag=None):
        L = self.w[name]
        B, H, W, _ = x.shape
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        out = self.buf(tag or name, (B, Ho, Wo, L.npad), F16)
        return ops.conv3x3_nhwc(x, L.w, out, L.b, stride, relu_in, act, res1, res2)

    def _conv1(self, x, name, tag=None):
        L = self.w[name]
        B, H, W, Cc = x.shape
        out = self.buf(tag or name, (B, H, W, L.npad), F16)
        ops.linear(x.view(-1, Cc), L.w, out.view(-1, L.npad), L.b)
        return out

    def _convT(self, x, name):
        L = self.w[name]
        B, H, W, Cc = x.shape
        cout = L.npad // (L.s * L.s)
        out = self.buf(name, (B, H * L.s, W * L.s, cout), F16)
        return ops.conv_transpose_nhwc(x, L.w, out, L.b, L.s)

    def _up2(self, x, tag):
        B, H, W, Cc = x.shape
        return ops.upsample2x(x, self.buf(tag, (B, 2 * H, 2 * W, Cc), F16))

    def _rcu(self, x, p, res2=None):
        t = self._conv3(x, p + ".conv1", relu_in=True, act=2)
        return self._conv3(t, p + ".conv2", res1=x, res2=res2)

    def _fusion(self, p, x0, x1=None):
        out = x0 if x1 is None else self._rcu(x1, p + ".resConfUnit1", res2=x0)
        out = self._rcu(out, p + ".resConfUnit2")
        out = self._up2(out, p + ".up")
        return self._conv1(out, p + ".out_conv")

    def _dpt(self, p, toks16: List[torch.Tensor], B, nh, nw):
        """toks16: 4 fp16 tensors [B*nh*nw, C_i] (NHWC token maps).  Returns fp16 [B*H*W, last_dim] features."""
        a = p + ".act_postprocess"
        t = [x.view(B, nh, nw, -1) for x in toks16]
        l0 = self._convT(self._conv1(t[0], a + ".0.0"), a + ".0.1")
        l1 = self._convT(self._conv1(t[1], a + ".1.0"), a + ".1.1")
        l2 = self._conv1(t[2], a + ".2.0")
        l3 = self._conv3(self._conv1(t[3], a + ".3.0"), a + ".3.1", stride=2)
        L = [self._conv3(l, f"{p}.scratch.layer_rn.{i}") for i, l in enumerate((l0, l1, l2, l3))]
        p4 = self._fusion(p + ".scratch.refinenet4", L[3])
        if p4.shape[1] != L[2].shape[1] or p4.shape[2] != L[2].shape[2]:
            p4 = p4[:, : L[2].shape[1], : L[2].shape[2]].contiguous()          # dpt_head.py:63-65 crop
        p3 = self._fusion(p + ".scratch.refinenet3", p4, L[2])
        p2 = self._fusion(p + ".scratch.refinenet2", p3, L[1])
        p1 = self._fusion(p + ".scratch.refinenet1", p2, L[0])
        o = self._conv3(p1, p + ".head.0")
        o = self._up2(o, p + ".head.up")
        o = self._conv3(o, p + ".head.2", act=2)
        return o

    def _dpt_pts(self, p, toks16, B, nh, nw, H, W, key_pts, key_conf, res, out=None):
        o = self._dpt(p, toks16, B, nh, nw)
        if out is None:
            pts = torch.empty((B, H, W, 3), dtype=F32, device=self.device)
            conf = torch.empty((B, H, W), dtype=F32, device=self.device)
        else:
            pts, conf = out
        ops.dpt_final(o.view(B * H * W, -1), self.w[p + ".head.4.w"], self.w[p + ".head.4.b"], 0, pts, conf)
        if res is not None:
            res[key_pts], res[key_conf] = pts, conf

    # ------------------------------------------------------------------ window forward
    @torch.no_grad()
    def forward_window(self, imgs: torch.Tensor, return_taps: bool = False):
        """imgs [V,3,H,W] on the GPU (fp32 normalised or uint8).  Returns (list of V pred dicts, taps).
        With graphs enabled the prediction tensors are static buffers, valid until the next call."""
        if self.use_graphs and not return_taps:
            return self._graphed("win", lambda x: self._forward_window(x, False), imgs.contiguous())
        return self._forward_window(imgs, return_taps)

    @torch.no_grad()
    def encode_batch(self, imgs: torch.Tensor) -> torch.Tensor:
        """encoder features fp32 [B,N,E] of B images (one batched pass; a fresh tensor the caller may keep)."""
        if self.use_graphs:
            feat, _, _ = self._graphed("enc", self._encode, imgs.contiguous())
            return feat.clone()
        return self._encode(imgs)[0]

    @torch.no_grad()
    def decode_window(self, feats: torch.Tensor, H: int, W: int):
        """Window inference from cached encoder features [V,N,E] (fp32): recurrent decoder + heads only.  The trackers
        keep every keyframe's features (keyframe.featI, hislam2/keyframe.py:36), so a keyframe is encoded ONCE instead
        of once in the filter and again in every window it belongs to ('double encoding')."""
        if self.use_graphs:
            return self._graphed(("dec", H, W), lambda f: self._forward_window(None, False, feats=f, hw=(H, W)), feats.contiguous())
        return self._forward_window(None, False, feats=feats, hw=(H, W))

    def _forward_window(self, imgs, return_taps: bool = False, feats=None, hw=None):
        """one window: images [V,3,H,W] (encoder + decoder + heads) or cached features [V,N,E] (decoder + heads)"""
        P = self.cfg.patch_size
        if feats is None:
            V, _, H, W = imgs.shape
            feat, feat16, _ = self._encode(imgs)
            res, taps = self._decode(feat[None], feat16[None], H, W, return_taps)
        else:
            H, W = hw
            res, taps = self._decode(feats[None], None, H, W, return_taps)
            V = feats.shape[0]
        preds = [{k: v[i:i + 1] for k, v in res.items()} for i in range(V)]
        return preds, taps

    @torch.no_grad()
    def decode_windows(self, feats: torch.Tensor, H: int, W: int):
        """Batched window inference: feats [Wn,V,N,E] fp32 (cached encoder features of Wn INDEPENDENT tracking windows:
        every window re-initialises state and pose memory, model.py:819-822) -> dict of stacked predictions
        (pts3d_in_self_view [Wn*V,H,W,3], conf_self [Wn*V,H,W], camera_pose [Wn*V,7], window-major).  The recurrent decoder
        is sequential over the V views but its GEMMs/attention are batched over the Wn windows (M = Wn*768 rows),
        which lifts them from the latency-bound tile-64 regime into the MFMA-bound tile-128 regime."""
        if self.use_graphs:
            return self._graphed(("decW", H, W), lambda f: self._decode(f, None, H, W, False)[0], feats.contiguous())
        return self._decode(feats, None, H, W, False)[0]

    def _decode(self, feat, feat16, H, W, return_taps=False):
        """feat fp32 [Wn,V,N,E] (+ optional fp16 copy).  Returns (dict of [Wn*V,...] tensors, taps)."""
        cfg = self.cfg
        P, E, D, Ld = cfg.patch_size, cfg.enc_embed_dim, cfg.dec_embed_dim, cfg.dec_depth
        Wn, V, N, _ = feat.shape
        nh, nw = H // P, W // P
        S = cfg.state_size
        dev = self.device
        if feat16 is None:
            feat16 = self.buf("win.feat16", (Wn, V, N, E), F16)
            ops.cast_f16(feat.reshape(Wn * V * N, E), feat16.view(Wn * V * N, E))
        y, xx = torch.meshgrid(torch.arange(nh, device=dev), torch.arange(nw, device=dev), indexing="ij")
        pos1 = torch.stack([y.reshape(-1), xx.reshape(-1)], -1)[None]                                  # [1,N,2]
        pos_img = torch.cat([-torch.ones(1, 1, 2, dtype=torch.int64, device=dev), pos1], dim=1).expand(Wn, -1, -1).contiguous()
        pos_state = self.state_pos.expand(Wn, -1, -1).contiguous()
        taps = {"enc_feat": feat[0]} if return_taps else None

        # state init (model.py:538-568, 705-711): identical for every window
        st = [self.buf("dec.state0", (Wn * S, D), F32), self.buf("dec.state1", (Wn * S, D), F32)]
        s0 = self.buf("dec.s0", (S, D), F32)
        self._linear(self.register_tokens16, "decoder_embed_state", s0)
        st[0].view(Wn, S, D).copy_(s0[None].expand(Wn, -1, -1))
        im = [self.buf("dec.img0", (Wn * (N + 1), D), F32), self.buf("dec.img1", (Wn * (N + 1), D), F32)]
        # LayerNorm fold: every residual buffer of the decoder has its fp16 copy + slab statistics, written by the GEMM that fills it
        fold = self.ln_fold and D % 64 == 0 and self._folded("dec_blocks.0.attn.qkv")
        xs_of = {}
        if fold:
            for t_, nm in ((st[0], "dec.state0"), (st[1], "dec.state1"), (im[0], "dec.img0"), (im[1], "dec.img1")):
                xs_of[t_.data_ptr()] = self._xs(nm, t_)
        msz = self.mem0.shape[0]
        mem = [self.buf("dec.mem0", (Wn * msz, 2 * D), F32), self.buf("dec.mem1", (Wn * msz, 2 * D), F32)]
        mem[0].view(Wn, msz, 2 * D).copy_(self.mem0[None].expand(Wn, -1, -1))
        h1, h2 = Ld * 2 // 4, Ld * 3 // 4
        tok1 = self.buf("head.tok1", (Wn, V, N, D), F16)
        tok2 = self.buf("head.tok2", (Wn, V, N, D), F16)
        tok3 = self.buf("head.tok3", (Wn, V, N, D), F16)
        tok3_32 = self.buf("head.tok3_32", (Wn, V, N, D), F32) if not self.minimal else None
        pose_tok = self.buf("head.pose_tok", (Wn, V, D), F32)
        pose_tok16 = self.buf("head.pose_tok16", (Wn, V, D), F16)
        g32 = self.buf("dec.g32", (Wn, E), F32)
        g16 = self.buf("dec.g16", (Wn, E), F16)
        dn32 = self.buf("dec.dn32", (Wn * (N + 1), D), F32)
        dn16 = self.buf("dec.dn16", (Wn * (N + 1), D), F16)
        Lde = self.w["decoder_embed"]
        w_de = Lde.w.unsqueeze(0).expand(Wn, -1, -1)
        b_de = Lde.b.unsqueeze(0).expand(Wn, -1)
        states = []
        cs, cm = 0, 0             # current state / mem buffer index
        fork = self.dual_stream and self.use_graphs and torch.cuda.is_current_stream_capturing()
        if fork and self._side is None:
            self._side = torch.cuda.Stream()
        head_fork = fork and self.head_overlap and cfg.head_type == "dpt"
        head_pts = head_conf = None
        if head_fork:
            if self._head_stream is None:
                self._head_stream = torch.cuda.Stream()
            head_pts = torch.empty((Wn * V, H, W, 3), dtype=F32, device=dev)
            head_conf = torch.empty((Wn * V, H, W), dtype=F32, device=dev)
        for i in range(V):
            ops.colmean_batched(feat[:, i], g32)                 # global feature of view i, every window, one launch
            ops.cast_f16(g32, g16)
            a, b = im[0], im[1]
            a3 = a.view(Wn, N + 1, D)
            if i == 0:
                a3[:, 0] = self.pose_token
            else:
                a3[:, 0] = self._mem_inquire(g16, mem[cm], Wn)
            ops.linear_batched(feat16[:, i], w_de, a3[:, 1:], b_de)                                   # decoder_embed
            s_a, s_b = st[cs], st[cs ^ 1]
            # after the window's last view the recurrent state and the pose memory are never read again (the next window
            # re-initialises both, model.py:819-822): their final updates are skipped unless the caller asked for them
            dead_tail = (i == V - 1) and not return_taps
            for l in range(Ld):
                # LayerNorm fold: from the second layer of a view on, the inputs of a layer were written by the previous layer's fc2 together
                # with their fp16 copies and slab statistics (the first layer's inputs come from decoder_embed / the state carry: LayerNorm launches)
                small = Wn * (N + 1) <= self.pair_rows
                pair = self.pair_gemm or small
                use_fold = fold and (small or not pair)        # (a forced pair form on a large batch runs the plain 128 / 256 pair kernels)
                xa, xsa = (xs_of[a.data_ptr()], xs_of[s_a.data_ptr()]) if (use_fold and l > 0) else (None, None)
                xb, xsb = (xs_of[b.data_ptr()], xs_of[s_b.data_ptr()]) if use_fold else (None, None)
                if dead_tail and l == Ld - 1:
                    self._dec_block("deci", f"dec_blocks.{l}", a, s_a, pos_img, pos_state, cfg.dec_num_heads, b, Wn, xs=xa, ys=xsa, os_=xb)
                    s_a, s_b = s_b, s_a
                    a, b = b, a
                    continue
                if pair:
                    self._dec_layer_pair(l, a, s_a, b, s_b, pos_img, pos_state, Wn, fork=fork, xs=(xa, xsa) if xa is not None else None,
                                         os_=(xb, xsb) if xb is not None else None)
                else:
                    pre = self.dual_ln and D in (768, 1024, 1536) and xa is None
                    if pre:
                        self._dual_norms(l, a, s_a)
                    if fork:
                        cur = torch.cuda.current_stream()
                        kvf = self.kv_fork and Wn * (N + 1) <= self.kv_fork_rows
                        kS = kI = None
                        if kvf:
                            # one-window schedule: the key / value projections of both blocks (LayerNorm_y -> projk|projv -> RoPE: they read the
                            # layer's INPUTS only) leave the two block chains and run on two more capture streams, forked here -- at the layer
                            # level, siblings of the block streams -- and joined in front of each block's cross attention
                            if self._kv_streams is None:
                                self._kv_streams = (torch.cuda.Stream(), torch.cuda.Stream())
                            kS, kI = self._kv_streams
                            kS.wait_stream(cur)
                            kI.wait_stream(cur)
                            with torch.cuda.stream(kS):
                                self._dec_block("decs", f"dec_blocks_state.{l}", s_a, a, pos_state, pos_img, cfg.state_dec_num_heads, s_b, Wn, pre_ln=pre,
                                                xs=xsa, ys=xa, os_=xsb, kv_only=True)
                            with torch.cuda.stream(kI):
                                self._dec_block("deci", f"dec_blocks.{l}", a, s_a, pos_img, pos_state, cfg.dec_num_heads, b, Wn, pre_ln=pre, xs=xa, ys=xsa,
                                                os_=xb, kv_only=True)
                        self._side.wait_stream(cur)
                        with torch.cuda.stream(self._side):
                            self._dec_block("decs", f"dec_blocks_state.{l}", s_a, a, pos_state, pos_img, cfg.state_dec_num_heads, s_b, Wn, pre_ln=pre,
                                            xs=xsa, ys=xa, os_=xsb, kv_stream=kS)
                        self._dec_block("deci", f"dec_blocks.{l}", a, s_a, pos_img, pos_state, cfg.dec_num_heads, b, Wn, pre_ln=pre, xs=xa, ys=xsa, os_=xb,
                                        kv_stream=kI)
                        cur.wait_stream(self._side)
                    else:
                        self._dec_block("decs", f"dec_blocks_state.{l}", s_a, a, pos_state, pos_img, cfg.state_dec_num_heads, s_b, Wn, pre_ln=pre,
                                        xs=xsa, ys=xa, os_=xsb)
                        self._dec_block("deci", f"dec_blocks.{l}", a, s_a, pos_img, pos_state, cfg.dec_num_heads, b, Wn, pre_ln=pre, xs=xa, ys=xsa, os_=xb)
                s_a, s_b = s_b, s_a
                a, b = b, a
                if l + 1 == h1 or l + 1 == h2:
                    tk = tok1 if l + 1 == h1 else tok2
                    tk[:, i].copy_(a.view(Wn, N + 1, D)[:, 1:])      # fp32 -> fp16 (round to nearest even), one strided copy
            # final norms (model.py:694-697): new state = dec_norm_state(state), img = dec_norm(img)
            if not dead_tail:
                self._ln(s_a, "dec_norm_state", out32=s_b)
            new_state = s_b
            self._ln(a, "dec_norm", out16=dn16, out32=dn32)
            dn16v, dn32v = dn16.view(Wn, N + 1, D), dn32.view(Wn, N + 1, D)
            tok3[:, i].copy_(dn16v[:, 1:])
            if tok3_32 is not None:
                tok3_32[:, i].copy_(dn32v[:, 1:])
            pose_tok[:, i].copy_(dn32v[:, 0])
            pose_tok16[:, i].copy_(dn16v[:, 0])
            if not dead_tail:
                self._mem_update(mem[cm], g16, dn32v[:, 0], mem[cm ^ 1], Wn)
                cm ^= 1
            if head_fork:
                # fork: this view's DPT head (batch = the Wn windows) runs beside the decoder of the next view
                cur = torch.cuda.current_stream()
                self._head_stream.wait_stream(cur)
                with torch.cuda.stream(self._head_stream):
                    for c0 in range(0, Wn, self.head_chunk):
                        c1 = min(Wn, c0 + self.head_chunk)
                        nb = c1 - c0
                        tk = []
                        for name, src, dim in (("f", feat16, E), ("t1", tok1, D), ("t2", tok2, D), ("t3", tok3, D)):
                            t = self.buf("head.view." + name, (nb, N, dim), F16)
                            t.copy_(src[c0:c1, i])
                            tk.append(t.view(nb * N, dim))
                        pv = self.buf("head.view.pts", (nb, H, W, 3), F32)
                        cv = self.buf("head.view.conf", (nb, H, W), F32)
                        self._dpt_pts("downstream_head.dpt_self", tk, nb, nh, nw, H, W, None, None, None, out=(pv, cv))
                        head_pts.view(Wn, V, H, W, 3)[c0:c1, i].copy_(pv)
                        head_conf.view(Wn, V, H, W)[c0:c1, i].copy_(cv)
            # the state ping-pong: make st[cs] hold the new state for the next view
            cs = 0 if new_state is st[0] else 1
            if return_taps:
                states.append((new_state.view(Wn, S, D)[0].clone(), mem[cm].view(Wn, msz, 2 * D)[0].clone()))
        if return_taps:
            taps["states"] = states

        # ---- heads, batched over all Wn*V views
        BV = Wn * V
        h = "downstream_head"
        ph = self.buf("head.pose_h", (BV, self.w[h + ".pose_head.mlp.fc1"].npad), F16)
        self._linear(pose_tok16.view(BV, D), h + ".pose_head.mlp.fc1", ph, act=1)
        praw = self.buf("head.pose_raw", (BV, 8), F32)
        self._linear(ph, h + ".pose_head.mlp.fc2", praw)
        pose = torch.empty((BV, 7), dtype=F32, device=dev)
        ops.postprocess_pose(praw[:, :7].contiguous(), pose)
        res: Dict[str, torch.Tensor] = {"camera_pose": pose}
        toks = [feat16.reshape(BV * N, E), tok1.view(BV * N, D), tok2.view(BV * N, D), tok3.view(BV * N, D)]
        if head_fork:
            torch.cuda.current_stream().wait_stream(self._head_stream)          # join the head branch
            res["pts3d_in_self_view"], res["conf_self"] = head_pts, head_conf
        elif cfg.head_type == "dpt":
            # chunks of <= 8 views keep the implicit-GEMM grids (M/128 row tiles on grid.y) inside the 65535 limit
            pts = torch.empty((BV, H, W, 3), dtype=F32, device=dev)
            conf = torch.empty((BV, H, W), dtype=F32, device=dev)
            for c0 in range(0, BV, 8):
                c1 = min(BV, c0 + 8)
                tk = [t.view(BV, N, -1)[c0:c1].reshape((c1 - c0) * N, -1) for t in toks]
                self._dpt_pts(h + ".dpt_self", tk, c1 - c0, nh, nw, H, W, None, None, None, out=(pts[c0:c1], conf[c0:c1]))
            res["pts3d_in_self_view"], res["conf_self"] = pts, conf
        else:
            self._linear_head(h + ".proj", tok3.view(BV * N, D), BV, nh, nw, True, "pts3d_in_self_view", "conf_self", res)
        if not self.minimal:
            posBV = pos1.expand(BV, -1, -1).contiguous()
            self._cross_heads(toks, tok3_32.view(BV, N, D), pose_tok.view(BV, D), posBV, BV, nh, nw, H, W, res)
        return res, taps

    def _linear_head(self, p, tok16, V, nh, nw, pos_z, key_pts, key_conf, res, rgb=False):
        P = self.cfg.patch_size
        hdim = self.w[p + ".fc1"].npad
        hb = self.buf(p + ".h", (tok16.shape[0], hdim), F16)
        self._linear(tok16, p + ".fc1", hb, act=1)
        nout = self.w[p + ".fc2"].npad
        raw = self.buf(p + ".raw", (tok16.shape[0], nout), F32)
        self._linear(hb, p + ".fc2", raw)
        nch = nout // (P * P)
        # pixel shuffle: token (py,px), channel (c,iy,ix) -> pixel (py*P+iy, px*P+ix), channel c   (linear_head.py:310-313)
        fmap = raw.view(V, nh, nw, nch, P, P).permute(0, 1, 4, 2, 5, 3).reshape(V * nh * P * nw * P, nch).contiguous()
        H, W = nh * P, nw * P
        pts = torch.empty((V, H, W, 3), dtype=F32, device=self.device)
        if rgb:
            res[key_pts] = ((torch.sigmoid(fmap) * (1 - 2e-6) + 1e-6 - 0.5) * 2).view(V, H, W, 3)
            return
        conf = torch.empty((V, H, W), dtype=F32, device=self.device)
        ops.postprocess_pts(fmap, pos_z, pts, conf)
        res[key_pts], res[key_conf] = pts, conf

    def _cross_heads(self, toks, tok3_32, pose_tok, pos, V, nh, nw, H, W, res):
        """pts3d_in_other_view / conf / rgb (dpt_head.py:219-259, linear_head.py:299-346); not consumed by SLAM."""
        cfg = self.cfg
        h = "downstream_head"
        D, N = cfg.dec_embed_dim, nh * nw
        heads = cfg.dec_num_heads
        x = self.buf("ft.x", (V * N, D), F32)
        x.copy_(tok3_32.view(V * N, D))
        ln16 = self.buf("ft.ln16", (V * N, D), F16)
        mod = self.buf("ft.mod", (V, 2 * D), F32)
        for i in range(2):
            p = f"{h}.final_transform.{i}"
            for n, fn in (("norm1", "attn"), ("norm2", "mlp")):
                L = self.w[f"{p}.{n}.mlp.1"]
                ops.gemv(pose_tok, L.w, mod, L.b, silu_in=True)
                for v in range(V):            # modulation vectors are per view (batch element)
                    sh, sc = mod[v, :D].contiguous(), mod[v, D:].contiguous()
                    g, b = self.w[f"{p}.{n}.norm"]
                    ops.layernorm(x[v * N:(v + 1) * N], g, b, cfg.ln_eps, ln16[v * N:(v + 1) * N], None, sc, sh)
                if fn == "attn":
                    self._self_attn("ft", ln16, V, N, heads, pos, p + ".attn", x, x)
                else:
                    self._mlp("ft", ln16, p + ".mlp", x, x)
        tc16 = self.buf("ft.tc16", (V * N, D), F16)
        ops.cast_f16(x, tc16)
        if cfg.head_type == "dpt":
            if cfg.rgb_head:
                p = h + ".dpt_rgb"
                o = self._dpt(p, toks, V, nh, nw)
                rgb = torch.empty((V, H, W, 3), dtype=F32, device=self.device)
                ops.dpt_final(o.view(V * H * W, -1), self.w[p + ".head.4.w"], self.w[p + ".head.4.b"], 1, rgb, None)
                res["rgb"] = rgb
            self._dpt_pts(h + ".dpt_cross", toks[:3] + [tc16], V, nh, nw, H, W, "pts3d_in_other_view", "conf", res)
        else:
            if cfg.rgb_head:
                self._linear_head(h + ".rgb_proj", toks[3], V, nh, nw, False, "rgb", None, res, rgb=True)
            self._linear_head(h + ".cross_proj", tc16, V, nh, nw, False, "pts3d_in_other_view", "conf", res)

    # ------------------------------------------------------------------ reference-shaped entry points
    @torch.no_grad()
    def forward(self, views, ret_state=False):
        """views: list of view dicts (hislam2/track_frontend.py:51-72); only the SLAM mode is supported
        (img_mask=True, ray_mask=False, update=True, reset=False for every view, batch 1)."""
        for v in views:
            if v["img"].shape[0] != 1:
                raise NotImplementedError("batch size per view must be 1 (as in the SLAM trackers)")
            if not bool(v["img_mask"].all()) or bool(v["ray_mask"].any()) or bool(v["reset"].any()) or \
                    (v.get("update") is not None and not bool(v["update"].all())):
                raise NotImplementedError("only img_mask=True, ray_mask=False, update=True, reset=False is supported")
        imgs = torch.cat([v["img"] for v in views], 0).to(self.device, F32)
        preds, _ = self.forward_window(imgs)
        out = ARCroco3DStereoOutput(ress=preds, views=views)
        return (out, None) if ret_state else out

    __call__ = forward


# name aliases so `from cut3r_slam_amd.model import ARCroco3DStereo` reads like the reference import
ARCroco3DStereo = Cut3rModel


def _config_from_ctor_string(s: str, base: Cut3rConfig) -> Cut3rConfig:
    """Parse 'ARCroco3DStereo(ARCroco3DStereoConfig(state_size=768, ..., enc_embed_dim=1024, ...))' (the string the reference
    eval()s, model.py:72-92) without eval.  Output activations other than the ones this runtime implements --
    depth_mode ('exp', -inf, inf), conf_mode ('exp', 1, inf), pose_mode ('exp', -inf, inf) (heads/postprocess.py:11-63) -- and a
    checkpoint without pose head are refused instead of being silently mis-evaluated."""
    if not s:
        return base
    d = base.to_dict()
    for key in ("state_size", "local_mem_size", "enc_embed_dim", "enc_depth", "enc_num_heads", "dec_embed_dim",
                "dec_depth", "dec_num_heads", "state_dec_num_heads", "ray_enc_depth", "patch_size", "mlp_ratio"):
        m = re.search(rf"\b{key}\s*=\s*(\d+)", s)
        if m:
            d[key] = int(m.group(1))
    m = re.search(r"head_type\s*=\s*['\"](\w+)['\"]", s)
    if m:
        d["head_type"] = m.group(1)
    m = re.search(r"img_size\s*=\s*[\(\[]\s*(\d+)\s*,\s*(\d+)\s*[\)\]]", s)
    if m:
        d["img_size"] = (int(m.group(1)), int(m.group(2)))
    for key in ("rgb_head", "pose_head"):
        m = re.search(rf"\b{key}\s*=\s*(True|False)", s)
        if m:
            d[key] = m.group(1) == "True"
    m = re.search(r"pos_embed\s*=\s*['\"]RoPE(\d+(?:\.\d+)?)['\"]", s)
    if m:
        d["rope_freq"] = float(m.group(1))
    elif re.search(r"pos_embed\s*=", s):
        raise NotImplementedError("only RoPE position embeddings (pos_embed='RoPE<freq>') are implemented")
    want = {"depth_mode": ("exp", "-inf", "inf"), "conf_mode": ("exp", "1", "inf"), "pose_mode": ("exp", "-inf", "inf")}
    for key, exp in want.items():
        m = re.search(rf"\b{key}\s*=\s*[\(\[]\s*['\"](\w+)['\"]\s*,\s*([^,]+?)\s*,\s*([^\)\]]+?)\s*[\)\]]", s)
        if m:
            got = (m.group(1), m.group(2).replace("float('inf')", "inf").replace('float("inf")', "inf").replace("1.0", "1").replace(" ", ""),
                   m.group(3).replace("float('inf')", "inf").replace('float("inf")', "inf").replace(" ", ""))
            if got != exp:
                raise NotImplementedError(f"{key}={got}: this runtime implements {exp} only")
    if not d.get("pose_head", True):
        raise NotImplementedError("checkpoints without pose head are not supported (the SLAM trackers need camera_pose)")
    return Cut3rConfig.from_dict(d)
