"""GPU parity of the full CUT3R runtime (HIP kernels through the C ABI) against the reference golden fixtures and
against the CPU oracle.

Tolerances (stated): GEMM/attention operands are fp16 with fp32 accumulation (the
reference runs TF32: the same 10-bit mantissa).  Since round 4 the golden test applies the TF32-budget rule of
tests/test_precision_gpu.py against the reference's own outputs -- e_hip <= 2 e_tf32 + 2e-4 per output -- plus absolute caps of
3 x the measured errors (CAP below); the u8 / minimal / oracle tests keep absolute bounds of the same size.
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from cut3r_slam_amd.config import Cut3rConfig  # noqa: E402
from cut3r_slam_amd.model import Cut3rModel  # noqa: E402
from cut3r_slam_amd.weights import synth_state_dict  # noqa: E402
from oracle import cut3r_oracle as O  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda:0"


def _relerr(got, ref):
    got, ref = torch.as_tensor(got).double().cpu(), torch.as_tensor(ref).double().cpu()
    return float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-12))


def _check(name, got, ref, tol, log, mask=None):
    if mask is not None:
        got = torch.as_tensor(got).cpu()[mask]
        ref = torch.as_tensor(ref).cpu()[mask]
    e = _relerr(got, ref)
    log.append(f"{name}: {e:.2e} (tol {tol:.0e})")
    assert e <= tol, "\n".join(log)


# absolute caps of the golden test = 3 x the errors measured on MI355X (round 4, `pytest -s`: profiles/r04/achieved_errors.txt), per output
# class, next to the TF32-budget rule below.  (Rounds 1-3 allowed 5e-3 / 1e-2 / 2e-2: a regression that tripled the error passed.)
CAP = {"model_tiny_dpt": {"enc_feat": 1.5e-3, "state": 2.5e-3, "mem": 2.5e-3, "camera_pose": 4e-3, "pts": 6e-3, "conf": 4e-3, "rgb": 4e-3},
       "model_tiny_linear": {"enc_feat": 1.5e-3, "state": 2.5e-3, "mem": 2.5e-3, "camera_pose": 4e-3, "pts": 6e-3, "conf": 4e-3, "rgb": 4e-3},
       "model_medium_dpt": {"enc_feat": 2e-3, "state": 2.5e-3, "mem": 2.5e-3, "camera_pose": 6e-3, "pts": 8e-3, "conf": 4e-3, "rgb": 4e-3}}


def _cls(k):
    return "camera_pose" if k == "camera_pose" else "rgb" if k == "rgb" else "conf" if k.startswith("conf") else "pts"


@pytest.mark.parametrize("name", ["model_tiny_dpt", "model_tiny_linear", "model_medium_dpt"])
def test_model_matches_reference_golden(name):
    """every output of the network against the REFERENCE's own outputs (tests/golden/*.npz), under the TF32-budget rule of
    tests/test_precision_gpu.py: with e_x = max|x - reference| / max|reference|,  e_hip <= 2 e_tf32 + 2e-4  where e_tf32 is the error of the CPU
    restatement run with TF32-rounded operands (the reference's own arithmetic on its GPUs: src/croco/models/croco.py:13) against the same
    reference outputs -- plus the absolute caps above."""
    f = np.load(os.path.join(GOLD, name + ".npz"))
    cfg = Cut3rConfig.from_dict(json.loads(bytes(f["config_json"]).decode()))
    sd = synth_state_dict(cfg, int(f["seed"]))
    model = Cut3rModel(cfg, sd, DEV, minimal=False)
    imgs = torch.from_numpy(f["imgs"])
    x = O.normalize(imgs)
    with O.matmul_precision("tf32"):
        tf_feat, _ = O.encode_image(cfg, sd, x[:1])
        tf_preds, tf_states = O.forward_views(cfg, sd, x, minimal=False, return_states=True)
    log, cap = [], CAP[name]

    def check(tag, got, tf, ref, c, mask=None):
        got, tf, ref = (torch.as_tensor(t).float().cpu() for t in (got, tf, ref))
        if mask is not None:
            got, tf, ref = got[mask], tf[mask], ref[mask]
        e_hip, e_tf = _relerr(got, ref), _relerr(tf, ref)
        log.append(f"{tag}: hip {e_hip:.2e} tf32 {e_tf:.2e} (cap {cap[c]:.1e})")
        assert e_hip <= cap[c] and e_hip <= 2.0 * e_tf + 2e-4, "\n".join(log)

    feat, pos, _ = model.encode_image({"img": model.normalize(imgs[:1].float()).to(DEV)})
    torch.cuda.synchronize()
    assert torch.equal(pos.cpu(), torch.from_numpy(f["enc_pos0"]))
    check("enc_feat", feat, tf_feat, f["enc_feat0"], "enc_feat")
    preds, taps = model.forward_window(model.normalize(imgs.float()).to(DEV), return_taps=True)
    torch.cuda.synchronize()
    for i, (st, m) in enumerate(taps["states"]):
        check(f"state{i+1}", st[None], tf_states[i + 1][0], f[f"state{i+1}_feat"], "state")
        check(f"mem{i+1}", m[None], tf_states[i + 1][1], f[f"state{i+1}_mem"], "mem")
    for i, p in enumerate(preds):
        for k, v in p.items():
            mask = None
            if cfg.head_type == "linear" and k == "pts3d_in_self_view":
                # pos_z (linear_head.py:316) multiplies xyz by sign(z): a pixel whose regressed z is ~0 flips sign under
                # ANY rounding difference (also TF32 vs fp32 in the reference), so those pixels are excluded
                r = torch.from_numpy(f[f"pred{i}_{k}"])
                mask = (r[..., 2] > 0.03 * r.abs().max())
                assert mask.float().mean() > 0.8
            check(f"pred{i}.{k}", v, tf_preds[i][k], f[f"pred{i}_{k}"], _cls(k), mask)
        assert set(p) == {"camera_pose", "pts3d_in_self_view", "conf_self", "rgb", "pts3d_in_other_view", "conf"}
    print(f"[golden {name}] " + " | ".join(log))


def test_u8_input_and_minimal_mode_and_reference_forward_api():
    f = np.load(os.path.join(GOLD, "model_tiny_dpt.npz"))
    cfg = Cut3rConfig.from_dict(json.loads(bytes(f["config_json"]).decode()))
    sd = synth_state_dict(cfg, int(f["seed"]))
    model = Cut3rModel(cfg, sd, DEV, minimal=True)
    imgs = torch.from_numpy(f["imgs"])
    preds, _ = model.forward_window(imgs.to(DEV))             # uint8 path: normalisation fused into im2col
    log = []
    for i, p in enumerate(preds):
        assert set(p) == {"camera_pose", "pts3d_in_self_view", "conf_self"}
        for k, v in p.items():
            _check(f"u8 pred{i}.{k}", v, f[f"pred{i}_{k}"], 4e-3 if k == "camera_pose" else 6e-3, log)
    from cut3r_slam_amd.inference import inference
    from cut3r_slam_amd.track_frontend import make_views
    out, _ = inference(make_views(model, imgs), model, DEV)
    for i, p in enumerate(out["pred"]):
        _check(f"inference() pred{i}", p["pts3d_in_self_view"], f[f"pred{i}_pts3d_in_self_view"], 6e-3, log)


def test_medium_config_head_dims_48_64_vs_oracle():
    """production head widths (64 / 48 / 128-style memory heads) at a size the CPU oracle finishes in seconds."""
    cfg = Cut3rConfig(img_size=(64, 96), enc_embed_dim=256, enc_depth=3, enc_num_heads=4, dec_embed_dim=192, dec_depth=4,
                      dec_num_heads=3, state_dec_num_heads=4, state_size=30, local_mem_size=16, ray_enc_depth=1,
                      head_type="dpt", rgb_head=True)
    sd = synth_state_dict(cfg, 11)
    g = torch.Generator().manual_seed(0)
    imgs = torch.randint(0, 256, (4, 3, 64, 96), generator=g, dtype=torch.uint8)
    ref, ref_states = O.forward_views(cfg, sd, O.normalize(imgs), minimal=True, return_states=True)
    model = Cut3rModel(cfg, sd, DEV, minimal=True)
    preds, taps = model.forward_window(imgs.to(DEV), return_taps=True)
    torch.cuda.synchronize()
    log = []
    for i in range(4):
        _check(f"state{i+1}", taps["states"][i][0][None], ref_states[i + 1][0], 2.5e-3, log)      # 3 x measured (7e-4 ... 8e-4)
        _check(f"pose{i}", preds[i]["camera_pose"], ref[i]["camera_pose"], 6e-3, log)              # measured 2.0e-3
        _check(f"pts{i}", preds[i]["pts3d_in_self_view"], ref[i]["pts3d_in_self_view"], 8e-3, log)  # measured 2.5e-3
        _check(f"conf{i}", preds[i]["conf_self"], ref[i]["conf_self"], 4e-3, log)                  # measured 1.3e-3
    print("\n".join(log))


def test_fused_rope_epilogue_gives_the_bits_of_the_rope_launch():
    """Cut3rModel.fused_rope (default 1: RoPE of 64-wide heads in the q / k projection epilogue) against the stand-alone RoPE
    launches, whole window, heads of 64 (encoder, image side) and 48 (state side): every output bit-identical."""
    cfg = Cut3rConfig(img_size=(64, 96), enc_embed_dim=256, enc_depth=2, enc_num_heads=4, dec_embed_dim=192, dec_depth=3,
                      dec_num_heads=3, state_dec_num_heads=4, state_size=30, local_mem_size=16, ray_enc_depth=1, head_type="dpt", rgb_head=True)
    sd = synth_state_dict(cfg, 5)
    g = torch.Generator().manual_seed(1)
    imgs = torch.randint(0, 256, (4, 3, 64, 96), generator=g, dtype=torch.uint8).to(DEV)
    outs = []
    for mode in (0, 1, 2):
        model = Cut3rModel(cfg, sd, DEV, minimal=True)
        model.fused_rope = mode
        model.use_graphs = False
        preds, _ = model.forward_window(imgs)
        outs.append([{k: v.clone() for k, v in p.items()} for p in preds])
    torch.cuda.synchronize()
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            for k in a:
                assert torch.equal(a[k], b[k]), k


def test_decode_from_cached_features_equals_full_window():
    """encode once + decode_window(features) must give the window result (features are batch-invariant bit for bit)."""
    f = np.load(os.path.join(GOLD, "model_tiny_dpt.npz"))
    cfg = Cut3rConfig.from_dict(json.loads(bytes(f["config_json"]).decode()))
    model = Cut3rModel(cfg, synth_state_dict(cfg, int(f["seed"])), DEV, minimal=True)
    imgs = torch.from_numpy(f["imgs"]).to(DEV)
    full, _ = model.forward_window(imgs)
    full = [{k: v.clone() for k, v in p.items()} for p in full]
    feats = torch.cat([model.encode_batch(imgs[i:i + 1]) for i in range(3)], 0)          # per-frame encodes (B=1)
    assert torch.equal(feats, model.encode_batch(imgs))                                   # == batched encode (B=3)
    dec, _ = model.decode_window(feats, imgs.shape[2], imgs.shape[3])
    for a, b in zip(full, dec):
        for k in a:
            assert torch.equal(a[k], b[k]), k


def test_batched_windows_equal_individual_windows():
    """decode_windows over Wn independent windows == each window alone (GEMM rows / attention batches are independent)."""
    cfg = Cut3rConfig(img_size=(64, 96), enc_embed_dim=256, enc_depth=2, enc_num_heads=4, dec_embed_dim=192, dec_depth=4,
                      dec_num_heads=3, state_dec_num_heads=4, state_size=30, local_mem_size=16, ray_enc_depth=1, head_type="dpt")
    model = Cut3rModel(cfg, synth_state_dict(cfg, 5), DEV, minimal=True)
    g = torch.Generator().manual_seed(1)
    imgs = torch.randint(0, 256, (9, 3, 64, 96), generator=g, dtype=torch.uint8).to(DEV)
    feats = model.encode_batch(imgs)                                    # [9,N,E]
    wins = torch.stack([feats[0:3], feats[3:6], feats[6:9]], 0)         # 3 windows of 3 views
    res = model.decode_windows(wins, 64, 96)
    res = {k: v.clone() for k, v in res.items()}
    for w in range(3):
        single, _ = model.decode_window(wins[w], 64, 96)
        for v in range(3):
            for k in ("pts3d_in_self_view", "conf_self", "camera_pose"):
                a, b = res[k][w * 3 + v], single[v][k][0]
                assert torch.allclose(a, b, rtol=1e-4, atol=1e-5), (w, v, k, float((a - b).abs().max()))


def test_pair_launch_form_of_a_decoder_layer_gives_the_same_bits():
    """`Cut3rModel.pair_rows`: below that many rows a decoder layer runs as pair launches (state + image projection in one grid of 64 x 64
    tiles: cut3r_gemm_f16_pair with the LayerNorm fold, the fused RoPE of the 64-wide heads and the compile-time epilogues).  Same tile
    body, same row arithmetic: every output equals the default form bit for bit, for one window and for three batched ones."""
    cfg = Cut3rConfig(img_size=(64, 96), enc_embed_dim=256, enc_depth=2, enc_num_heads=4, dec_embed_dim=192, dec_depth=4,
                      dec_num_heads=3, state_dec_num_heads=4, state_size=30, local_mem_size=16, ray_enc_depth=1, head_type="dpt")
    sd = synth_state_dict(cfg, 5)
    g = torch.Generator().manual_seed(1)
    imgs = torch.randint(0, 256, (9, 3, 64, 96), generator=g, dtype=torch.uint8).to(DEV)
    outs = []
    for rows in (0, 10 ** 9):
        model = Cut3rModel(cfg, sd, DEV, minimal=True)
        model.pair_rows = rows
        feats = model.encode_batch(imgs)
        wins = torch.stack([feats[0:3], feats[3:6], feats[6:9]], 0)
        res = {k: v.clone() for k, v in model.decode_windows(wins, 64, 96).items()}
        single, _ = model.decode_window(wins[1], 64, 96)
        outs.append((res, [{k: v.clone() for k, v in p.items()} for p in single]))
        del model
    for k in ("pts3d_in_self_view", "conf_self", "camera_pose"):
        assert torch.equal(outs[0][0][k], outs[1][0][k]), k
        for v in range(3):
            assert torch.equal(outs[0][1][v][k], outs[1][1][v][k]), (k, v)
