#!/usr/bin/env python3
"""Tracking-only run over an image directory with the reference's command line (/root/reference/demo_s.py:117-172):

    python demo.py --imagedir data/Replica/room0/colors --calib calib/replica.txt --config config/replica_config.yaml \\
                   --output outputs/room0 [--kf_every 10] [--ckpt_path checkpoints/cut3r_512_dpt_4_64.pth]

Writes <output>/traj_kf.txt and <output>/intrinsics.npy exactly as the reference does (evo-compatible TUM rows).  The
Gaussian-splatting mapper and its viewers are out of scope: --droidvis/--gsvis/--gtdepthdir/--posedir/
--weights are accepted and ignored.  Without a checkpoint, `--synthetic-weights` runs the production-shape network with
seeded random weights (throughput / plumbing runs only).
"""
import argparse
import os
import sys
import time

import torch
import yaml

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def load_config(path):
    """hislam2/util/utils.py `load_config`: YAML with an optional `inherit_from` parent"""
    with open(path, "r") as f:
        cfg = yaml.safe_load(f) or {}
    parent = cfg.get("inherit_from")
    if parent:
        base = load_config(parent if os.path.isabs(parent) else os.path.join(os.path.dirname(path), parent))

        def merge(a, b):
            for k, v in b.items():
                if isinstance(v, dict) and isinstance(a.get(k), dict):
                    merge(a[k], v)
                else:
                    a[k] = v
            return a
        cfg = merge(base, cfg)
    return cfg


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--imagedir", type=str, required=True, help="path to image directory")
    p.add_argument("--posedir", type=str, default=None)
    p.add_argument("--calib", type=str, required=True, help="path to calibration file")
    p.add_argument("--config", type=str, default=None, help="path to configuration file")
    p.add_argument("--output", default="outputs/demo", help="path to save output")
    p.add_argument("--gtdepthdir", type=str, default=None)
    p.add_argument("--weights", default=None)
    p.add_argument("--buffer", type=int, default=-1, help="number of keyframes to buffer (default: 1/5 of total frames + 150)")
    p.add_argument("--undistort", action="store_true")
    p.add_argument("--cropborder", type=int, default=0)
    p.add_argument("--droidvis", action="store_true")
    p.add_argument("--gsvis", action="store_true")
    p.add_argument("--start", type=int, default=0)
    p.add_argument("--length", type=int, default=100000)
    p.add_argument("--ckpt_path", type=str, default="./checkpoints/cut3r_512_dpt_4_64.pth")
    p.add_argument("--kf_every", type=int, default=-1)
    p.add_argument("--synthetic-weights", action="store_true", help="seeded random weights of the production architecture")
    p.add_argument("--seed", type=int, default=0, help="seed of --synthetic-weights")
    p.add_argument("--small", action="store_true", help="debug: tiny architecture (with --synthetic-weights)")
    p.add_argument("--window-batch", type=int, default=1, help="tracking windows decoded together (1 = reference schedule)")
    p.add_argument("--lookahead", type=int, default=1, help="overlap mode (kf_every = -1): hold back LOOKAHEAD x skip frames, encode their tested "
                   "frames as one batch and take the keyframe decisions on the device (1 = the reference's frame-by-frame loop; identical "
                   "results, latency LOOKAHEAD x skip frames; with --window-batch the found keyframes are tracked several windows at a time)")
    p.add_argument("--device", default="cuda:0")
    p.add_argument("--gs-final-iters", type=int, default=None, help="iterations of the mapper's closing global BA (default: the configured "
                   "position_lr_max_steps, as the reference; 0 skips it)")
    p.add_argument("--gs", action="store_true", help="attach the Gaussian-splatting mapper (hislam2/hi2.py:47-48 always does; needs the "
                   "Mapping / Training / opt_params sections of the config, config/scannet_config.yaml:44-79)")
    args = p.parse_args(argv)
    os.makedirs(args.output, exist_ok=True)

    from cut3r_slam_amd import stream
    from cut3r_slam_amd.config import production_config, tiny_config
    from cut3r_slam_amd.model import Cut3rModel
    from cut3r_slam_amd.slam import Cut3rSlam, DEFAULT_CONFIG
    from cut3r_slam_amd.weights import synth_state_dict

    cfg = load_config(args.config) if args.config else {}
    trk = cfg.setdefault("Tracking", {})
    for k, v in DEFAULT_CONFIG["Tracking"].items():
        sec = trk.setdefault(k, {})
        for kk, vv in v.items():
            sec.setdefault(kk, vv)
    if args.kf_every > 0:
        trk["motion_filter"]["kf_every"] = args.kf_every
    trk["frontend"]["window_batch"] = args.window_batch

    if args.synthetic_weights:
        mcfg = tiny_config("dpt") if args.small else production_config()
        model = Cut3rModel(mcfg, synth_state_dict(mcfg, seed=args.seed), args.device, minimal=True)
    else:
        model = Cut3rModel.from_pretrained(args.ckpt_path, device=args.device, minimal=True)

    n_files = len(os.listdir(args.imagedir))
    buffer = min(1000, n_files // 5 + 150) if args.buffer < 0 else args.buffer
    slam, t0, nframes = None, time.time(), 0
    frames = stream.mono_stream(args.imagedir, args.calib, args.undistort, args.cropborder, args.start, args.length, device=args.device)

    def make_slam(image_ds, intr_ds):
        s_ = Cut3rSlam(model, cfg, (image_ds.shape[2], image_ds.shape[3]), buffer=buffer, device=args.device)
        if args.gs:
            from cut3r_slam_amd.gs_mapper import GSMapper
            if "Training" not in cfg or "opt_params" not in cfg:
                raise SystemExit("--gs needs the Training and opt_params sections in --config")
            k = [float(v) for v in intr_ds[0].reshape(-1)[:4]]
            s_.mapper = GSMapper(cfg, k[0], k[1], k[2], k[3], downsample_ratio=s_.downsample_ratio, device=args.device)
        return s_

    def items():
        # demo_s.py:158-159: a --length cut does NOT flush the tail window (the flags look at the directory, not at the cut)
        for t, image, intr, image_ds, intr_ds, is_last in frames:
            yield (t, image, intr[0].float(), image_ds, intr_ds[0].float(), t + args.start == n_files - 2, t + args.start == n_files - 1)

    if args.lookahead > 1:
        it = items()
        first = next(it, None)
        if first is not None:
            slam = make_slam(first[3], first[4][None])
            counted = []

            def chain():
                yield first
                yield from it
            slam.run_stream(chain(), lookahead=args.lookahead, on_frame=lambda t_, out_: counted.append(t_))
            nframes = len(counted)
    else:
        for item in items():
            if slam is None:
                slam = make_slam(item[3], item[4][None])
            slam.run(item[0], item[1], item[2], item[3], item[4], second_last_frame=item[5], last_frame=item[6])
            nframes += 1
    if slam is None:
        raise SystemExit(f"{args.imagedir}: no frames")
    torch.cuda.synchronize()
    traj = stream.save_trajectory(slam, args.imagedir, args.output, start=args.start)
    if slam.mapper is not None and slam.mapper.viewpoints:
        slam.terminate(add_kf=False, finalize_iters=args.gs_final_iters)  # demo_s.py:171 (the trajectory above is the tracker's, as demo_s.py:168-169)
        ev = slam.mapper.eval_rendering_kf()                              # demo_s.py:175-190 evaluates the renderings of the keyframes
        slam.mapper.save(os.path.join(args.output, "gaussians.safetensors"))
        print(f"GS mapper: {len(slam.mapper.gaussians)} Gaussians, {len(slam.mapper.viewpoints)} keyframes, PSNR {ev['mean_psnr']:.2f} dB, "
              f"SSIM {ev['mean_ssim']:.4f} -> {args.output}/gaussians.safetensors")
    print(f"{nframes} frames, {len(traj)} keyframes, {len(slam.graph.edges_numpy()[0])} graph edges in {time.time() - t0:.1f}s "
          f"-> {args.output}/traj_kf.txt")
    return 0


if __name__ == "__main__":
    sys.exit(main())
