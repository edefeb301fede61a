#!/usr/bin/env python3
"""Throughput of the CUT3R-SLAM tracking hot path on MI355X (BASELINE.json metric: frames/s on 640x480 input).

Workload (BASELINE configs[1], "Replica room0 640x480: ViT pointmap + factor-graph step, tracking only, GS off"):
synthetic 640x480 stream -> tracking resolution 384x512 (demo_s.py:69-73), production-shape network (ViT-L encoder,
768-d dual decoder, DPT head; seeded random weights -- no checkpoint exists), fixed keyframe cadence kf_every=10
(hislam2/motion_filter.py:83,109,124).  One STEP = one steady-state tracking window = 50 input frames:
5 keyframe-filter encoder passes + one 6-view window inference + chaining/alignment + covisibility-graph update
of the 5 new keyframes (hislam2/hi2.py:101-133 without the GS mapper).  Frames are resident in HBM before the
timed region.  value = frames processed by all ranks / max-over-ranks time.

N > 1 (one process per GPU, torch.distributed/RCCL): windows are sharded across ranks (every window re-initialises
the recurrent state, src/dust3r/model.py:819-822, so windows are independent network evaluations); each step every
rank infers ONE window, the three consumed outputs are all-gathered over xGMI and the cheap sequential chaining +
graph update of all N windows runs replicated.  Per-GPU work is fixed => "scaling": "weak".
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def synth_frames(n, H, W, device, seed=0):
    """seeded low-pass noise panned smoothly: u8 [n,3,H,W] on the GPU (data: synthetic)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    base = torch.rand(3, H // 8 + 64, W // 8 + 64, generator=g)
    base = torch.nn.functional.interpolate(base[None], scale_factor=8, mode="bilinear", align_corners=False)[0]
    base = (base * 255).to(device)
    frames = torch.empty(n, 3, H, W, dtype=torch.uint8, device=device)
    for t in range(n):
        dx, dy = int(2 * t) % 400, int(1 * t) % 300
        frames[t] = base[:, dy:dy + H, dx:dx + W].round().clamp(0, 255).to(torch.uint8)
    return frames


class GemmProbe:
    """HIP-event timing of every launch of the two large-tile GEMM kernels (128^2 and 256^2) on the launch stream."""

    KERNELS = {128: "gemm_kernel<128,128,2,4,2> (v_mfma_f32_16x16x32_f16, 8 waves, 2 workgroups/CU)",
               256: "gemm256_kernel (256x256x64, v_mfma_f32_16x16x32_f16, 8 waves ping-pong, 1 workgroup/CU)"}

    def __init__(self):
        self.ev = {128: [], 256: []}
        self.flops = {128: 0.0, 256: 0.0}
        self.bytes = {128: 0.0, 256: 0.0}

    def install(self):
        from cut3r_slam_amd import _lib
        lib = _lib.load()
        raw = lib.cut3r_gemm_f16
        probe = self

        def wrapped(dref, stream):
            d = dref._obj
            tile = lib.cut3r_gemm_tile_for(dref)
            if tile in (128, 256):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                rc = raw(dref, stream)
                e.record()
                probe.ev[tile].append((s, e))
                nb = max(d.batch, 1)
                probe.flops[tile] += 2.0 * d.M * d.N * d.K * nb
                a_bytes = d.M * d.Cin * 2 if d.conv_k == 3 else d.M * d.K * 2        # a conv input is read once
                probe.bytes[tile] += nb * (a_bytes + d.N * d.K * 2 + d.M * d.N * (2 if d.out_f16 else 4)
                                           + (d.M * d.N * (2 if d.res1_f16 else 4) if d.res1 else 0))
                return rc
            return raw(dref, stream)

        self._lib, self._raw = lib, raw
        lib.cut3r_gemm_f16 = wrapped

    def remove(self):
        self._lib.cut3r_gemm_f16 = self._raw

    def result(self):
        """{tile: (launches, total ms, flops, algorithmic bytes)}"""
        torch.cuda.synchronize()
        return {t: (len(ev), sum(s.elapsed_time(e) for s, e in ev), self.flops[t], self.bytes[t]) for t, ev in self.ev.items()}


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def host_cores():
    """threads we may actually use: the affinity mask, capped at the GPU box's per-GPU CPU share"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--small", action="store_true", help="debug: tiny network (NOT a valid benchmark line)")
    ap.add_argument("--window-batch", type=int, default=8, help="tracking windows pushed through the decoder together "
                    "(buffered-stream throughput mode; 1 = the reference's one-window-at-a-time schedule)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    emu = int(os.environ.get("CUT3R_EMULATE_WORLD", "0"))     # debug: rank 0 of an `emu`-GPU job on ONE GPU (replay load only)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("CUT3R_DIST_BACKEND", "nccl") != "nccl":
        local_rank = 0                 # functional run: all ranks share GPU 0
    dist_on = world > 1 or os.environ.get("CUT3R_FORCE_DIST") == "1"     # the env flag rehearses the RCCL path with one rank
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    if dist_on:
        import torch.distributed as dist
        backend = os.environ.get("CUT3R_DIST_BACKEND", "nccl")       # "gloo": functional multi-rank run with every rank on ONE GPU
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29577")
            dist.init_process_group(backend, rank=0, world_size=1, **({"device_id": torch.device(dev)} if backend == "nccl" else {}))
        else:
            dist.init_process_group(backend, **({"device_id": torch.device(dev)} if backend == "nccl" else {}))

    from cut3r_slam_amd.config import production_config, tiny_config
    from cut3r_slam_amd.model import Cut3rModel
    from cut3r_slam_amd.slam import Cut3rSlam
    from cut3r_slam_amd.weights import synth_state_dict
    from cut3r_slam_amd import dist as cdist

    H, W, KF_EVERY, WIN = 384, 512, 10, 5
    WB = max(1, args.window_batch)
    cfg = tiny_config("dpt") if args.small else production_config()
    t0 = time.time()
    torch.set_num_threads(host_cores())
    sd = synth_state_dict(cfg, seed=0)
    log(f"weights synthesised in {time.time() - t0:.1f}s")
    model = Cut3rModel(cfg, sd, dev, minimal=True)
    torch.cuda.synchronize()
    t_build = time.time() - t0
    log(f"model resident in HBM after {t_build:.1f}s")

    if emu > 1:
        world = emu                                # after the process-group decisions above: no collective is created
    frames_per_step = KF_EVERY * WIN * WB          # per rank: one step = WB windows, pushed through the network together
    total_steps = args.warmup + args.steps
    probe_steps = 0 if (args.no_roofline or dist_on) else args.steps      # second, instrumented pass
    n_kf = 7 + WIN * WB * world * (total_steps + probe_steps) + 2
    config = {"Tracking": {"motion_filter": {"thresh": 0.9, "skip": 5, "skip_blur": False, "kf_every": KF_EVERY},
                           "frontend": {"iteration": 0, "window_batch": 1}}}
    slam = Cut3rSlam(model, config, (H, W), buffer=n_kf + 8, device=dev)
    intr = torch.tensor([600.0 * W / 1200.0, 600.0 * H / 680.0, 599.5 * W / 1200.0, 339.5 * H / 680.0])  # calib/replica.txt scaled
    # rank r owns windows [r*WB, (r+1)*WB) of every step; chaining + graph update of step s overlap the network pass of step s+1
    runner = cdist.ShardedTracker(slam, world, rank, wb=WB, pipelined=os.environ.get("CUT3R_PIPELINE", "1") == "1",
                                  force_collective=dist_on)
    runner.emulate_gather = emu > 1
    frames = synth_frames(runner.frames_needed(total_steps + probe_steps, KF_EVERY, WIN), H, W, dev, seed=0)

    # prologue (untimed): the 6-keyframe initialisation window
    log(f"{frames.shape[0]} synthetic frames resident; running the initialisation window")
    t = 0
    while not slam.keyframes.is_initialized:
        slam.run(t, frames[t:t + 1], intr, frames[t:t + 1], intr)
        t += 1
    torch.cuda.synchronize()
    log("initialised; warmup")

    def one_step(t):
        return runner.step(frames, t, KF_EVERY, WIN, intr)

    def barrier():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        t = one_step(t)
    runner.flush()
    barrier()
    log("timed region")
    for k in runner.stats:
        runner.stats[k] = 0
    tic = time.perf_counter()
    for _ in range(args.steps):
        t = one_step(t)
    runner.flush()                       # the timed region holds exactly K network passes and K replays
    barrier()
    elapsed = time.perf_counter() - tic
    if dist_on:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    frames_total = frames_per_step * args.steps * world
    value = frames_total / elapsed
    log(f"timed region done: {elapsed:.3f}s for {frames_total} frames -> {value:.1f} frames/s")
    from cut3r_slam_amd.track_frontend import TIMING
    log(f"replay round trips per window [ms]: log-depth {1e3 * TIMING['sync1_s'] / max(1, TIMING['windows']):.2f}, "
        f"window update {1e3 * TIMING['sync2_s'] / max(1, TIMING['windows']):.2f}")
    log("host wall-clock per step [ms]: " + ", ".join(f"{k[:-2]} {1e3 * v / max(1, runner.stats['steps']):.2f}" for k, v in runner.stats.items() if k != "steps"))

    roofline, cpu_base = None, None
    if rank == 0 and not args.no_roofline and not dist_on:
        # second, instrumented pass over the same number of steps: HIP events around every launch of the dominant
        # kernel (tile-128 MFMA GEMM: encoder linears + DPT convolutions) on the launch stream
        need = frames_per_step * args.steps
        if t + need + 1 <= frames.shape[0]:
            probe = GemmProbe()
            probe.install()
            model.use_graphs = False      # events must bracket live launches, not a graph replay
            for _ in range(args.steps):
                t = one_step(t)
            runner.flush()
            res = probe.result()
            probe.remove()
            model.use_graphs = True
            dom = max(res, key=lambda tk: res[tk][1])          # the kernel with the largest total time in this workload
            n, ms, fl, by = res[dom]
            if n:
                ach = fl / (ms * 1e-3) / 1e12
                traffic = None
                pj = os.path.join(ROOT, "profiles", "r01", f"pmc_traffic_gemm{dom}.json")
                if os.path.isfile(pj):          # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command (tools/pmc_traffic.py)
                    traffic = json.load(open(pj)).get("traffic_bytes_per_launch")
                other = {}
                for tk, (n2, ms2, fl2, by2) in res.items():
                    if tk != dom and n2:
                        other = {"kernel": GemmProbe.KERNELS[tk], "launches": n2, "avg_launch_us": round(ms2 * 1e3 / n2, 2),
                                 "achieved": round(fl2 / (ms2 * 1e-3) / 1e12, 2), "frac": round(fl2 / (ms2 * 1e-3) / 1e12 / 2500.0, 4),
                                 "share_of_large_gemm_time": round(ms2 / (ms + ms2), 3)}
                roofline = {"bound": "mfma", "kernel": GemmProbe.KERNELS[dom], "achieved": round(ach, 2),
                            "peak": 2500.0, "unit": "TFLOP/s", "frac": round(ach / 2500.0, 4), "traffic": traffic,
                            "launches": n, "avg_launch_us": round(ms * 1e3 / n, 2), "flops_per_launch": fl / n,
                            "algorithmic_bytes_per_launch": by / n, "second_kernel": other}
    if rank == 0 and world == 1 and emu <= 1 and not args.no_cpu_baseline and not args.small:
        log("cpu baseline (oracle on host cores)")
        cpu_base = cpu_baseline(cfg, sd, frames[:2].cpu(), frames_per_step)
        log("cpu baseline done")

    dump = os.environ.get("CUT3R_DUMP_STATE")
    if dump:                               # tests: the replicated result of every rank
        import numpy as np
        k = slam.tracker.t1
        ii, jj, age = slam.graph.edges_numpy()
        np.savez(f"{dump}.rank{rank}.npz", pose=slam.keyframes.pose[:k].numpy(), depth_sum=slam.keyframes.depth[:k].double().sum(dim=(1, 2)).cpu().numpy(),
                 w2c=slam.keyframes.w2c[:k].cpu().numpy(), ii=ii, jj=jj, k=k)
    if rank == 0:
        out = {
            "metric": "frames/sec (ViT pointmap + covisibility-graph tracking step) on 640x480" + (f" [DEBUG: rank 0 of an emulated {emu}-GPU job]" if emu > 1 else ""), "value": round(value, 2),
            "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"Replica-shaped 640x480 stream -> 384x512 tracking res; kf_every=10; step = {WB} window(s) "
                                   f"= {frames_per_step} frames: each new keyframe through the ViT-L encoder once (batched), "
                                   "6-view recurrent decoder + DPT head per window (windows batched through the decoder), "
                                   "log-depth/pose chaining + covisibility-graph update per keyframe; "
                                   "ViT-L/24 enc, 768/12 dual decoder, DPT head, random init; GS backend off"
                                   + (" [DEBUG --small]" if args.small else ""),
                       "frames_per_step": frames_per_step, "window_views": 6, "window_batch": WB, "parallelism": f"window-sharded x{world}"},
            "roofline": roofline, "cpu_baseline": cpu_base, "build_s": round(t_build, 1),
        }
        print(json.dumps(out))
    if dist_on:
        dist.destroy_process_group()


def cpu_baseline(cfg, sd, imgs_u8, frames_per_step):
    """Oracle (kind 'port') timed on the host cores on a BOUNDED sample: one keyframe-filter encode + one 2-view window
    at 384x512, extrapolated to a step (5 encodes + 6 views; the model cost is linear in views)."""
    from oracle import cut3r_oracle as O
    cores = host_cores()
    torch.set_num_threads(cores)
    x = O.normalize(imgs_u8)
    with torch.no_grad():
        t0 = time.perf_counter()
        O.encode_image(cfg, sd, x[:1])
        t_enc = time.perf_counter() - t0
        t0 = time.perf_counter()
        O.forward_views(cfg, sd, x[:2], minimal=True)
        t_win2 = time.perf_counter() - t0
    # same algorithmic work as the GPU path per 50-frame window: 5 new keyframes encoded once + 6 views decoded
    # (the 2-view sample contains 2 encodes + 2 decodes; the model cost is linear in views)
    t_dec_view = max(t_win2 - 2 * t_enc, 0.0) / 2
    window_s = 5 * t_enc + 6 * t_dec_view
    return {"value": round(50.0 / window_s, 3), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"oracle/cut3r_oracle.py fp32: 1 encode_image ({t_enc:.2f} s) + one 2-view window ({t_win2:.2f} s) at "
                      f"384x512, extrapolated to a 50-frame window = 5 encodes + 6 decoder/head views"}


if __name__ == "__main__":
    main()
