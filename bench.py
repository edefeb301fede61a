#!/usr/bin/env python3
"""Throughput + trajectory parity of the CUT3R-SLAM tracking hot path on MI355X (BASELINE.json metric: frames/s on 640x480
input; ATE-RMSE vs the reference path).

Headline `value` (BASELINE configs[1], "Replica room0 640x480: ViT pointmap + factor-graph step, tracking only, GS off"):
synthetic 640x480 stream -> tracking resolution 384x512 (demo_s.py:69-73), production-shape network (ViT-L encoder, 768-d
dual decoder, DPT head; seeded random weights -- no checkpoint exists), BUFFERED FIXED-CADENCE mode: kf_every=10
(hislam2/motion_filter.py:83,109,124; SURVEY 8(d) names it the throughput schedule because its keyframes do not depend on
feature numerics), `--window-batch` (default 28) tracking windows pushed through the decoder together.  One STEP =
window_batch steady-state windows = window_batch*50 frames: each new keyframe through the encoder once, 6-view recurrent
decoder + DPT head per window, chaining/alignment + covisibility-graph update per keyframe (hislam2/hi2.py:101-133 without
the GS mapper).  The stream is a succession of Replica-shaped SEQUENCES (`--sequence-windows`, default 40 windows = 2000 frames,
whatever the number of GPUs): the last keyframe of a sequence is keyframe 0 of the next one, whose first window is handled like the
reference's initialisation window, with an empty covisibility graph (TrackFrontend.sequence_windows: every sequence equals a
fresh run over its frames, tests/test_slam_gpu.py).  One resident recording is read cyclically; frames are in HBM before the
timed region.  value = frames of all ranks / max-over-ranks time.

Beside it, in the same JSON line (rank 0, N = 1):
  operating_points   the reference's own schedules on the same network: fixed cadence with ONE window at a time
                     (window_batch=1), the default batch over ONE endless stream (no sequence cuts: every keyframe is tested
                     against all earlier ones), and the maintained configs' OVERLAP mode (kf_every=-1, skip=5, thresh=0.9,
                     config/scannet_config.yaml:20-25): encoder + patch-overlap test every 5th frame, one window at a time,
                     through Cut3rSlam.run -- frame by frame, and with the batched look-ahead of the buffered driver; and the GS
                     mapper (BASELINE config 5's backend) on a synthetic 6-keyframe window: seconds, ms per render iteration, PSNR
  trajectory_parity  the metric's second half: Cut3rSlam on HIP vs the CPU restatement of the reference loop
                     (oracle/slam_run.py) on the same seeded stream and weights, medium config, both keyframe modes:
                     Sim(3)-aligned ATE-RMSE (evo_ape -vas semantics), keyframe agreement, edge-list equality
  roofline           dominant MFMA GEMM kernel (+ the other large-tile GEMM and the attention kernels), HIP-event timed
  cpu_baseline       the oracle timed on the host cores on a bounded sample of the same loop

N > 1 (one process per GPU, torch.distributed/RCCL): windows are sharded across ranks (every window re-initialises the
recurrent state, src/dust3r/model.py:819-822); per window 352 bytes of scalars are all-gathered, the chain is a replicated host
scan, the stride-2 stores are completed by in-place all-gathers and the overlap counts by one all-reduce (cut3r_slam_amd/dist.py).
Per-GPU work is fixed => "scaling": "weak".
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H, W, KF_EVERY, WIN = 384, 512, 10, 5
PEAK_F16 = 2500.0          # TFLOP/s dense fp16/bf16 MFMA (MI355X_MICROARCH.md)


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


H0, W0 = 480, 640          # the camera's frame size (BASELINE metric: "on 640x480"); demo_s.py:69-73 resizes to 384x512 for tracking


def synth_camera_frames(n, device, seed=0):
    """the same panned texture as raw camera frames: u8 [n,480,640,3] (cv2.imread layout) resident on the GPU"""
    g = torch.Generator(device="cpu").manual_seed(seed)
    base = torch.rand(3, H0 // 8 + 64, W0 // 8 + 64, generator=g)
    base = torch.nn.functional.interpolate(base[None], scale_factor=8, mode="bilinear", align_corners=False)[0]
    base = (base * 255).round().clamp(0, 255).to(torch.uint8).permute(1, 2, 0).contiguous().to(device)
    frames = torch.empty(n, H0, W0, 3, dtype=torch.uint8, device=device)
    for t in range(n):
        dx, dy = int(2 * t) % 400, int(1 * t) % 300
        frames[t] = base[dy:dy + H0, dx:dx + W0]
    return frames


def to_tracking(frame_hwc):
    """one raw frame u8 [1,H0,W0,3] -> [1,3,384,512] (cut3r_resize_linear_u8 = cv2.resize INTER_LINEAR)"""
    from cut3r_slam_amd import ops
    return ops.resize_linear_u8(frame_hwc[0].contiguous(), H, W, chw_out=True)[None]


class FrameLoop:
    """a resident recording of `base.shape[0]` frames read as an endless stream: frame f is base[f % period] (slices may not wrap:
    the drivers read single frames and short runs)"""

    def __init__(self, base, virtual_len):
        self.base, self.period = base, base.shape[0]
        self.shape = (int(virtual_len),) + tuple(base.shape[1:])
        self.is_cuda, self.device, self.dtype = base.is_cuda, base.device, base.dtype

    def __getitem__(self, key):
        if isinstance(key, slice):
            a = 0 if key.start is None else key.start
            b = self.shape[0] if key.stop is None else key.stop
            s = a % self.period
            if key.step not in (None, 1) or s + (b - a) > self.period:
                raise IndexError("FrameLoop: contiguous slices inside one period only")
            return self.base[s:s + (b - a)]
        return self.base[int(key) % self.period]


class KernelProbe:
    """HIP-event timing, on the launch stream, of every launch of the large-tile GEMM kernels and of the fused attention
    kernels (the C-ABI entry points are wrapped; events bracket live launches of an eager pass, never a graph replay)."""

    GEMM = {128: "gemm_kernel<128,128,2,4,2> (v_mfma_f32_16x16x32_f16, 8 waves, 2 workgroups/CU)",
            256: "gemm256_kernel (256x256x64, v_mfma_f32_16x16x32_f16, 8 waves ping-pong, 1 workgroup/CU)",
            64: "gemm_kernel<64,64> (4 waves)", 192: "gemm_kernel<192,128> (DPT 3x3 convolutions, 128 output channels)",
            128192: "gemm_kernel<128,192> (48-wide heads)", 16: "gemm_skinny_kernel (M <= 64: weight stream, HBM-bound)"}

    def __init__(self, all_tiles=False):
        self.rec = {}          # key -> [events, flops, bytes]
        self.all_tiles = all_tiles

    def _add(self, key, s, e, flops, nbytes):
        r = self.rec.setdefault(key, [[], 0.0, 0.0])
        r[0].append((s, e))
        r[1] += flops
        r[2] += nbytes

    def install(self):
        from cut3r_slam_amd import _lib
        lib = _lib.load()
        raw_gemm, raw_attn = lib.cut3r_gemm_f16, lib.cut3r_attention_f16
        probe = self

        def gemm(dref, stream):
            d = dref._obj
            tile = lib.cut3r_gemm_tile_for(dref)
            if tile not in (128, 256) and not probe.all_tiles:
                return raw_gemm(dref, stream)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            rc = raw_gemm(dref, stream)
            e.record()
            nb = max(d.batch, 1)
            a_bytes = d.M * d.Cin * 2 if d.conv_k == 3 else d.M * d.K * 2        # a conv input is read once
            fl = 2.0 * d.M * d.N * d.K * nb
            by = nb * (a_bytes + d.N * d.K * 2 + d.M * d.N * (2 if d.out_f16 else 4) + (d.M * d.N * (2 if d.res1_f16 else 4) if d.res1 else 0))
            probe._add(("gemm", tile), s, e, fl, by)
            probe._add(("shape", tile, d.M, d.N, d.K, int(d.conv_k), "f16" if d.out_f16 else "f32", "res" if d.res1 else "", int(d.act)), s, e, fl, by)
            return rc

        def attn(q, k, v, o, B, Hh, Nq, Nk, D, *rest):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            rc = raw_attn(q, k, v, o, B, Hh, Nq, Nk, D, *rest)
            e.record()
            if Nq >= 128 and Nk >= 128:
                probe._add(("attn", D, "enc" if D == 64 and Hh == 16 else "dec"), s, e, 4.0 * B * Hh * Nq * Nk * D,
                           2.0 * B * Hh * D * (2 * Nq + 2 * Nk))
            return rc

        self._lib, self._raw = lib, (raw_gemm, raw_attn)
        lib.cut3r_gemm_f16, lib.cut3r_attention_f16 = gemm, attn

    def remove(self):
        self._lib.cut3r_gemm_f16, self._lib.cut3r_attention_f16 = self._raw

    def result(self):
        """{key: (launches, total ms, flops, algorithmic bytes)}"""
        torch.cuda.synchronize()
        return {k: (len(ev), sum(s.elapsed_time(e) for s, e in ev), fl, by) for k, (ev, fl, by) in self.rec.items()}


PEAK_HBM = 8000.0          # GB/s (MI355X_MICROARCH.md)


def shape_entries(res, top=14):
    """the GEMM launches of an instrumented pass by shape: algorithmic FLOPs and bytes per launch, the time each roof alone would allow
    (dense fp16 MFMA peak, HBM peak), which one binds, and the achieved fraction of THAT roof -- short-K projections with an fp32
    residual read + write are HBM-bound, not MFMA-bound"""
    out = []
    sh = {k: v for k, v in res.items() if k[0] == "shape" and v[0]}
    tot = sum(v[1] for v in sh.values())
    for k, (n, ms, fl, by) in sorted(sh.items(), key=lambda kv: -kv[1][1])[:top]:
        us = ms * 1e3 / n
        t_mfma, t_hbm = fl / n / (PEAK_F16 * 1e12) * 1e6, by / n / (PEAK_HBM * 1e9) * 1e6
        out.append({"tile": k[1], "M": k[2], "N": k[3], "K": k[4], "conv": k[5], "out": k[6], "residual": bool(k[7]), "act": k[8], "launches": n,
                    "avg_launch_us": round(us, 2), "share_of_gemm_time": round(ms / tot, 4), "tflops": round(fl / n / us / 1e6, 1),
                    "algorithmic_gb_per_s": round(by / n / us / 1e3, 1), "roof_us_mfma": round(t_mfma, 2), "roof_us_hbm": round(t_hbm, 2),
                    "bound": "hbm" if t_hbm > t_mfma else "mfma", "frac_of_binding_roof": round(max(t_mfma, t_hbm) / us, 4)})
    return out


def probe_entries(res):
    """[{kernel, launches, avg_launch_us, achieved TFLOP/s, frac of the dense fp16 MFMA peak}] of an instrumented pass, by total time"""
    out = []
    for k, (n, ms, fl, by) in sorted(res.items(), key=lambda kv: -kv[1][1]):
        if not n or ms <= 0 or k[0] == "shape":
            continue
        name = KernelProbe.GEMM.get(k[1], f"gemm tile {k[1]}") if k[0] == "gemm" else \
            (f"attn_pipe_kernel<{k[1]},1,4,4>" if k[1] in (48, 64) else f"attn_kernel<{k[1]},4>") + \
            f" ({'encoder self-attention' if k[2] == 'enc' else 'decoder self/cross attention'})"
        out.append({"kernel": name, "bound": "mfma" if not (k[0] == "gemm" and k[1] == 16) else "hbm", "launches": n,
                    "avg_launch_us": round(ms * 1e3 / n, 2), "total_ms": round(ms, 3), "achieved": round(fl / (ms * 1e-3) / 1e12, 2), "peak": PEAK_F16,
                    "unit": "TFLOP/s", "frac": round(fl / (ms * 1e-3) / 1e12 / PEAK_F16, 4), "flops_per_launch": fl / n})
    return out


def mfma_probe(dev, seconds=1.2):
    """what the matrix pipe of THIS device sustains on operands that toggle: cut3r_mfma_probe (bare v_mfma_f32_16x16x32_f16 loop, two waves
    per SIMD on every CU) launched back to back for `seconds`, HIP events around the last launches, in-kernel clock from the s_memtime /
    s_memrealtime stamps of the last one.  Reported BESIDE the nominal 2.5 PFLOP/s peak, never instead of it."""
    import ctypes
    from cut3r_slam_amd import _lib
    lib = _lib.load()
    grid, iters = 256, 6000
    out = {}
    for label, data in (("random_operands", torch.randn(1 << 20, device=dev).half()), ("zero_operands", torch.zeros(1 << 20, device=dev, dtype=torch.float16))):
        sink = torch.empty(grid * 512, device=dev)
        stamps = torch.zeros(grid, 2, dtype=torch.int64, device=dev)
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

        def launch():
            rc = lib.cut3r_mfma_probe(ctypes.c_void_p(data.data_ptr()), data.numel(), iters, grid, ctypes.c_void_p(sink.data_ptr()),
                                      ctypes.c_void_p(stamps.data_ptr()), st)
            if rc != 0:
                raise RuntimeError(f"cut3r_mfma_probe: error {rc}")
        launch()
        torch.cuda.synchronize()
        t0 = time.time()
        n = 0
        while time.time() - t0 < seconds:
            for _ in range(20):
                launch()
            torch.cuda.synchronize()
            n += 20
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            launch()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        s_ = stamps.cpu().double()
        clk = float((s_[:, 0] / s_[:, 1].clamp_min(1)).median()) * 0.1                 # GHz: shader cycles per 100-MHz tick
        tf = grid * 8 * iters * 16 * 16384 / (us * 1e-6) / 1e12
        out[label] = {"in_kernel_clock_ghz": round(clk, 3), "tflops": round(tf, 1), "us_per_launch": round(us, 1), "launches_before": n}
    return out


def guarded(label, fn, *a, **k):
    """a secondary leg (operating point, parity table, CPU baseline) must not take the headline line with it: its failure is logged
    with the traceback and reported in its place"""
    try:
        return fn(*a, **k)
    except Exception as ex:
        import traceback
        log(f"{label} FAILED: {type(ex).__name__}: {ex}")
        traceback.print_exc(file=sys.stderr)
        return {"error": f"{type(ex).__name__}: {ex}"}


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def host_cores():
    """threads we may actually use: the affinity mask, capped at the GPU box's per-GPU CPU share"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def git_sha():
    try:
        return subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        return None


PMC_KEYS = {128: "gemm_kernel<128, 128, 2, 4, 2", 256: "gemm256_kernel<"}       # prefixes: launch-weighted mean over the instances


def traffic_from_profile(tile, launches_in_run):
    """HBM-side bytes per launch of the dominant GEMM from the committed rocprofv3 --pmc passes of this same command (separate
    FETCH_SIZE / WRITE_SIZE runs, reduced on the GPU box by tools/pmc_reduce.py: 2 x FETCH_SIZE KiB + WRITE_SIZE KiB, the gfx950
    correction of MI355X_MICROARCH.md).  PMC counters need the profiler, so this is never measured inside the run itself: it is
    emitted with its provenance.  FETCH_SIZE counts what the 8 per-XCD L2s request from the fabric, so operands every XCD reads
    (the weight panel) count 8 times although the Infinity Cache serves them."""
    pj = next((q for q in (os.path.join(ROOT, "profiles", r, "pmc_traffic.json") for r in ("r04", "r03")) if os.path.isfile(q)), "")
    if pj:
        j = json.load(open(pj))
        cand = [v for n, v in j.get("kernels", {}).items() if tile in PMC_KEYS and n.startswith(PMC_KEYS[tile]) and v.get("launches")]
        k = None
        if cand:
            n_l = sum(v["launches"] for v in cand)
            k = {"launches": n_l, "traffic_bytes_per_launch": sum(v["traffic_bytes_per_launch"] * v["launches"] for v in cand) / n_l}
        if k:
            return {"bytes_per_launch": k.get("traffic_bytes_per_launch"), "source": os.path.relpath(pj, ROOT),
                    "profile_launches": k.get("launches"), "profile_git_sha": j.get("git_sha"), "run_launches": launches_in_run,
                    "note": "separate rocprofv3 --pmc passes of `bench.py --steps 1 --warmup 1` (FETCH_SIZE x2 per the gfx950 correction, WRITE_SIZE), averaged over "
                            "every launch of this kernel in the step; not measured in this run"}
    return None


class PowerSampler:
    """Package power and shader clock of the GPU this process runs on, read from the amdgpu hwmon files (power1_input in uW, freq1_input
    in Hz, power1_cap) by a thread every 20 ms while a leg runs.  A measurement aid: if the files are not there (or the card cannot be
    matched by PCI address) the sampler reports nothing and the bench line is unaffected.  No subprocess, no GPU call."""

    def __init__(self, device_index=0):
        self.dir, self.samples, self._stop, self._thr = None, [], False, None
        try:
            import glob
            pr = torch.cuda.get_device_properties(device_index)
            want = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}"
            for hw in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
                pci = os.path.basename(os.path.realpath(os.path.join(hw, "..", "..")))
                if pci.lower().startswith(want) and os.path.isfile(os.path.join(hw, "power1_input")):
                    self.dir = hw
                    break
        except Exception:
            self.dir = None

    def _read(self, name):
        with open(os.path.join(self.dir, name)) as f:
            return float(f.read().strip())

    def start(self):
        if self.dir is None:
            return self
        import threading
        self.samples, self._stop = [], False

        def loop():
            while not self._stop:
                try:
                    self.samples.append((self._read("power1_input") * 1e-6, self._read("freq1_input") * 1e-6))
                except Exception:
                    break
                time.sleep(0.02)
        self._thr = threading.Thread(target=loop, daemon=True)
        self._thr.start()
        return self

    def stop(self):
        if self._thr is None:
            return None
        self._stop = True
        self._thr.join(timeout=1.0)
        self._thr = None
        if len(self.samples) < 3:
            return None
        pw, ck = [a for a, _ in self.samples], [b for _, b in self.samples]
        try:
            cap = self._read("power1_cap") * 1e-6
        except Exception:
            cap = None
        return {"package_w_mean": round(sum(pw) / len(pw), 1), "package_w_max": round(max(pw), 1), "power_cap_w": cap,
                "frac_of_cap": round(sum(pw) / len(pw) / cap, 3) if cap else None,
                "sclk_mhz_mean": round(sum(ck) / len(ck), 0), "sclk_mhz_min": round(min(ck), 0), "sclk_mhz_max": round(max(ck), 0), "samples": len(pw),
                "source": "amdgpu hwmon (power1_input, freq1_input, power1_cap), sampled every 20 ms over the timed region"}


# ------------------------------------------------------------------------------------------------------------------ legs
def fixed_cadence_leg(model, slam_cls, cdist, dev, wb, steps, warmup, world=1, rank=0, dist_on=False, emu=0, probe_steps=0, barrier=None,
                      seq_windows=0, frame_source="host"):
    """buffered fixed-cadence schedule (kf_every=10) through the pipelined ShardedTracker; returns dict of results.  seq_windows > 0:
    the stream is a succession of sequences of that many windows (TrackFrontend.sequence_windows), 0: one endless sequence"""
    total_steps = warmup + steps
    n_kf = 7 + WIN * wb * world * (total_steps + probe_steps) + 2
    config = {"Tracking": {"motion_filter": {"thresh": 0.9, "skip": 5, "skip_blur": False, "kf_every": KF_EVERY},
                           "frontend": {"iteration": 0, "window_batch": 1, "sequence_windows": int(seq_windows)}}}
    # encoder features: only the windows in flight are ever read again in this mode (no loop closure): a ring of two steps
    slam = slam_cls(model, config, (H, W), buffer=n_kf + 8, device=dev, feat_buffer=max(64, 2 * (WIN * wb * world + 1)))
    intr = torch.tensor([600.0 * W / 1200.0, 600.0 * H / 680.0, 599.5 * W / 1200.0, 339.5 * H / 680.0])  # calib/replica.txt scaled
    runner = cdist.ShardedTracker(slam, world, rank, wb=wb, pipelined=os.environ.get("CUT3R_PIPELINE", "1") == "1", force_collective=dist_on)
    runner.emulate_gather = emu > 1
    need = runner.frames_needed(total_steps + probe_steps, KF_EVERY, WIN)
    # one synthetic 640x480 recording of a Replica-shaped sequence (2000 frames), resident in HBM as RAW camera frames and read
    # cyclically; every keyframe is resized to the tracking resolution INSIDE the timed region (ShardedTracker._append)
    # frame_source "host" (default): the recording sits in PINNED HOST memory and every keyframe the step reads crosses the link inside
    # the timed region (asynchronous copy + resize on an upload stream, cut3r_slam_amd.dist.PinnedFrames) -- the reference uploads
    # every frame it touches (hislam2/motion_filter.py:78,91); "host-all": ALL frames cross the link (also the 9 of 10 that a
    # fixed-cadence stream never reads); "resident": the recording is in HBM before the timer starts (rounds 1-3)
    rec = synth_camera_frames((seq_windows if seq_windows > 0 else 40) * WIN * KF_EVERY, dev, seed=0)
    if frame_source in ("host", "host-all"):
        host = rec.cpu()
        del rec
        torch.cuda.empty_cache()
        frames = cdist.PinnedFrames(host, need, dev, upload_every_frame=(frame_source == "host-all"))
    else:
        frames = FrameLoop(rec, need)
    t = 0
    while not slam.keyframes.is_initialized:          # prologue (untimed): the 6-keyframe initialisation window
        f = to_tracking(frames[t:t + 1])
        slam.run(t, f, intr, f, intr)
        t += 1
    torch.cuda.synchronize()
    for _ in range(warmup):
        t = runner.step(frames, t, KF_EVERY, WIN, intr)
    runner.flush()
    barrier()
    for k in runner.stats:
        runner.stats[k] = 0
    up0 = getattr(frames, "bytes_uploaded", 0)
    sampler = PowerSampler(torch.device(dev).index or 0).start() if (world == 1 and emu <= 1) else None
    tic = time.perf_counter()
    for _ in range(steps):
        t = runner.step(frames, t, KF_EVERY, WIN, intr)
    runner.flush()                       # the timed region holds exactly K network passes and K replays
    barrier()
    elapsed = time.perf_counter() - tic
    power = sampler.stop() if sampler is not None else None
    return {"elapsed": elapsed, "slam": slam, "runner": runner, "frames": frames, "t": t, "intr": intr, "power": power,
            "frames_per_step": KF_EVERY * WIN * wb, "health": tracking_health(slam),
            "h2d_bytes_per_step": (getattr(frames, "bytes_uploaded", 0) - up0) / max(1, steps)}


def tracking_health(slam):
    """is the leg's result a valid trajectory?  chained-scale book-keeping of the tracker + finiteness of every tracked pose"""
    st = dict(slam.tracker.scale_stats)
    k = int(slam.tracker.t1)
    poses = slam.keyframes.pose[:k].numpy()
    bad = np.flatnonzero(~np.isfinite(poses).all(axis=1))
    return {"tracked_keyframes": k, "poses_finite": bool(bad.size == 0), "first_nonfinite_keyframe": (int(bad[0]) if bad.size else None),
            "nonfinite_windows": int(st["nonfinite_windows"]), "steady_windows": int(st["windows"]),
            "log_scale_absmax": round(float(st["log_scale_absmax"]), 4), "log_scale_last": round(float(st["log_scale_last"]), 4)}


def overlap_mode_leg(model, slam_cls, dev, n_frames=800, lookahead=16, plain_frames=200, deep_frames=2800, deep_lookahead=280, deep_wb=28):
    """the maintained configs' schedule (kf_every=-1, skip=5, thresh=0.9) through Cut3rSlam.run on a content-driven stream
    (cut3r_slam_amd.synth.slideshow_stream: one keyframe per 10 frames): frame by frame (the reference's loop), with a 16-frame look-ahead
    of the keyframe test (one 6-view window at a time), and -- the mode's throughput form -- with `deep_lookahead` tested frames held back
    and the keyframes they yield tracked `deep_wb` windows at a time (`Cut3rSlam.run_buffered` + `Tracking.frontend.window_batch`:
    bit-identical keyframes, poses, depths and edges, tests/test_e2e_gpu.py)"""
    from cut3r_slam_amd import synth
    intr = torch.tensor([600.0 * W / 1200.0, 600.0 * H / 680.0, 599.5 * W / 1200.0, 339.5 * H / 680.0])
    out = {"config": "kf_every=-1, skip=5, thresh=0.9 (config/scannet_config.yaml:20-25); window_batch=1 unless the entry says otherwise"}
    legs = [("buffered_lookahead", n_frames, 160, {"lookahead": lookahead}, 1), ("frame_by_frame", plain_frames, 160, None, 1)]
    if deep_frames > 0:
        # (the warm-up holds two whole decoder batches: graph captures and workspaces of the batched shapes)
        legs.append(("buffered_lookahead_window_batch", deep_frames, 2 * deep_wb * 50 + 200, {"lookahead": deep_lookahead, "pipeline": True}, deep_wb))
    frames = synth.slideshow_stream(max(w + n for _, n, w, _, _ in legs), H, W, hold=10, seed=0, device=dev)
    for name, n, warm, kw, wb in legs:
        config = {"Tracking": {"motion_filter": {"thresh": 0.9, "skip": 5, "skip_blur": False, "kf_every": -1},
                               "frontend": {"iteration": 0, "window_batch": wb}}}
        slam = slam_cls(model, config, (H, W), buffer=(warm + n) // 10 + 16, device=dev)
        if kw is None:
            for t in range(warm):
                slam.run(t, frames[t:t + 1], intr, frames[t:t + 1], intr)
        else:
            slam.run_buffered(frames[:warm], intr, mark_tail=False, **kw)
        torch.cuda.synchronize()
        k0, w0 = slam.keyframes.counter.value, slam.tracker.t1
        tic = time.perf_counter()
        if kw is None:
            for t in range(warm, warm + n):
                slam.run(t, frames[t:t + 1], intr, frames[t:t + 1], intr)
        else:
            slam.run_buffered(frames[warm:warm + n], intr, t_start=warm, mark_tail=False, **kw)
        torch.cuda.synchronize()
        el = time.perf_counter() - tic
        out[name] = {"frames_per_s": round(n / el, 1), "frames": n, "keyframes": slam.keyframes.counter.value - k0,
                     "tested_frames": n // 5, "windows": (slam.tracker.t1 - w0) // 5, "ms_per_frame": round(1e3 * el / n, 3), "window_batch": wb,
                     "health": tracking_health(slam)}
        if kw is not None:
            out[name]["lookahead_tested_frames"] = kw["lookahead"]
            out[name]["latency_frames"] = kw["lookahead"] * 5
        del slam
    return out


def loop_closure_leg(cfg, slam_cls, dev, iters=2000, n_frames=1000, warm=160, wb=1):
    """BASELINE configs[2]: the tracking loop WITH the loop-closure backend (hislam2/hi2.py:112-121, track_backend.py:527-586), one window
    at a time, Tracking.frontend.iteration = 2000 (config/scannet_config.yaml:33).  A random-weight network recognises no place, so the
    weights are synth.loop_state_dict (pose head damped: every keyframe stays covisible with the early ones) -- the backend then fires by
    itself every other eligible window: detect_loop -> NMS -> 6-view re-tracking -> fused Adam over the submap corrections -> rewrite."""
    from cut3r_slam_amd import synth
    from cut3r_slam_amd.model import Cut3rModel
    sd = synth.loop_state_dict(cfg, seed=0, enc_residual_gain=0.1, depth_relief=0.02)
    model = Cut3rModel(cfg, sd, dev, minimal=True)
    config = {"Tracking": {"motion_filter": {"thresh": 0.9, "skip": 5, "skip_blur": False, "kf_every": KF_EVERY},
                           "frontend": {"iteration": int(iters), "window_batch": int(wb)}}}
    intr = torch.tensor([600.0 * W / 1200.0, 600.0 * H / 680.0, 599.5 * W / 1200.0, 339.5 * H / 680.0])
    warm = max(warm, (2 * 5 * int(wb) + 8) * KF_EVERY)      # two whole decoder batches before the clock starts (graph captures, workspaces)
    frames = synth_frames(warm + n_frames, H, W, dev, seed=0)
    slam = slam_cls(model, config, (H, W), buffer=(warm + n_frames) // KF_EVERY + 16, device=dev)
    closures = []
    real_run = slam.backend.run

    def timed_run(*a, **k):
        torch.cuda.synchronize()
        tic = time.perf_counter()
        out = real_run(*a, **k)
        torch.cuda.synchronize()
        if out[0]:
            closures.append(time.perf_counter() - tic)
        return out
    slam.backend.run = timed_run
    for t in range(warm):
        slam.run(t, frames[t:t + 1], intr, frames[t:t + 1], intr)
    torch.cuda.synchronize()
    n_warm = len(closures)
    w0 = slam.tracker.t1
    tic = time.perf_counter()
    for t in range(warm, warm + n_frames):
        slam.run(t, frames[t:t + 1], intr, frames[t:t + 1], intr)
    torch.cuda.synchronize()
    el = time.perf_counter() - tic
    timed = closures[n_warm:]
    be = slam.backend
    out = {"config": f"kf_every={KF_EVERY}, window_batch={wb}, Tracking.frontend.iteration={iters} (config/scannet_config.yaml:33), {n_frames} frames; weights = "
                     "synth.loop_state_dict (every keyframe covisible with the early ones: the backend fires by itself every other eligible window)",
           "frames_per_s": round(n_frames / el, 1), "ms_per_frame": round(1e3 * el / n_frames, 3), "windows": (slam.tracker.t1 - w0) // 5,
           "closures": len(timed), "ms_per_closure": round(1e3 * sum(timed) / max(1, len(timed)), 2),
           "ms_per_closure_max": round(1e3 * max(timed), 2) if timed else None,
           "submaps_at_last_closure": (be.closed_loop["idx_current"][-1] // 5 + 1) if be.closed_loop["idx_current"] else 0,
           "closed_loops": [[int(c), int(m)] for c, m in zip(be.closed_loop["idx_current"], be.closed_loop["idx_matched"])][-6:],
           "health": tracking_health(slam)}
    del slam, model
    torch.cuda.empty_cache()
    return out


def trajectory_parity_leg(dev, production_model=None, production_sd=None):
    """GPU path vs CPU restatement of the reference loop on the same seeded stream + weights: ATE-RMSE with Sim(3) alignment, keyframe
    agreement, edge lists.  Medium config at 64x96 in both keyframe modes (tests/test_e2e_gpu.py asserts the same) and -- a third entry
    -- the PRODUCTION network at 384x512 over two tracking windows + the closing window (tests/test_e2e_production_gpu.py asserts it over
    three; bounded here to about a minute of CPU oracle per precision)."""
    from cut3r_slam_amd import synth
    from cut3r_slam_amd.eval_ate import ate_rmse
    from cut3r_slam_amd.model import Cut3rModel
    from cut3r_slam_amd.slam import Cut3rSlam
    from oracle import slam_run as SR
    Hm, Wm = 64, 96
    intr_m = np.array([80.0, 80.0, 47.5, 31.5], np.float32)
    cfg = synth.medium_config()
    sd = synth.tracking_state_dict(cfg, 11)
    model = Cut3rModel(cfg, sd, dev, minimal=True)
    res = {"config": "medium (enc 256/3/4, dec 192/4/3 + 4 state heads, DPT head) at 64x96, and the production network at 384x512; "
                     "oracle/slam_run.py fp32 (and TF32-rounded operands) on the host",
           "alignment": "Sim(3) Umeyama, RMSE of translation residuals (evo_ape tum -vas, scripts/run_scannet.py:34-36)"}
    legs = [("fixed_cadence_kf_every_2", cfg, sd, model, {"thresh": 0.9, "skip": 1, "kf_every": 2}, synth.pan_stream(70, Hm, Wm, 5, 2, 1, 0), intr_m),
            ("overlap_mode_skip_2", cfg, sd, model, {"thresh": 0.9, "skip": 2, "kf_every": -1}, synth.slideshow_stream(150, Hm, Wm, 4, 3), intr_m)]
    if production_model is not None:
        intr_p = np.array([600.0 * W / 1200.0, 600.0 * H / 680.0, 599.5 * W / 1200.0, 339.5 * H / 680.0], np.float32)
        legs.append(("production_384x512_kf_every_2", production_model.cfg, production_sd, production_model, {"thresh": 0.9, "skip": 1, "kf_every": 2},
                     synth.pan_stream(23, H, W, 9, 6, 1, 0), intr_p))
    for tag, cfg_l, sd_l, model_l, mf, frames, intr in legs:
        tic = time.perf_counter()
        so = SR.run_stream(cfg_l, sd_l, frames, intr, mf, precision="fp32")
        # (the TF32 leg of the production entry is asserted by tests/test_e2e_production_gpu.py: 0.19 mm HIP vs 0.21 mm TF32 over three
        #  windows, profiles/r03/achieved_errors.txt; here it would add another minute of CPU)
        sotf = SR.run_stream(cfg_l, sd_l, frames, intr, mf, precision="tf32") if model_l is not production_model else None
        t_oracle = time.perf_counter() - tic
        conf = {"Tracking": {"motion_filter": dict(mf), "frontend": {"iteration": 0}}}
        Hl, Wl = frames.shape[2:]
        slam = Cut3rSlam(model_l, conf, (Hl, Wl), buffer=frames.shape[0] + 8, device=dev)
        fr = frames.to(dev)
        n = fr.shape[0]
        it = torch.from_numpy(intr)
        for t in range(n):
            slam.run(t, fr[t:t + 1], it, fr[t:t + 1], it, second_last_frame=(t == n - 2), last_frame=(t == n - 1))
        torch.cuda.synchronize()
        ts, poses = slam.trajectory()
        tg = np.concatenate([ts.reshape(-1, 1).astype(np.float64), poses.astype(np.float64)], 1)
        tr, ttf = so.trajectory(), (sotf.trajectory() if sotf is not None else None)
        kf_gpu, kf_ref = set(tg[:, 0].tolist()), set(tr[:, 0].tolist())
        ii, jj, _ = slam.graph.edges_numpy()
        e_gpu, e_ref = list(zip(ii.tolist(), jj.tolist())), list(zip(so.graph.ii, so.graph.jj))
        first_div = next((k for k, (a, b) in enumerate(zip(e_gpu, e_ref)) if a != b), None)
        if first_div is None and len(e_gpu) != len(e_ref):
            first_div = min(len(e_gpu), len(e_ref))
        diff = sorted(set(e_gpu) ^ set(e_ref))
        a = ate_rmse(tg, tr, 0.01, True)
        atf = ate_rmse(ttf, tr, 0.01, True) if ttf is not None else {"rmse": None}
        path = float(np.linalg.norm(np.diff(tr[:, 1:4], axis=0), axis=1).sum())
        res[tag] = {"keyframes": len(tr), "windows": len(so.windows), "path_length_m": round(path, 4),
                    "ate_rmse_m": a["rmse"], "ate_max_m": a["max"], "sim3_scale": a["scale"], "ate_rmse_mm_per_m": 1e3 * a["rmse"] / max(path, 1e-9),
                    "ate_rmse_m_cpu_tf32_vs_cpu_fp32": atf["rmse"],
                    "keyframe_agreement": len(kf_gpu & kf_ref) / max(1, len(kf_gpu | kf_ref)),
                    "edges_gpu": len(e_gpu), "edges_cpu": len(e_ref), "edge_lists_equal": e_gpu == e_ref, "first_divergent_edge": first_div,
                    "differing_edges": [[int(i), int(j), so.graph.ratios.get((max(i, j), min(i, j)))] for i, j in diff[:8]],
                    "cpu_oracle_s": round(t_oracle, 1)}
        del slam
    if production_model is not None:
        # fourth entry: against the REFERENCE'S OWN loop at production shape -- tests/golden/loop_production.npz holds the keyframe trajectory
        # that the reference's kfFilter + TrackFrontend.run produced on the CPU with its ViT-L / DPT model for these weights and this stream
        # (generated in the build container by tests/golden/make_fixtures.py; the reference itself does not travel)
        fpath = os.path.join(ROOT, "tests", "golden", "loop_production.npz")
        if os.path.exists(fpath):
            f = np.load(fpath)
            frames = synth.pan_stream(33, H, W, 9, 6, 1, 0)
            if int(frames.long().sum()) == int(f["frames_sum"]):
                conf = {"Tracking": {"motion_filter": {"thresh": 0.9, "skip": 1, "kf_every": 2}, "frontend": {"iteration": 0}}}
                # (the fixture's weights, not the bench's: synth.tracking_state_dict(production_config(), 0, enc_residual_gain=0.1))
                model_f = Cut3rModel(production_model.cfg, synth.tracking_state_dict(production_model.cfg, 0, enc_residual_gain=0.1), dev, minimal=True)
                slam = Cut3rSlam(model_f, conf, (H, W), buffer=41, device=dev)
                fr, it, n = frames.to(dev), torch.from_numpy(f["intrinsic"]), frames.shape[0]
                for t in range(n):
                    slam.run(t, fr[t:t + 1], it, fr[t:t + 1], it, second_last_frame=(t == n - 2), last_frame=(t == n - 1))
                torch.cuda.synchronize()
                ts, poses = slam.trajectory()
                tg = np.concatenate([ts.reshape(-1, 1).astype(np.float64), poses.astype(np.float64)], 1)
                k = int(f["t1"])
                tr = np.concatenate([f["keyframes"][:k].reshape(-1, 1).astype(np.float64), f["pose"].astype(np.float64)], 1)
                a = ate_rmse(tg, tr, 0.01, True)
                ii, jj, _ = slam.graph.edges_numpy()
                path = float(np.linalg.norm(np.diff(tr[:, 1:4], axis=0), axis=1).sum())
                res["production_384x512_vs_reference_loop"] = {
                    "reference": "hislam2 MotionFilter.kfFilter + TrackFrontend.run with the reference's ARCroco3DStereo on the CPU (tests/golden/loop_production.npz)",
                    "keyframes": len(tr), "keyframe_agreement": float(np.array_equal(tg[:, 0], tr[:, 0])), "path_length_m": round(path, 4),
                    "ate_rmse_m": a["rmse"], "ate_max_m": a["max"], "sim3_scale": a["scale"], "ate_rmse_mm_per_m": 1e3 * a["rmse"] / max(path, 1e-9),
                    "edge_lists_equal": bool(np.array_equal(ii, f["ii"]) and np.array_equal(jj, f["jj"])), "edges_gpu": int(len(ii)), "edges_reference": int(len(f["ii"]))}
                del slam, model_f
    return res


def cpu_baseline(cfg, sd, imgs_u8, frames_per_window=50):
    """Oracle (kind 'port') timed on the host cores on a BOUNDED sample of the tracking loop at 384x512: one keyframe-filter
    encode + patch-overlap test, one 2-view window through the network (the model cost is linear in views: extrapolated to
    5 encodes + 6 views), the window alignment of 6 views and 50 covisibility-graph updates at the real map sizes."""
    from oracle import cut3r_oracle as O
    from oracle import slam_oracle as SO
    from oracle import slam_run as SR
    cores = host_cores()
    torch.set_num_threads(cores)
    x = O.normalize(imgs_u8)
    with torch.no_grad():
        t0 = time.perf_counter()
        f0, _ = O.encode_image(cfg, sd, x[:1])
        t_enc = time.perf_counter() - t0
        t0 = time.perf_counter()
        SR.patch_overlap_ratio_f32(f0[0], f0[0].roll(1, 0))
        t_ovl = time.perf_counter() - t0
        t0 = time.perf_counter()
        preds = O.forward_views(cfg, sd, x[:2], minimal=True)
        t_win2 = time.perf_counter() - t0
    t_dec_view = max(t_win2 - 2 * t_enc, 0.0) / 2
    # chaining of a 6-view window + 50 graph updates on maps of the real size (network outputs of the sample, repeated)
    pts = torch.cat([preds[i % 2]["pts3d_in_self_view"] for i in range(6)], 0).abs() + 0.5
    conf = torch.cat([preds[i % 2]["conf_self"] for i in range(6)], 0)
    enc = torch.cat([preds[i % 2]["camera_pose"] for i in range(6)], 0)
    nkf = 64
    st = {"pose": torch.zeros(nkf, 7), "depth": torch.ones(nkf, H, W), "submap_ds": torch.ones(nkf // 5 + 1, 6, H // 2, W // 2, 3),
          "conf_ds": torch.zeros(nkf // 5 + 1, 6, H // 2, W // 2)}
    st["pose"][:, 6] = 1
    t0 = time.perf_counter()
    full = SO.track_window(st, 0, 6, pts, conf, enc, True)
    t_align = time.perf_counter() - t0
    graph = SR.RefGraph()
    g = np.random.default_rng(0)
    c2w = np.tile(np.eye(4, dtype=np.float32), (60, 1, 1))
    c2w[:, :3, 3] = g.normal(0, 0.7, (60, 3)).astype(np.float32)
    pm_all = np.ascontiguousarray(np.broadcast_to(full[0][2][::2, ::2].numpy()[None], (60, H // 2, W // 2, 3)))
    cur_pm = full[1][2].numpy()
    K4 = np.array([256.0, 203.3, 255.8, 191.7], np.float32)
    t0 = time.perf_counter()
    for i in range(10, 60):
        graph.add(i, c2w[:i], pm_all[:i], c2w[i], cur_pm, K4)
    t_graph = (time.perf_counter() - t0) / 50
    window_s = 5 * (t_enc + t_ovl) + 6 * t_dec_view + t_align + 5 * t_graph
    return {"value": round(frames_per_window / window_s, 3), "unit": "frames/s", "cores": cores, "kind": "port",
            "stage_ms": {"kf_filter_encode": round(1e3 * t_enc, 1), "patch_overlap": round(1e3 * t_ovl, 1), "decoder_heads_per_view": round(1e3 * t_dec_view, 1),
                         "window_alignment_6_views": round(1e3 * t_align, 1), "graph_update_per_keyframe_avg_35_prev": round(1e3 * t_graph, 1)},
            "sample": f"oracle fp32 at 384x512: 1 encode_image ({t_enc:.2f} s) + patch-overlap test + one 2-view window ({t_win2:.2f} s) + "
                      f"alignment of 6 views + 50 graph updates (10..59 previous keyframes); composed into a 50-frame window = 5 (encode + test) + "
                      "6 decoder/head views + 1 alignment + 5 graph updates"}


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N fresh ranks (one per GPU) the way the driver's own multi-GPU command does
    -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <same flags>` --
    as a CHILD process (this process has made no GPU call and makes none), relay rank 0's JSON line on stdout, everything else on stderr,
    and return the children's exit code."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // n)))
    log(f"starting {n} ranks: {' '.join(cmd)}")
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = []
    for line in proc.stdout:
        (lines.append if line.lstrip().startswith("{") else sys.stderr.write)(line)
    rc = proc.wait()
    if rc == 0 and not lines:
        log("the ranks exited without a result line")
        return 1
    for line in lines[-1:]:
        sys.stdout.write(line)
    sys.stdout.flush()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-operating-points", action="store_true")
    ap.add_argument("--no-trajectory-parity", action="store_true")
    ap.add_argument("--no-production-parity", action="store_true", help="skip the 384x512 entry of trajectory_parity (about two minutes of CPU oracle)")
    ap.add_argument("--small", action="store_true", help="debug: tiny network (NOT a valid benchmark line)")
    ap.add_argument("--sequence-windows", type=int, default=40, help="windows per sequence (40 = a Replica-shaped 2000-frame sequence at "
                    "kf_every=10, whatever the number of GPUs: a bigger job runs through more sequences per step); 0 = one endless stream")
    ap.add_argument("--sequence-per-gpu", action="store_true", help="multiply --sequence-windows by the number of GPUs (sequences that grow "
                    "with the job: the covisibility test of a keyframe then grows with it too)")
    ap.add_argument("--frames", choices=("host", "host-all", "resident"), default="host", help="where the camera recording lives: pinned host "
                    "memory with every keyframe uploaded inside the timed region (default), every FRAME uploaded, or resident in HBM")
    ap.add_argument("--window-batch", type=int, default=28, help="tracking windows pushed through the decoder together "
                    "(buffered-stream throughput mode; 1 = the reference's one-window-at-a-time schedule)")
    args = ap.parse_args()

    # ---- `--gpus N` starts the N ranks itself (one process per GPU) when no launcher has done so: fresh child processes through
    #      torch.distributed.run, started BEFORE anything in this process touches the GPU; this process only relays rank 0's JSON line and
    #      the exit code.  Under a launcher (WORLD_SIZE set) --gpus must agree with it: a mislabelled run is refused, never reported.
    if args.gpus < 1:
        raise SystemExit("bench: --gpus must be >= 1")
    backend_env = os.environ.get("CUT3R_DIST_BACKEND", "nccl")
    emu_env = int(os.environ.get("CUT3R_EMULATE_WORLD", "0"))
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            if backend_env == "nccl" and torch.cuda.device_count() < args.gpus:       # device_count() does not initialise the GPU
                raise SystemExit(f"bench: --gpus {args.gpus} over RCCL needs {args.gpus} visible GPUs, found {torch.cuda.device_count()} "
                                 "(CUT3R_DIST_BACKEND=gloo runs a functional multi-rank job on one GPU)")
            raise SystemExit(launch_ranks(args.gpus))
    elif int(os.environ["WORLD_SIZE"]) != args.gpus and emu_env <= 1:
        raise SystemExit(f"bench: --gpus {args.gpus} disagrees with WORLD_SIZE={os.environ['WORLD_SIZE']} of the launcher: pass the same number "
                         "(`python bench.py --gpus N` starts the N ranks itself)")
    elif backend_env == "nccl" and torch.cuda.device_count() < int(os.environ.get("LOCAL_WORLD_SIZE", os.environ["WORLD_SIZE"])):
        raise SystemExit(f"bench: {os.environ['WORLD_SIZE']} RCCL ranks need one GPU each, found {torch.cuda.device_count()}")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    emu = int(os.environ.get("CUT3R_EMULATE_WORLD", "0"))     # debug: rank 0 of an `emu`-GPU job on ONE GPU (replay load only)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("CUT3R_DIST_BACKEND", "nccl") != "nccl":
        local_rank = 0                 # functional run: all ranks share GPU 0
    dist_on = world > 1 or os.environ.get("CUT3R_FORCE_DIST") == "1"     # the env flag rehearses the RCCL path with one rank
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    dist = None
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
    from cut3r_slam_amd import synth
    from cut3r_slam_amd import dist as cdist

    WB = max(1, args.window_batch)
    cfg = tiny_config("dpt") if args.small else production_config()
    t0 = time.time()
    torch.set_num_threads(host_cores())
    # random init through the reference key schema; the encoder's residual branches are damped so that patch features stay
    # content dependent (the overlap-mode keyframe test needs that; cut3r_slam_amd/synth.py) -- no effect on the arithmetic
    # depth_relief: see synth.tracking_state_dict (keeps the chained scale of an uncut synthetic stream in range)
    sd = synth.tracking_state_dict(cfg, seed=0, enc_residual_gain=0.1, depth_relief=float(os.environ.get("CUT3R_DEPTH_RELIEF", "0.02")))
    log(f"weights synthesised in {time.time() - t0:.1f}s")
    model = Cut3rModel(cfg, sd, dev, minimal=True)
    torch.cuda.synchronize()
    t_build = time.time() - t0
    log(f"model resident in HBM after {t_build:.1f}s")

    def barrier():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    if emu > 1:
        world = emu                                # after the process-group decisions above: no collective is created
    single = rank == 0 and world == 1 and emu <= 1 and not dist_on
    probe_steps = args.steps if (single and not args.no_roofline) else 0
    log("fixed-cadence leg: initialisation window, warmup, timed region")
    # weak scaling: every GPU adds 28 windows per step of the same Replica-shaped sequences (per-GPU work fixed: a keyframe is tested
    # against the <= 200 keyframes of its own sequence); --sequence-per-gpu makes the sequences grow with the job instead
    SEQ = max(0, args.sequence_windows) * (world if args.sequence_per_gpu else 1)
    emu_rank = int(os.environ.get("CUT3R_EMULATE_RANK", "0")) % max(1, world) if emu > 1 else rank    # debug: rehearse another rank's load
    leg = fixed_cadence_leg(model, Cut3rSlam, cdist, dev, WB, args.steps, args.warmup, world, emu_rank, dist_on, emu, probe_steps, barrier,
                            seq_windows=SEQ, frame_source=args.frames)
    elapsed, slam, runner, frames, t, intr = (leg[k] for k in ("elapsed", "slam", "runner", "frames", "t", "intr"))
    frames_per_step = leg["frames_per_step"]
    leg_h2d = leg["h2d_bytes_per_step"]
    leg_power = leg.get("power")
    if leg_power:                              # energy of the timed region: mean package power x time
        leg_power["joules_per_step"] = round(leg_power["package_w_mean"] * elapsed / max(1, args.steps), 1)
        leg_power["joules_per_frame"] = round(leg_power["package_w_mean"] * elapsed / max(1, args.steps * frames_per_step), 4)
    health = leg["health"]
    if not health["poses_finite"] or health["nonfinite_windows"]:
        raise SystemExit(f"bench: the headline leg produced non-finite poses: {health}")      # never report a throughput for it
    if dist_on:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    hbm_peak = torch.cuda.max_memory_allocated()            # weights + graphs' static buffers + activations + this leg's stores
    frames_total = frames_per_step * args.steps * world
    value = frames_total / elapsed
    log(f"timed region done: {elapsed:.3f}s for {frames_total} frames -> {value:.1f} frames/s")
    from cut3r_slam_amd.track_frontend import TIMING
    log(f"replay round trips per window [ms]: log-depth {1e3 * TIMING['sync1_s'] / max(1, TIMING['windows']):.2f}, "
        f"window update {1e3 * TIMING['sync2_s'] / max(1, TIMING['windows']):.2f}")
    log("host wall-clock per step [ms]: " + ", ".join(f"{k[:-2]} {1e3 * v / max(1, runner.stats['steps']):.2f}" for k, v in runner.stats.items() if k != "steps"))

    dump = os.environ.get("CUT3R_DUMP_STATE")
    if dump:                               # tests: the replicated result of every rank
        k = slam.tracker.t1
        ii, jj = slam.graph.edges_absolute()
        # (depth of the last keyframe: written by the owner of the last window only until the next window rewrites it)
        np.savez(f"{dump}.rank{rank}.npz", pose=slam.keyframes.pose[:k].numpy(), depth_sum=slam.keyframes.depth[:k - 1].double().sum(dim=(1, 2)).cpu().numpy(),
                 submap_sum=slam.keyframes.submap_ds[:(k - 1) // 5].double().sum(dim=(2, 3, 4)).cpu().numpy(),
                 w2c=slam.keyframes.w2c[:k].cpu().numpy(), ii=ii, jj=jj, k=k)

    roofline, cpu_base, op_points, traj = None, None, None, None
    if probe_steps:
        # second, instrumented pass over the same number of steps: HIP events around every launch of the large-tile GEMM
        # and attention kernels on the launch stream
        need = frames_per_step * args.steps
        if t + need + 1 <= frames.shape[0]:
            probe = KernelProbe()
            probe.install()
            model.use_graphs = False      # events must bracket live launches, not a graph replay
            # ... and launches that run ALONE: without the encoder look-ahead stream and the replay side stream (round 4: with the frames
            # arriving on an upload stream the look-ahead pass overlapped the eager decoder, and every bracketed launch read 1.5 x its
            # duration -- the rates of time-shared kernels, not of the kernels)
            ahead_was, pipe_was = runner.encode_ahead, runner.pipelined
            runner.flush()
            torch.cuda.synchronize()      # (a look-ahead pass already issued has stored its features: the windows find them)
            runner.encode_ahead, runner.pipelined, runner._ahead = False, False, None
            for _ in range(args.steps):
                t = runner.step(frames, t, KF_EVERY, WIN, intr)
            runner.flush()
            res = probe.result()
            probe.remove()
            runner.encode_ahead, runner.pipelined = ahead_was, pipe_was
            model.use_graphs = True
            gem = {k[1]: v for k, v in res.items() if k[0] == "gemm" and v[0]}
            if gem:
                dom = max(gem, key=lambda tk: gem[tk][1])          # the GEMM kernel with the largest total time in this workload
                n, ms, fl, by = gem[dom]
                ach = fl / (ms * 1e-3) / 1e12
                tfp = traffic_from_profile(dom, n)
                tot_ms = sum(v[1] for v in gem.values())
                others = []
                for tk, (n2, ms2, fl2, by2) in gem.items():
                    if tk != dom:
                        others.append({"kernel": KernelProbe.GEMM[tk], "bound": "mfma", "launches": n2, "avg_launch_us": round(ms2 * 1e3 / n2, 2),
                                       "achieved": round(fl2 / (ms2 * 1e-3) / 1e12, 2), "peak": PEAK_F16, "unit": "TFLOP/s",
                                       "frac": round(fl2 / (ms2 * 1e-3) / 1e12 / PEAK_F16, 4), "share_of_large_gemm_time": round(ms2 / tot_ms, 3)})
                for k, (n2, ms2, fl2, by2) in sorted((k, v) for k, v in res.items() if k[0] == "attn" and v[0]):
                    others.append({"kernel": (f"attn_pipe_kernel<{k[1]},1,4,4>" if k[1] in (48, 64) else f"attn_kernel<{k[1]},4>")
                                             + f" ({'encoder self-attention [B,16,768,64]' if k[2] == 'enc' else 'decoder self/cross attention'})",
                                   "bound": "mfma", "launches": n2, "avg_launch_us": round(ms2 * 1e3 / n2, 2), "achieved": round(fl2 / (ms2 * 1e-3) / 1e12, 2),
                                   "peak": PEAK_F16, "unit": "TFLOP/s", "frac": round(fl2 / (ms2 * 1e-3) / 1e12 / PEAK_F16, 4), "flops_per_launch": fl2 / n2})
                roofline = {"bound": "mfma", "kernel": KernelProbe.GEMM[dom], "achieved": round(ach, 2), "peak": PEAK_F16, "unit": "TFLOP/s",
                            "frac": round(ach / PEAK_F16, 4), "traffic": (tfp or {}).get("bytes_per_launch"), "traffic_from_profile": tfp,
                            "launches": n, "avg_launch_us": round(ms * 1e3 / n, 2), "flops_per_launch": fl / n,
                            "algorithmic_bytes_per_launch": by / n, "share_of_large_gemm_time": round(ms / tot_ms, 3),
                            "second_kernel": others[0] if others else None, "other_kernels": others, "gemm_shapes": shape_entries(res)}
                try:
                    mp = mfma_probe(dev)
                    sus = mp["random_operands"]["tflops"]
                    roofline["sustained_mfma"] = dict(mp, note="bare MFMA loop of this device (no LDS, no memory traffic), N(0,1) vs all-zero fp16 operands: "
                                                                "the 2.5 PFLOP/s `peak` assumes 2.4 GHz, which the chip does not hold while its operands toggle",
                                                      frac_of_sustained=round(ach / sus, 4) if sus > 0 else None)
                    if sus > 0:
                        for e in others:                        # the same ratio for the other MFMA kernels of the table
                            if e.get("bound") == "mfma" and e.get("achieved") is not None:
                                e["frac_of_sustained"] = round(e["achieved"] / sus, 4)
                except Exception as ex:      # a measurement aid: never fails the bench line
                    roofline["sustained_mfma"] = {"error": str(ex)}
    if single and not args.small and not args.no_operating_points:
        log("operating points: fixed cadence with one window at a time")
        del leg, runner, slam
        op_points = {}
        l1 = fixed_cadence_leg(model, Cut3rSlam, cdist, dev, 1, 16, 4, barrier=barrier, probe_steps=4)
        op_points["fixed_cadence_window_batch_1"] = {
            "config": "kf_every=10, window_batch=1: the reference's one-window-at-a-time schedule (50 frames of buffering)",
            "frames_per_s": round(16 * l1["frames_per_step"] / l1["elapsed"], 1), "ms_per_window": round(1e3 * l1["elapsed"] / 16, 3), "windows": 16,
            "health": l1["health"], "power": l1.get("power")}
        if not args.no_roofline:
            # the M = 769 / 768 shapes of ONE window (decoder GEMMs of 4 x 7 row tiles of 128^2, encoder batch 5): HIP events around
            # every GEMM / attention launch of four eager windows
            pr = KernelProbe(all_tiles=True)
            pr.install()
            model.use_graphs = False
            tt = l1["t"]
            for _ in range(4):
                tt = l1["runner"].step(l1["frames"], tt, KF_EVERY, WIN, l1["intr"])
            l1["runner"].flush()
            r1 = pr.result()
            pr.remove()
            model.use_graphs = True
            ent = probe_entries(r1)
            tot_fl, tot_ms = sum(v[2] for k, v in r1.items() if k[0] != "shape"), sum(v[1] for k, v in r1.items() if k[0] != "shape")
            op_points["fixed_cadence_window_batch_1"]["roofline"] = {
                "note": "eager pass, HIP events on the launch stream: per-kernel rates of the one-window shapes (M = 769 / 768 rows; in the timed graph "
                        "replay the two decoder streams and the DPT head of the previous view run side by side)",
                "mfma_frac_of_probed_kernel_time": round(tot_fl / (tot_ms * 1e-3) / 1e12 / PEAK_F16, 4) if tot_ms > 0 else None,
                "kernels": ent, "gemm_shapes": shape_entries(r1, 8)}
        del l1
        if SEQ > 0:
            log("operating points: the same schedule over ONE endless stream (covisibility tests against every earlier keyframe)")
            le = fixed_cadence_leg(model, Cut3rSlam, cdist, dev, WB, args.steps, args.warmup, barrier=barrier, seq_windows=0)
            op_points["endless_stream"] = {
                "config": f"kf_every=10, window_batch={WB}, no sequence cuts: one stream of {(args.steps + args.warmup) * le['frames_per_step']} frames; every new keyframe "
                          "is tested against ALL earlier ones (factor_graph.py:148-197), so the step time grows with the stream",
                "frames_per_s": round(args.steps * le["frames_per_step"] / le["elapsed"], 1), "ms_per_step": round(1e3 * le["elapsed"] / args.steps, 3),
                "keyframes_at_end": int(le["slam"].tracker.t1), "valid": bool(le["health"]["poses_finite"] and not le["health"]["nonfinite_windows"]),
                "health": le["health"]}
            del le
        log("operating points: the headline schedule with EVERY frame uploaded from pinned host memory")
        la = fixed_cadence_leg(model, Cut3rSlam, cdist, dev, WB, max(2, args.steps // 2), args.warmup, barrier=barrier, seq_windows=SEQ, frame_source="host-all")
        ns = max(2, args.steps // 2)
        op_points["every_frame_uploaded"] = {
            "config": f"the headline schedule (kf_every=10, window_batch={WB}) with ALL {la['frames_per_step']} frames of a step crossing the host link (pinned memory, "
                      "asynchronous copies on the upload stream), as the reference's motion filter uploads every frame it is handed (hislam2/motion_filter.py:78); "
                      "the 9 of 10 frames a fixed-cadence stream never reads are dropped on arrival",
            "frames_per_s": round(ns * la["frames_per_step"] / la["elapsed"], 1), "ms_per_step": round(1e3 * la["elapsed"] / ns, 3),
            "h2d_mb_per_step": round(la["h2d_bytes_per_step"] / 1e6, 1), "h2d_gb_per_s": round(la["h2d_bytes_per_step"] * ns / la["elapsed"] / 1e9, 2),
            "health": la["health"]}
        del la
        log("operating points: overlap mode (kf_every=-1, skip=5, thresh=0.9)")
        op_points["overlap_mode"] = guarded("overlap mode leg", overlap_mode_leg, model, Cut3rSlam, dev)
        log("operating points: loop closure on (BASELINE configs[2])")
        op_points["loop_closure_on"] = guarded("loop closure leg", loop_closure_leg, cfg, Cut3rSlam, dev)
        # the same loop with the windows decoded four at a time: the backend takes its turn after every window of a batch (identical
        # closures and stores, tests/test_e2e_gpu.py), latency 200 frames
        op_points["loop_closure_on_window_batch_4"] = guarded("loop closure leg, window batch 4", loop_closure_leg, cfg, Cut3rSlam, dev, wb=4)
        log("operating points: GS mapper on a synthetic window (rasteriser forward + backward, pose refinement, mapping)")
        op_points["gs_mapper_synthetic_window"] = guarded("GS mapper leg", synth.gs_mapper_window_leg, H, W, dev)
    if single and not args.no_trajectory_parity:
        log("trajectory parity leg (medium config, GPU vs CPU oracle)")
        traj = guarded("trajectory parity leg", trajectory_parity_leg, dev, None if (args.small or args.no_production_parity) else model, sd)
    if single and not args.no_cpu_baseline and not args.small:
        log("cpu baseline (oracle on host cores)")
        cpu_base = guarded("cpu baseline", cpu_baseline, cfg, sd, torch.cat([to_tracking(frames[i:i + 1]) for i in range(2)], 0).cpu())
        log("cpu baseline done")

    if rank == 0:
        out = {
            "metric": "frames/sec (ViT pointmap + covisibility-graph tracking step) on 640x480" + (f" [DEBUG: rank {emu_rank} of an emulated {emu}-GPU job]" if emu > 1 else ""), "value": round(value, 2),
            "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "keyframes_per_s": round(value / KF_EVERY, 2),
            "nonfinite_windows": health["nonfinite_windows"], "health": health, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": "BUFFERED FIXED-CADENCE mode: Replica-shaped 640x480 sequences ("
                                   + {"host": "raw u8 frames streamed from pinned host memory: every keyframe the step reads is uploaded (async H2D on an upload stream) ",
                                      "host-all": "raw u8 frames streamed from pinned host memory: EVERY frame is uploaded (async H2D on an upload stream), keyframes ",
                                      "resident": "raw u8 frames resident in HBM; every keyframe "}[args.frames]
                                   + "and resized to the 384x512 tracking resolution inside the timed region, cv2.resize INTER_LINEAR semantics); kf_every=10; "
                                   + (f"a sequence = {SEQ} windows = {SEQ * 50} frames (Replica room0: 2000 frames, x{world} GPUs), sequences follow each other without "
                                      "a gap (the last keyframe of one is keyframe 0 of the next: initialisation window, empty graph); " if SEQ > 0 else "ONE endless stream; ")
                                   + f"step = {WB} window(s) "
                                   f"= {frames_per_step} frames ({frames_per_step} frames of buffering): each new keyframe through the ViT-L encoder once (batched), "
                                   "6-view recurrent decoder + DPT head per window (windows batched through the decoder), "
                                   "log-depth/pose chaining + covisibility-graph update per keyframe; "
                                   "ViT-L/24 enc, 768/12 dual decoder, DPT head, random init; GS backend off; the reference's own "
                                   "schedules (one window at a time; overlap mode) are in `operating_points`"
                                   + (" [DEBUG --small]" if args.small else ""),
                       "frames_per_step": frames_per_step, "window_views": 6, "window_batch": WB, "sequence_windows": SEQ, "parallelism": f"window-sharded x{world}",
                       "frame_source": args.frames, "h2d_mb_per_step": round(leg_h2d / 1e6, 2)},
            "power": leg_power, "roofline": roofline, "cpu_baseline": cpu_base, "operating_points": op_points, "trajectory_parity": traj,
            "ate_rmse_m": (traj or {}).get("fixed_cadence_kf_every_2", {}).get("ate_rmse_m"),
            "ate_rmse_m_production_384x512": (traj or {}).get("production_384x512_kf_every_2", {}).get("ate_rmse_m"),
            "ate_rmse_m_production_384x512_vs_reference_loop": (traj or {}).get("production_384x512_vs_reference_loop", {}).get("ate_rmse_m"),
            "hbm_peak_gb": round(hbm_peak / 1e9, 2),
            "memory_plan_8_gpus_25_steps_gb": {k: (round(v / 1e9, 2) if k != "keyframes" else v) for k, v in
                                               cdist.memory_plan(8, 25, WB, H, W, workspace_bytes=max(0, hbm_peak - cdist.memory_plan(1, args.steps + args.warmup + (probe_steps or 0), WB, H, W, weights_bytes=0)["total"])).items()
                                               if k != "fits"},
            "build_s": round(t_build, 1), "git_sha": git_sha(),
        }
        print(json.dumps(out))
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
