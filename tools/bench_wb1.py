#!/usr/bin/env python3
"""ms per tracking window of the one-window schedule (kf_every=10, window_batch=1: bench.py's fixed_cadence_window_batch_1 operating
point), alone -- for A/B runs of the CUT3R_* switches in separate processes.  usage: bench_wb1.py [windows=32]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cut3r_slam_amd import dist as cdist
from cut3r_slam_amd.slam import Cut3rSlam

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
from cut3r_slam_amd import synth
from cut3r_slam_amd.config import production_config
from cut3r_slam_amd.model import Cut3rModel
cfg = production_config()
model = Cut3rModel(cfg, synth.tracking_state_dict(cfg, seed=0, enc_residual_gain=0.1, depth_relief=0.02), dev, minimal=True)
best = None
for rep in range(3):
    l1 = bench.fixed_cadence_leg(model, Cut3rSlam, cdist, dev, 1, n, 4, barrier=lambda: torch.cuda.synchronize())
    ms = 1e3 * l1["elapsed"] / n
    best = ms if best is None else min(best, ms)
    st = l1["runner"].stats
    print(f"rep {rep}: {ms:.3f} ms / window ({n * l1['frames_per_step'] / l1['elapsed']:.0f} frames/s); host wall-clock per window [ms]: "
          + ", ".join(f"{k[:-2]} {1e3 * v / max(1, st['steps']):.2f}" for k, v in st.items() if k != "steps"), flush=True)
    del l1
    torch.cuda.empty_cache()
sw = {k: v for k, v in os.environ.items() if k.startswith("CUT3R_")}
print(f"best {best:.3f} ms / window with {sw}")
