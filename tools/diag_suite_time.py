"""diagnostic (GPU box): where the e2e tests spend their time -- host thread configuration, CPU oracle at several thread counts, HIP run"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "torch threads", torch.get_num_threads(), "interop", torch.get_num_interop_threads())
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    if os.path.exists(p):
        print(p, open(p).read().strip())
print({k: v for k, v in os.environ.items() if "THREADS" in k or k.startswith("OMP") or k.startswith("MKL")})
from cut3r_slam_amd import synth
from oracle import slam_run as SR
H, W = 64, 96
INTR = np.array([80.0, 80.0, 47.5, 31.5], np.float32)
cfg = synth.medium_config(); sd = synth.tracking_state_dict(cfg, 11)
mf = {"thresh": 0.9, "skip": 1, "kf_every": 2}
frames = synth.pan_stream(70, H, W, pool=5, num=2, den=1, seed=0)
for nt in (torch.get_num_threads(), 16, 8, 4, 1):
    torch.set_num_threads(nt)
    t = time.time(); so = SR.run_stream(cfg, sd, frames, INTR, mf, precision="fp32"); print("oracle medium fp32 threads", nt, round(time.time() - t, 1), "s", flush=True)
torch.set_num_threads(8)
t = time.time(); so = SR.run_stream(cfg, sd, frames, INTR, mf, precision="tf32"); print("oracle medium tf32 threads 8", round(time.time() - t, 1), "s", flush=True)
from cut3r_slam_amd.model import Cut3rModel
from cut3r_slam_amd.slam import Cut3rSlam
t = time.time()
model = Cut3rModel(cfg, sd, "cuda:0", minimal=True)
conf = {"Tracking": {"motion_filter": dict(mf), "frontend": {"iteration": 0}}}
slam = Cut3rSlam(model, conf, (H, W), buffer=frames.shape[0] + 8, device="cuda:0")
fr, it, n = frames.to("cuda:0"), torch.from_numpy(INTR), frames.shape[0]
for k in range(n):
    slam.run(k, fr[k:k + 1], it, fr[k:k + 1], it, second_last_frame=(k == n - 2), last_frame=(k == n - 1))
torch.cuda.synchronize()
print("HIP run medium", round(time.time() - t, 1), "s", flush=True)
# production-shape oracle: one 2-view window, threads 16 vs 8
from cut3r_slam_amd.config import production_config
from oracle import cut3r_oracle as O
cfgp = production_config(); sdp = synth.tracking_state_dict(cfgp, 0, enc_residual_gain=0.1)
x = O.normalize(synth.pan_stream(2, 384, 512, pool=9, num=6, den=1, seed=0))
for nt in (16, 8):
    torch.set_num_threads(nt)
    t = time.time(); O.forward_views(cfgp, sdp, x, minimal=True); print("oracle production 2 views threads", nt, round(time.time() - t, 1), "s", flush=True)
