"""Multi-GPU tracking: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the
CPU tests).

What shards: every `inference()` call re-initialises the recurrent state and the pose memory
(/root/reference/src/dust3r/model.py:819-822), so tracking WINDOWS are independent network evaluations; only the cheap
post-hoc chaining (/root/reference/hislam2/track_frontend.py:216-234: needs the previous window's last depth and pose)
and the graph update are sequential.  So per step each rank runs the ViT on ONE window (its 5 keyframe-filter encodes
+ the 6-view inference), the three consumed outputs (pts3d_in_self_view, conf_self, camera_pose) are exchanged with
ONE all_gather per tensor (19 MB per rank at 384x512 -- small messages: latency, not ring bandwidth, matters), and
every rank replays the chaining + graph update of the N windows in sequence order, keeping the keyframe store and the
graph replicated (no second collective, and any rank can serve the trajectory).
"""
from __future__ import annotations

from typing import Callable, List, Sequence, Tuple

import torch
import torch.distributed as dist


def window_ranges(first_t0: int, world: int, win: int = 5) -> List[Tuple[int, int]]:
    """keyframe ranges [t0, t1) of the `world` windows of one step; consecutive windows share one keyframe."""
    return [(first_t0 + win * j, first_t0 + win * j + win + 1) for j in range(world)]


def all_gather_outputs(outs: Sequence[torch.Tensor], world: int) -> List[List[torch.Tensor]]:
    """outs: this rank's tensors (same shapes on every rank).  Returns per-rank lists, in rank order."""
    gathered = []
    for t in outs:
        t = t.contiguous()
        buf = torch.empty((world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        if t.is_cuda:
            dist.all_gather_into_tensor(buf.view(-1), t.view(-1))       # one RCCL all-gather per tensor
        else:
            parts = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(parts, t)                                   # gloo (CPU tests)
            buf = torch.stack(parts, 0)
        gathered.append(buf)
    return [[g[r] for g in gathered] for r in range(world)]


class ShardedTracker:
    """Drives a replicated `Cut3rSlam` with window-sharded network inference."""

    def __init__(self, slam, world: int, rank: int, infer_fn: Callable = None, track_fn: Callable = None,
                 append_fn: Callable = None):
        self.slam, self.world, self.rank = slam, world, rank
        self.infer_fn = infer_fn or (lambda t0, t1: slam.tracker.infer(t0=t0, t1=t1))
        self.track_fn = track_fn or (lambda t0, t1, outs: slam.tracker.track(t0, t1, outputs=outs))
        self.append_fn = append_fn or self._append

    def frames_needed(self, total_steps: int, kf_every: int, win: int) -> int:
        """frames consumed by the 7-keyframe initialisation plus `total_steps` sharded steps"""
        return (7 + win * self.world * total_steps + 1) * kf_every + 1

    def _append(self, kf_index: int, frame, tstamp, intr, mine: bool):
        slam = self.slam
        # fixed cadence: the encoder pass of a keyframe is deferred to the window that owns it (TrackFrontend.window_features)
        slam.keyframes.append(tstamp, frame[0], None, None, None, None, intr, None, None)

    def step(self, frames, t, kf_every, win, intr):
        """Advance `world` windows (= world*win*kf_every frames).  Returns the new frame counter."""
        slam, world, rank = self.slam, self.world, self.rank
        tracker = slam.tracker
        first_t0 = tracker.t1 - 1
        ranges = window_ranges(first_t0, world, win)
        # 1. keyframe filter in fixed-cadence mode: every kf_every-th frame is a keyframe (motion_filter.py:83,109,124);
        #    every rank registers all of them, but only the owner of a window runs the encoder on its new keyframes
        n_frames = world * win * kf_every
        for f in range(t, t + n_frames):
            if f % kf_every == 0:
                k = slam.keyframes.counter.value
                owner = min(max((k - first_t0 - 1) // win, 0), world - 1) if k > first_t0 else 0
                self.append_fn(k, frames[f:f + 1], f, intr, owner == rank)
        # 2. this rank's window through the network
        t0, t1 = ranges[rank]
        outs = self.infer_fn(t0, t1)
        # 3. one exchange over xGMI
        per_rank = all_gather_outputs(outs, world)
        # 4. replicated sequential chaining + graph update, in window order
        for (a, b), o in zip(ranges, per_rank):
            self.track_fn(a, b, tuple(o))
            tracker.t1 = b
        return t + n_frames
