"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's loop-closure optimisations
(/root/reference/hislam2/track_backend.py:256-299 first loop, :400-461 later loops) and of the store updates around them
(:301-346, :463-516, :566-575): torch.optim.Adam over se(3) vectors, SE3.exp(.).matrix() realised as torch.matrix_exp of the
4x4 twist (the same function lietorch evaluates; autograd through it), fp64.  PARITY UNPINNED: lietorch is an empty
submodule in the reference and no fixture exists at this boundary; conventions (tangent = [tau, phi]) from its call sites."""
import torch


def twist(xi):
    tau, phi = xi[:, :3], xi[:, 3:]
    z = torch.zeros(xi.shape[0], dtype=xi.dtype)
    M = torch.stack([
        torch.stack([z, -phi[:, 2], phi[:, 1], tau[:, 0]], -1),
        torch.stack([phi[:, 2], z, -phi[:, 0], tau[:, 1]], -1),
        torch.stack([-phi[:, 1], phi[:, 0], z, tau[:, 2]], -1),
        torch.stack([z, z, z, z], -1)], 1)
    return M


def loop_closure_init(submaps, mask, cur, cur_lc, iters, lr=5e-4, dtype=torch.float64):
    """submaps [B,6,h,w,3]; mask bool [B-1,N]; cur, cur_lc [N,3].  Returns (xi [B,6], T [B,4,4], losses)."""
    sub = submaps.to(dtype)
    B = sub.shape[0]
    fl = torch.stack([sub[:, 0], sub[:, -1]], 1).reshape(B, 2, -1, 3)
    cur, cur_lc = cur.to(dtype).reshape(1, -1, 3), cur_lc.to(dtype).reshape(1, -1, 3)
    p = torch.nn.Parameter(torch.zeros(B - 1, 6, dtype=dtype))
    opt = torch.optim.Adam([{"params": p, "lr": lr}])
    lie0 = torch.zeros(1, 6, dtype=dtype)
    losses = []
    for _ in range(iters):
        opt.zero_grad()
        T = torch.matrix_exp(twist(torch.cat([lie0, p], 0)))
        R, t = T[:, :3, :3], T[:, :3, 3].unsqueeze(1)
        cur_al = torch.matmul(cur, R[-1].unsqueeze(0).transpose(1, 2)) + t[-1].unsqueeze(0)
        l_cur = (cur_al - cur_lc).abs().mean()
        fla = torch.matmul(fl, R.transpose(1, 2).unsqueeze(1)) + t.unsqueeze(1)
        l_fl = (fla[:-1, -1] - fla[1:, 0])[mask].abs().mean()
        loss = l_fl + l_cur
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    xi = torch.cat([lie0, p.detach()], 0)
    return xi, torch.matrix_exp(twist(xi)), losses


def loop_closure(submaps, lc_all, pm_cur, sub_cur_all, sub_matched_all, iters, lr=5e-4, dtype=torch.float64):
    """Second and later loops (/root/reference/hislam2/track_backend.py:400-461), fp64.  submaps [B,6,h,w,3] (global frame),
    lc_all [Bc,6,h,w,3] (re-tracked lc submaps, previous ones already moved by their matched transform), pm_cur [Bc,h,w,3] (the
    loops' current maps, global frame), sub_cur_all / sub_matched_all: submap index of every loop's current / matched
    keyframe.  Returns (xi [B,6], T [B,4,4], xi_m [Bc,6], Tm [Bc,4,4], losses)."""
    sub, lc, cur = submaps.to(dtype), lc_all.to(dtype), pm_cur.to(dtype)
    B, Bc = sub.shape[0], lc.shape[0]
    fl = torch.stack([sub[:, 0], sub[:, -1]], 1).reshape(B, 2, -1, 3)
    lc_fl = torch.stack([lc[:, 0], lc[:, -1]], 1).reshape(Bc, 2, -1, 3)
    cur = cur.reshape(Bc, -1, 3)
    sc = torch.as_tensor(sub_cur_all, dtype=torch.long)
    sm = torch.as_tensor(sub_matched_all, dtype=torch.long)
    a_lie = torch.nn.Parameter(torch.zeros(B - 1, 6, dtype=dtype))
    m_lie = torch.nn.Parameter(torch.zeros(Bc, 6, dtype=dtype))
    opt = torch.optim.Adam([{"params": a_lie, "lr": lr}, {"params": m_lie, "lr": lr}])
    lie0 = torch.zeros(1, 6, dtype=dtype)
    losses = []
    for _ in range(iters):
        opt.zero_grad()
        T = torch.matrix_exp(twist(torch.cat([lie0, a_lie], 0)))
        R, t = T[:, :3, :3], T[:, :3, 3].unsqueeze(1)
        Tm = torch.matrix_exp(twist(m_lie))
        Rm, tm = Tm[:, :3, :3], Tm[:, :3, 3].unsqueeze(1)
        fla = torch.matmul(fl, R.transpose(1, 2).unsqueeze(1)) + t.unsqueeze(1)
        lca = torch.matmul(lc_fl, Rm.transpose(1, 2).unsqueeze(1)) + tm.unsqueeze(1)
        cura = torch.matmul(cur, R[sc].transpose(1, 2)) + t[sc]
        fl_loss = (fla[:-1, -1] - fla[1:, 0]).abs().mean()
        matched_loss = (lca[:, 0] - fla[sm, 0]).abs().mean()
        current_lc_loss = (cura - lca[:, -1]).abs().mean()
        loss = fl_loss + current_lc_loss + matched_loss
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    xi = torch.cat([lie0, a_lie.detach()], 0)
    xi_m = m_lie.detach()
    return xi, torch.matrix_exp(twist(xi)), xi_m, torch.matrix_exp(twist(xi_m)), losses


class LoopCloser:
    """Stateful fp64 restatement of the store updates around the two optimisers (track_backend.py:301-346, 463-516, 566-575):
    submap rewrite, pose rewrite (c2w <- T_b c2w, keyframes 0..(sub+1)*5), bookkeeping of `closed_loop`.
    `submaps` [S,6,h,w,3], `conf_ds` [S,6,h,w], `pose` [K,7] (t, q_xyzw) are torch CPU tensors, updated in place."""

    def __init__(self, submaps, conf_ds, pose, iters, dtype=torch.float64):
        """dtype: arithmetic of the two optimisers (fp64 = the oracle; fp32 = what the reference's torch.optim.Adam on CUDA tensors
        computes in, used to measure how far two correct implementations of these L1 objectives drift apart)"""
        self.dtype = dtype
        self.sub, self.conf, self.pose, self.iters = submaps.double(), conf_ds.double(), pose.double(), iters
        self.initialized = False
        self.closed = {"idx_current": [], "idx_matched": [], "pointmaps_lc": []}
        self.losses = []

    @staticmethod
    def _pose_mat(p7):
        from scipy.spatial.transform import Rotation
        T = torch.eye(4, dtype=torch.float64)
        T[:3, :3] = torch.from_numpy(Rotation.from_quat(p7[3:].numpy()).as_matrix())
        T[:3, 3] = p7[:3]
        return T

    def _rewrite(self, sub1, T):
        from scipy.spatial.transform import Rotation
        B = sub1 + 1
        self.sub[:B] = torch.einsum("bij,bshwj->bshwi", T[:, :3, :3], self.sub[:B]) + T[:, :3, 3].reshape(B, 1, 1, 1, 3)
        for i in range(B * 5 + 1):
            b = min(i // 5, B - 1)
            Tn = T[b] @ self._pose_mat(self.pose[i])
            q = Rotation.from_matrix(Tn[:3, :3].numpy()).as_quat()
            self.pose[i] = torch.cat([Tn[:3, 3], torch.from_numpy(q)])

    def close(self, pm_lc, idx_matched, idx_current):
        """pm_lc [6,h,w,3]: the re-tracked submap [5 keyframes of the matched submap, current keyframe] (TrackBackend.track)"""
        sub1 = idx_current // 5
        pm_lc = pm_lc.double()
        if not self.initialized:
            B = sub1 + 1
            mask = (self.conf[:sub1, 5] > 0).reshape(B - 1, -1)
            cur = self.sub[sub1, idx_current % 5].reshape(-1, 3)
            xi, T, losses = loop_closure_init(self.sub[:B], mask, cur, pm_lc[-1].reshape(-1, 3), self.iters, dtype=self.dtype)
            T = T.double()
            self._rewrite(sub1, T)
            self.initialized = True
            stored = pm_lc
        else:
            prev_cur = self.closed["idx_current"]
            lc_all = torch.stack(self.closed["pointmaps_lc"] + [pm_lc], 0)
            pm_cur = torch.stack([self.sub[c // 5, c % 5] for c in prev_cur] + [self.sub[sub1, idx_current % 5]], 0)
            sc = [c // 5 for c in prev_cur] + [sub1]
            sm = [m // 5 for m in self.closed["idx_matched"]] + [idx_matched // 5]
            xi, T, xi_m, Tm, losses = loop_closure(self.sub[:sub1 + 1], lc_all, pm_cur, sc, sm, self.iters, dtype=self.dtype)
            T, Tm = T.double(), Tm.double()
            self._rewrite(sub1, T)
            lc_al = torch.einsum("kij,kshwj->kshwi", Tm[:, :3, :3], lc_all) + Tm[:, :3, 3].reshape(-1, 1, 1, 1, 3)
            for i in range(len(prev_cur)):
                self.closed["pointmaps_lc"][i] = lc_al[i]
            stored = lc_al[-1]
        self.closed["idx_current"].append(idx_current)
        self.closed["idx_matched"].append(idx_matched)
        self.closed["pointmaps_lc"].append(stored)
        self.losses.append(losses)
        return xi
