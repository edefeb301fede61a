"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's first loop-closure optimisation
(/root/reference/hislam2/track_backend.py:256-299): torch.optim.Adam over B-1 se(3) vectors, SE3.exp(.).matrix()
realised as torch.matrix_exp of the 4x4 twist (the same function lietorch evaluates; autograd through it), fp64."""
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
