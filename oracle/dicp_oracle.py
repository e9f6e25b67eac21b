"""CPU oracle for the differentiable-ICP hot path.

TEST INFRASTRUCTURE ONLY.  This file is the *checker*, never the product: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it.  ``dicp_amd`` never does, and has no CPU compute path of its own.

It restates, with PyTorch-CPU ops issued in the reference's order (so that both
its results and its cost profile are the reference's), the per-iteration path of
utiasASRL/dICP:

* correspondence        /root/reference/dICP/nn.py:11-125
* trim / robust weights /root/reference/dICP/loss.py:11-58
* the Gauss-Newton loop /root/reference/dICP/ICP.py:88-303

Parity is PINNED: ``tests/golden/*.npz`` were produced by importing the reference
itself in the build container (``tests/golden/make_golden.py``) and
``tests/test_oracle_golden.py`` holds this file to those vectors (poses, costs,
per-iteration deltas, gradients) as well as to the reference's own known-answer
test (/root/reference/tests/test_nn.py:10,20-21,36-37).

Inputs to :func:`icp_batched` are the already-batched tensors that the reference's
``batch_size_handling`` (ICP.py:305-511) hands to its loop.
"""
import torch


# --------------------------------------------------------------------------- nn
def handle_dimensions(x, y):
    """Layout normalisation, nn.py:94-125 (including its quirks).

    x: (n,3) | (N,n,3) | (N,3,n) | (N,6,n>6)  ->  (N,n,3)
    y: (m,c) | (c,m) | (N,m,c) | (N,c,m), c in {3,6}  ->  (N,m,c)
    """
    xu = x.unsqueeze(0) if x.dim() == 2 else x
    rows, cols = xu.shape[-2], xu.shape[-1]
    if rows == 3 or (rows == 6 and rows < cols):
        # nn.py:109 indexes the *original* x with three subscripts, so a 2-D
        # (3,n) query raises IndexError there; keep that behaviour.
        xu = x[:, :3, :].transpose(1, 2)
    assert xu.shape[2] == 3, "x must have 3 elements in the second dimension."

    yu = y.unsqueeze(0) if y.dim() == 2 else y
    rows, cols = yu.shape[-2], yu.shape[-1]
    if rows == 3 or (rows == 6 and rows < cols):
        yu = yu.transpose(1, 2)
    assert yu.shape[2] in (3, 6), "y must have 3 or 6 elements in the second dimension."
    return xu, yu


NN_CHUNK = None     # rows of x per cdist call (None: all at once, as the reference does).  The (N,n,m) matrix of a 65536-point
                    # cloud is 17 GB and autograd would keep one per iteration; the argmin carries no gradient (nn.py:35), so
                    # taking it chunk by chunk under no_grad returns the same indices and the same graph for what follows.


def nn_index(x, y):
    """Hard 1-NN index, nn.py:32-35 (== :83-86): cdist -> argmin (ties: lowest index)."""
    if NN_CHUNK is None:
        d = torch.cdist(x, y[:, :, :3], p=2)
        return torch.argmin(d, dim=2)
    with torch.no_grad():
        yy = y[:, :, :3].detach()
        return torch.cat([torch.argmin(torch.cdist(x[:, s:s + NN_CHUNK].detach(), yy, p=2), dim=2)
                          for s in range(0, x.shape[1], NN_CHUNK)], dim=1)


def nn_hard(x, y):
    """nn.py:23-40 / 72-92: neighbours = whole target rows gathered at the argmin."""
    idx = nn_index(x, y)
    sel = idx.unsqueeze(2).repeat(1, 1, y.shape[-1])
    return torch.gather(input=y, dim=1, index=sel)


def nn_gumbel(x, y, eps, tau, U=None):
    """Gumbel-softmax soft neighbour, nn.py:43-70.  ``U`` injects the uniform draw
    (the reference calls ``torch.rand`` at :60) so that results are reproducible."""
    diff = x.unsqueeze(2) - y.unsqueeze(1)[:, :, :, :3]
    d2 = torch.sum(diff ** 2, dim=3)
    if U is None:
        U = torch.rand(d2.shape, device=d2.device)
    g = -torch.log(-torch.log(U + eps) + eps)
    p = torch.softmax((-d2 + g) / tau, dim=2)
    return p @ y


def find_nn(x, y, differentiable=True, use_gumbel=True, eps=1e-20, tau=0.1, U=None):
    """nn.find_nn, nn.py:11-21."""
    xu, yu = handle_dimensions(x, y)
    if differentiable and use_gumbel:
        return nn_gumbel(xu, yu, eps, tau, U)
    return nn_hard(xu, yu)


# ------------------------------------------------------------------------- loss
def loss_weight(err, name, metric, differentiable=False, tanh_steepness=10.0):
    """loss.get_weight, loss.py:11-58.  err: (n,r) or (N,n,r) -> (n,) or (N,n)."""
    axis = 1 if err.dim() == 2 else 2
    if name == "huber":
        e = torch.linalg.norm(err, axis=axis)
        if differentiable:                                   # pseudo-Huber, :30
            return metric ** 2 / (metric ** 2 + e ** 2)
        return torch.where(e > metric, metric / e, torch.ones_like(e))   # :32
    if name == "cauchy":                                     # :41
        return 1.0 / (1.0 + (torch.linalg.norm(err, axis=axis) / metric) ** 2)
    if name == "trim":
        e = torch.linalg.norm(err, axis=axis)
        if differentiable:                                   # :54
            return 0.5 * torch.tanh(tanh_steepness * (metric - e) - 3.0) + 0.5
        shp = (err.shape[0], 1) if err.dim() == 2 else (err.shape[0], err.shape[1])
        one = torch.ones(shp, dtype=err.dtype, device=err.device)
        return torch.where(e < metric, one, torch.zeros_like(one))       # :56-58
    raise ValueError("Invalid loss name: {}".format(name))


# -------------------------------------------------------------------------- ICP
def skew(v):
    """ICP.py:513-531: (N,n,3) -> (N,n,3,3) with v^ = [[0,-z,y],[z,0,-x],[-y,x,0]]."""
    x, y, z = v[:, :, 0], v[:, :, 1], v[:, :, 2]
    o = torch.zeros_like(x)
    return torch.stack([o, -z, y, z, o, -x, -y, x, o], dim=2).view(v.shape[0], v.shape[1], 3, 3)


def icp_batched(source, target, T_init, w_init, *, icp_type="pt2pl", differentiable=True,
                max_iterations=100, tolerance=1e-12, trim_dist=None, loss_fn=None, dim=3,
                const_iter=False, tanh_steepness=5.0, match_ratio_thresh=0.0,
                use_gumbel=False, gumbel_eps=1e-10, gumbel_tau=0.1, record=None):
    """The loop of ICP.dICP, ICP.py:88-303, on batched inputs.

    source (N,n,3), target (N,m,3|6), T_init (N,4,4), w_init (N,n) for pt2pl or
    (N,3n) for pt2pt (ICP.py:508-509).  ``record``: optional dict that receives
    per-iteration lists ``idx, A, b, delta, C, r`` (detached) for kernel-level tests.
    """
    assert dim in (2, 3), "dim must be 2 or 3"
    N = source.shape[0]
    dev, dt = source.device, source.dtype
    deltas, weights, costs = [], [], []
    converged = torch.zeros(N, dtype=torch.bool, device=dev)
    num_iters = torch.zeros(N, dtype=dt, device=dev)
    match_ratio = torch.zeros(N, dtype=dt, device=dev)
    assert source.dtype == target.dtype == T_init.dtype      # ICP.py:96

    if icp_type == "pt2pl":
        assert target.shape[2] == 6                           # ICP.py:103
    else:
        target = target[:, :, :3]                             # ICP.py:105

    if dim == 2:                                              # ICP.py:107-116
        s2 = torch.zeros((N, source.shape[1], source.shape[2]), dtype=dt, device=dev)
        s2[:, :, :2] = source[:, :, :2]
        source = s2
        t2 = torch.zeros((N, target.shape[1], target.shape[2]), dtype=dt, device=dev)
        t2[:, :, :2] = target[:, :, :2]
        if icp_type == "pt2pl":
            t2[:, :, 3:5] = target[:, :, 3:5]
        target = t2

    C = T_init[:, 0:3, 0:3]                                   # ICP.py:125-129
    r = T_init[:, 0:3, 3:]
    ps = source.transpose(1, 2)
    ii = -1
    w = w_init
    for ii in range(max_iterations):
        pt = C @ ps + r                                       # :137
        nbr = find_nn(pt, target, differentiable, use_gumbel, gumbel_eps, gumbel_tau).transpose(1, 2)  # :140
        if record is not None:
            xu, yu = handle_dimensions(pt.detach(), target.detach())
            record.setdefault("idx", []).append(nn_index(xu, yu))
            record.setdefault("C", []).append(C.detach().clone())
            record.setdefault("r", []).append(r.detach().clone())

        e3 = (pt - nbr[:, :3]).transpose(1, 2)                # :143-149
        if icp_type == "pt2pl":
            nrm = nbr[:, 3:].transpose(1, 2)
            err = torch.sum(e3 * nrm, axis=2).unsqueeze(-1)
        else:
            err = e3

        tw = torch.ones((N, e3.shape[1]), dtype=dt, device=dev)          # :152-155
        if trim_dist is not None and trim_dist >= 0.0:
            tw = loss_weight(e3, "trim", trim_dist, differentiable, tanh_steepness)
        lw = torch.ones((N, err.shape[1]), dtype=dt, device=dev)         # :157-160
        if loss_fn is not None:
            lw = loss_weight(err, loss_fn["name"], loss_fn["metric"], differentiable, tanh_steepness)
        if icp_type == "pt2pt":                                          # :162-166
            tw = tw.repeat_interleave(3, dim=1)
            lw = lw.repeat_interleave(3, dim=1)
            err = err.reshape(N, -1, 1)
        w = w_init * tw * lw                                             # :169

        q = (C @ ps).transpose(1, 2)                                     # :171-183
        if icp_type == "pt2pl":
            JC = (skew(q).transpose(2, 3) @ nrm.unsqueeze(-1)).squeeze(-1)
            Jr = -nrm
        else:
            JC = skew(q).view(N, -1, 3)
            Jr = -torch.eye(3, device=dev).repeat(N, q.shape[1], 1)
        J = torch.cat((JC, Jr), dim=2)
        if dim == 2:                                                     # :186-189
            D = torch.zeros((6, 3), dtype=dt, device=dev)
            D[2, 0] = D[3, 1] = D[4, 2] = 1.0
            J = J @ D

        ws = torch.sqrt(w + 1.0e-10) - 1.0e-5                            # :194-196
        ew = ws.unsqueeze(-1) * err
        Jw = ws.unsqueeze(-1) * J
        JwT = Jw.transpose(1, 2)                                         # :199-201
        A = JwT @ Jw + 1e-12 * torch.eye(J.shape[2], dtype=dt, device=dev)
        step = -torch.linalg.inv(A) @ JwT @ ew
        if record is not None:
            record.setdefault("A", []).append(A.detach().clone())
            record.setdefault("b", []).append((JwT @ ew).detach().clone())
        if dim == 2:                                                     # :204-207
            full = torch.zeros((N, 6, 1), dtype=dt, device=dev)
            full[:, 2:5] = step
            step = full
        if record is not None:
            record.setdefault("delta", []).append(step.detach().clone())

        dC = torch.matrix_exp(skew(step[:, 0:3].transpose(1, 2)).squeeze(1))   # :210-217
        C = dC.transpose(1, 2) @ C
        r = r - step[:, 3:6]

        deltas.append(step.detach())                                     # :220-234
        wk = w.detach()
        if weights:
            allzero = (torch.sum(wk, dim=1) == 0.0).unsqueeze(-1) * torch.ones_like(wk)
            wk = torch.where(allzero != 0, weights[-1].squeeze(-1), wk)
        weights.append(wk.unsqueeze(-1))
        cost = (ew.transpose(1, 2) @ ew).detach()
        if costs:
            cost = torch.where(cost == 0.0, costs[-1], cost)
        costs.append(cost)

        nrm_step = torch.linalg.norm(step, axis=1).detach().squeeze(-1)  # :237-260
        hit = nrm_step < tolerance
        converged = torch.where(hit, torch.ones_like(hit), converged)
        if bool(hit.any()) and not const_iter:
            num_iters = torch.where(hit, num_iters + (ii + 1) * (num_iters == 0), num_iters)
            cur = torch.sum(w > match_ratio_thresh, dim=1)
            start = torch.sum(w_init > match_ratio_thresh, dim=1)
            start[start == 0] = 1
            match_ratio = torch.where(hit, match_ratio + cur / start * (match_ratio == 0), match_ratio)
            w_init = w_init * torch.where(hit, torch.zeros_like(nrm_step), torch.ones_like(nrm_step)).unsqueeze(-1)
            if bool(hit.all()):
                break

    num_iters = torch.where(num_iters == 0, ii + 1, num_iters)           # :267-271
    cur = torch.sum(w > match_ratio_thresh, dim=1)
    start = torch.sum(w_init > match_ratio_thresh, dim=1)
    start[start == 0] = 1
    match_ratio = torch.where(match_ratio == 0, cur / start, match_ratio)

    pc = (C @ ps + r).transpose(1, 2)                                    # :274-280
    T = torch.diag_embed(torch.ones((N, 4), dtype=dt, device=dev))
    T[:, 0:3, 0:3] = C
    T[:, 0:3, 3] = r.squeeze(-1)
    return {                                                             # :283-303
        "pc": pc,
        "T": T,
        "costs": torch.stack(costs, dim=1).squeeze(-1),
        "deltas": torch.stack(deltas, dim=1),
        "weights": torch.stack(weights, dim=1),
        "stats": {"converged": converged, "iterations": num_iters, "matched_ratio": match_ratio},
    }


def knn_exact_f64(x, y):
    """Exact brute-force 1-NN in float64 with explicit differences (no
    ||x||^2+||y||^2-2xy cancellation); returns (idx, d2_best, d2_second).
    Used to judge index parity of fp32 kernels on large synthetic clouds: a
    mismatch only counts when best and runner-up are separated by more than the
    fp32 resolution of the expanded form ('Index parity')."""
    x = x.double()
    y = y[:, :, :3].double()
    idx = torch.empty(x.shape[:2], dtype=torch.long)
    best = torch.empty(x.shape[:2], dtype=torch.float64)
    second = torch.empty(x.shape[:2], dtype=torch.float64)
    step = max(1, (1 << 24) // max(1, y.shape[1]))
    for b in range(x.shape[0]):
        for s in range(0, x.shape[1], step):
            d = ((x[b, s:s + step, None, :] - y[b, None, :, :]) ** 2).sum(-1)
            k = min(2, d.shape[1])
            v, i = torch.topk(d, k, dim=1, largest=False)
            # lowest index among exact ties, like argmin
            idx[b, s:s + step] = torch.argmin(d, dim=1)
            best[b, s:s + step] = v[:, 0]
            second[b, s:s + step] = v[:, -1]
    return idx, best, second
