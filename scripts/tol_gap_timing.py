"""Tolerance mode at the headline shape: the host's time between learning K and the reverse sweep's first launch, split (perf_counter; medians over 30 calls).
usage (MI355X): PYTHONPATH=. python scripts/tol_gap_timing.py"""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dicp_amd import _loop, _ops
from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_pairs

B, n = 256, 16384
src, tgt = make_pairs(B, n, n, seed=3, dtype=torch.float32)
S, Tg = src.cuda(), tgt.cuda()
Ti = torch.eye(4).repeat(B, 1, 1).cuda()
kw = dict(trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0}, dim=3)
icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=50, tolerance=1e-4)
icp.const_iter = False
marks = {}
orig_conv, orig_bwd = _loop._converged_at, _loop.backward_once


def conv(p):
    r = orig_conv(p)
    if r is not None and "k_known" not in marks:
        marks["k_known"] = time.perf_counter()
    return r


def bwd(*a, **k):
    marks["bwd_entry"] = time.perf_counter()
    r = orig_bwd(*a, **k)
    marks["bwd_enqueued"] = time.perf_counter()
    return r


_loop._converged_at, _loop.backward_once = conv, bwd
rows = []
for it in range(40):
    marks.clear()
    s, t = S.detach().requires_grad_(True), Tg.detach().requires_grad_(True)
    t_start = time.perf_counter()
    out = icp.icp(s, t, Ti, **kw)
    marks["fwd_return"] = time.perf_counter()
    loss = out["T"].sum()
    marks["loss"] = time.perf_counter()
    loss.backward()
    marks["bwd_return"] = time.perf_counter()
    torch.cuda.synchronize()
    marks["done"] = time.perf_counter()
    if it >= 10:
        rows.append((marks["fwd_return"] - marks["k_known"], marks["loss"] - marks["fwd_return"], marks["bwd_entry"] - marks["loss"],
                     marks["bwd_enqueued"] - marks["bwd_entry"], marks["done"] - marks["bwd_enqueued"], marks["done"] - t_start))
names = ("K known -> icp() returns", "T.sum()", "backward(): autograd -> backward_once entry", "backward_once (plan, allocations, the one library call)", "rest of the GPU's backward after the host is done", "whole call")
for i, nm in enumerate(names):
    print("%-62s %.3f ms" % (nm, statistics.median(r[i] for r in rows) * 1e3))
