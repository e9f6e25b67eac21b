"""Per-point robust weights with the reference's ``dICP.loss.loss`` interface
(/root/reference/dICP/loss.py:3-58).  Inside ICP these formulas are fused into the
accumulate kernel; this class serves direct callers through dicp_loss_weight{,_bwd}.
"""
import torch

from . import _ops


class loss:
    def __init__(self, name="huber", metric=1.0, differentiable=False, tanh_steepness=10.0):
        self.name = name                      # "huber" | "cauchy" | "trim"
        self.metric = metric
        self.differentiable = differentiable
        self.tanh_steepness = tanh_steepness

    def get_weight(self, err):
        """err: (n,r) or (N,n,r), r in 1..3  ->  (n,) or (N,n)   (loss.py:11-19)."""
        if self.name not in ("huber", "cauchy", "trim"):
            raise ValueError("Invalid loss name: {}".format(self.name))             # loss.py:19
        home = err.device
        dev = home if err.is_cuda else _ops.compute_device()
        e = err.to(dev)
        lead = e.shape[:-1]
        w = _ops.loss_weight(e.reshape(-1, e.shape[-1]), self.name, self.differentiable,
                             self.metric, self.tanh_steepness).reshape(lead)
        if self.name == "trim" and not self.differentiable and err.dim() == 2:
            # loss.py:49,56-58 broadcasts (n,) against (n,1) for 2-D input: keep that shape
            w = w.unsqueeze(0).expand(err.shape[0], err.shape[0])
        return w if home == dev else w.to(home)
