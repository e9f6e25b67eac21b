"""Nearest-neighbour correspondence with the reference's ``dICP.nn.nn`` interface
(/root/reference/dICP/nn.py:4-125).

``find_nn(x, y)`` returns, for every query point, the matched target ROW (xyz and, if
present, the normal).  The hard path (what ICP uses) runs the fused brute-force kernel of
libdicp_hip.so -- the (N,n,m) distance matrix of nn.py:32 is never materialised -- followed
by a row gather whose backward is a scatter-add (the only gradient path: argmin has none,
so ``x`` receives no gradient, exactly as in the reference).  The Gumbel-softmax path
(nn.py:43-70) runs as fused flash-style kernels (dicp_gumbel_nn / _bwd): online softmax over
LDS-tiled targets, noise generated in-kernel (or injected for parity tests), gradients to both
``x`` and ``y`` by recomputation -- the (N,n,m) tensors of the reference never exist.
"""
import torch

from . import _lib, _ops


class nn:
    def __init__(self, differentiable=True, use_gumbel=True, eps=1e-20, tau=0.1):
        self.differentiable = differentiable
        self.use_gumbel = use_gumbel
        self.eps = eps
        self.tau = tau
        self.knn_variant = _lib.KNN_AUTO

    def find_nn(self, x, y, U=None):
        """x: (n,3) | (N,n,3) | (N,3,n) | (N,6,n>6);  y: (m,c) | (c,m) | (N,m,c) | (N,c,m), c in {3,6}
        -> (N,n,c).  ``U`` (optional, build-specific) injects the uniform draw of nn.py:60."""
        x_use, y_use = self._handle_dimensions(x, y)
        home = x_use.device
        dev = home if x_use.is_cuda else _ops.compute_device()
        x_dev, y_dev = x_use.to(dev), y_use.to(dev)
        if self.differentiable and self.use_gumbel:                       # nn.py:14-18
            out = self._soft(x_dev, y_dev, None if U is None else U.to(dev))
        else:
            out = self._hard(x_dev, y_dev)
        return out if home == dev else out.to(home)

    def nn_index(self, x, y):
        """Index form of the hard path (build-specific helper): (N,n) int64."""
        x_use, y_use = self._handle_dimensions(x, y)
        dev = x_use.device if x_use.is_cuda else _ops.compute_device()
        return self._index(x_use.to(dev), y_use.to(dev)).long().to(x_use.device)

    # nn.py:23-40 and :72-92 (identical bodies in the reference)
    def _index(self, x, y):
        if x.dtype != y.dtype:
            raise TypeError("x and y must share a dtype, got %s and %s" % (x.dtype, y.dtype))
        xd, yd = x.detach().contiguous(), y.detach().contiguous()
        if (self.knn_variant & 0xff) == _lib.KNN_SWEEP:
            frame = _ops.search_frame(yd)                       # the cloud's search frame (see _ops.ICPLoop): rows Q y + t, pose [Q | t]
            pose_s = _ops.search_pose(None, frame)
            sw = _ops.SweepIndex(yd, frame=frame)
            return sw.knn(xd, pose_s, sw.query_order(xd, pose_s), cfg=(self.knn_variant >> 8) & 0xff)
        frame = _ops.search_frame(yd)
        return _ops.knn(xd, _ops.search_pose(None, frame), _ops.pack_target(yd, frame), y.shape[1], self.knn_variant)

    def _hard(self, x, y):
        return _ops.gather_rows(y, self._index(x, y))

    # nn.py:43-70 as fused kernels: online softmax over LDS-tiled targets, noise generated (or injected) in-kernel
    def _soft(self, x, y, U):
        if x.dtype != y.dtype:
            raise TypeError("x and y must share a dtype, got %s and %s" % (x.dtype, y.dtype))
        return _ops.gumbel_nn(x, y, self.eps, self.tau, U=U)

    @staticmethod
    def _handle_dimensions(x, y):
        """nn.py:94-125, quirks included."""
        x_use = x.unsqueeze(0) if x.dim() == 2 else x
        if x_use.shape[-2] == 3 or (x_use.shape[-2] == 6 and x_use.shape[-2] < x_use.shape[-1]):
            x_use = x[:, :3, :].transpose(1, 2)      # nn.py:109 subscripts the ORIGINAL x: 2-D (3,n) raises IndexError
        assert x_use.shape[2] == 3, "x must have 3 elements in the second dimension."
        y_use = y.unsqueeze(0) if y.dim() == 2 else y
        if y_use.shape[-2] == 3 or (y_use.shape[-2] == 6 and y_use.shape[-2] < y_use.shape[-1]):
            y_use = y_use.transpose(1, 2)
        assert y_use.shape[2] == 3 or y_use.shape[2] == 6, "y must have 3 or 6 elements in the second dimension."
        return x_use, y_use
