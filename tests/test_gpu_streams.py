"""The library is ordered by the caller's stream: a call issued under torch.cuda.stream(s) runs on s (the raw handle PyTorch reports
as current), and returns the same bits as on the default stream.  `-m gpu`."""
import pytest
import torch

from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_pairs

pytestmark = pytest.mark.gpu


def test_side_stream_equals_default_stream():
    N, n = 6, 4096
    src, tgt = make_pairs(N, n, n, seed=2)
    src, tgt = src.cuda(), tgt.cuda()
    T0 = torch.eye(4, device="cuda").repeat(N, 1, 1)
    kw = dict(trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0})

    def run():
        icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=8, tolerance=1e-12)
        icp.const_iter = True
        s, t = src.detach().requires_grad_(True), tgt.detach().requires_grad_(True)
        out = icp.icp(s, t, T0, **kw)
        out["T"].sum().backward()
        return out["T"].detach(), s.grad, t.grad

    ref = run()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        # a long kernel first: if the call below were issued on the default stream it would overtake it and read stale inputs
        busy = torch.empty((256, 1024, 1024), device="cuda").normal_()
        src.add_(0.0)                                   # (inputs touched on the side stream, behind the long kernel)
        got = run()
    side.synchronize()
    torch.cuda.current_stream().wait_stream(side)
    assert torch.equal(ref[0], got[0])
    assert torch.allclose(ref[1], got[1], rtol=0, atol=1e-6) and torch.allclose(ref[2], got[2], rtol=0, atol=1e-6)
    del busy
