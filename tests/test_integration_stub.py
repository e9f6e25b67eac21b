"""The ctypes stub printed in INTEGRATION.md, executed exactly as written (a reference maintainer would paste it)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_integration_md_stub_runs_and_matches_the_package():
    from dicp_amd import _lib, _ops
    from dicp_amd.synthetic import make_pairs
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    code = md[md.index("# dICP/_hip.py"):]
    code = code[:code.index("```")]
    code = code.replace('ctypes.CDLL("libdicp_hip.so")', "ctypes.CDLL(%r)" % _lib.LIB_PATH)
    ns = {}
    exec(code, ns)
    src, tgt = make_pairs(3, 500, 700, seed=1)
    src, tgt = src.cuda(), tgt.cuda()
    ang = 0.2
    C = torch.tensor([[1, 0, 0], [0, torch.cos(torch.tensor(ang)), -torch.sin(torch.tensor(ang))],
                      [0, torch.sin(torch.tensor(ang)), torch.cos(torch.tensor(ang))]], device="cuda").repeat(3, 1, 1)
    r = torch.tensor([[0.1], [-0.2], [0.05]], device="cuda").repeat(3, 1, 1)
    idx = ns["nearest_index"](src, C, r, ns["pack"](tgt), tgt.shape[1])
    pose = torch.cat((C.reshape(3, 9), r.reshape(3, 3)), dim=1).contiguous()
    ref = _ops.knn(src, pose, _ops.pack_target(tgt), 700, _lib.KNN_VALU)
    assert torch.equal(idx, ref)
