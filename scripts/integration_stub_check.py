import ctypes, os, sys, re, torch
root = os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, root)
md = open(os.path.join(root, "INTEGRATION.md")).read()
code = md[md.index("# dICP/_hip.py"):]
code = code[:code.index("```")]
code = code.replace('ctypes.CDLL("libdicp_hip.so")', 'ctypes.CDLL(os.path.join(root, "dicp_amd", "libdicp_hip.so"))')
ns = {"os": os, "root": root}
exec(code, ns)
from dicp_amd import _ops, _lib
from dicp_amd.synthetic import make_pairs
src, tgt = make_pairs(3, 500, 700, seed=1); src, tgt = src.cuda(), tgt.cuda()
C = torch.eye(3, device="cuda").repeat(3, 1, 1); r = torch.zeros((3, 3, 1), device="cuda")
tgt4 = ns["pack"](tgt)
idx = ns["nearest_index"](src, C, r, tgt4, tgt.shape[1])
ref = _ops.knn(src, None, _ops.pack_target(tgt), 700, _lib.KNN_VALU)
torch.cuda.synchronize()
print("INTEGRATION.md stub runs; indices equal the package's:", bool(torch.equal(idx, ref)))
