import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from dicp_amd import _lib, _ops
from oracle import dicp_oracle as O
DEV="cuda"
g = torch.Generator().manual_seed(3)
base = torch.rand((2, 50, 3), generator=g, dtype=torch.float32) * 6
y = base.repeat(1, 7, 1)[:, torch.randperm(350, generator=g)]
y = torch.cat((y, torch.full((2, 30, 3), 6000.0)), dim=1).to(DEV)
x = (base[:, :40] + 0.01 * torch.rand((2, 40, 3), generator=g)).to(DEV)
brute = _ops.knn(x, None, _ops.pack_target(y), 380, _lib.KNN_VALU).cpu().long()
ref = O.nn_index(x.cpu().double(), y.cpu().double())
bad = (brute != ref).nonzero()
print("n bad", len(bad))
yc = y.cpu().double(); xc = x.cpu().double()
for b, i in bad[:6].tolist():
    jb, jr = int(brute[b, i]), int(ref[b, i])
    print(b, i, "brute", jb, yc[b, jb].tolist(), "ref", jr, yc[b, jr].tolist(), "d2", float(((xc[b,i]-yc[b,jb])**2).sum()), float(((xc[b,i]-yc[b,jr])**2).sum()))
    d = torch.cdist(xc[b:b+1], yc[b:b+1])[0, i]
    print("   cdist values at both:", float(d[jb]), float(d[jr]), "equal rows:", bool((yc[b,jb]==yc[b,jr]).all()))
