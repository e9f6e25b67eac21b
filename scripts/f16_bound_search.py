"""A SEARCH for the worst case of the matrix-core filter's error bound (csrc/knn_f16.hip, header): |filter - score()| / E over every (query, target) pair of
clouds built to stress each term of E -- through dicp_knn_f16_probe, which scores the pairs exactly as the searches do (same image, same query fragment, the
same MFMA).  The searches' index-identity rests on ratio <= 1; the MFMA-accumulation term of E was ASSUMED at 4x a measured worst case (4.97 u T over 4e5
dot products, scripts/ubench/mfma_f16_ubench.hip): this is the adversarial measurement behind that assumption.
usage (MI355X): PYTHONPATH=. python scripts/f16_bound_search.py > profiles/rNN_knn_f16_bound_search.txt        (tests/test_gpu_f16.py runs the same cases)"""
import ctypes
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dicp_amd import _lib, _ops


def surfaces(g, N, m, extent):
    return (torch.rand((N, m, 3), generator=g) - 0.5) * 2 * extent


def adversarial_cases(n=4096, m=4096, N=8, seed=0):
    """(name, src (N,n,3), tgt (N,m,3)) float32 clouds, identity pose.  The query the kernels score with is -(C p + r) = -p."""
    g = torch.Generator().manual_seed(seed)
    out = []
    y = surfaces(g, N, m, 40.0)
    out.append(("random, 80 m across", surfaces(g, N, n, 40.0), y))
    # maximal cancellation: x.y ~ 0.5|y|^2 -- the query IS a target (D = 0), and the cloud sits far from the origin relative to its size (T >> D)
    for off in (0.0, 300.0, 3000.0):
        t = surfaces(g, N, m, 20.0) + torch.tensor([off, -0.7 * off, 0.3 * off])
        idx = torch.randint(0, m, (N, n), generator=g)
        s = torch.gather(t, 1, idx[:, :, None].expand(N, n, 3)).clone()
        s[:, ::2] += 1e-3 * torch.randn((N, (n + 1) // 2, 3), generator=g)        # ... and every other one a millimetre off it
        out.append(("queries on targets, cloud %g m from the origin" % off, s, t))
    # f16-denormal halves: coordinates that f16 holds almost exactly -- the low term of the split is a denormal or zero (scale 2^k: extent 2^10..2^11 scaled)
    base = torch.randint(-1024, 1024, (N, m, 3), generator=g).float()
    t = base / 64.0                                                                   # |y| <= 16: s = 2^6..2^7, scaled values are integers (+ tiny)
    t = t + (torch.rand((N, m, 3), generator=g) - 0.5) * 2.0 ** -18                   # low terms of ~2^-12 scaled: f16 denormals (< 2^-14) and near them
    sq = torch.randint(-1024, 1024, (N, n, 3), generator=g).float() / 64.0 + (torch.rand((N, n, 3), generator=g) - 0.5) * 2.0 ** -19
    out.append(("f16-denormal low terms (grid coordinates + 2^-19)", sq, t))
    # the scale's boundaries: the largest coordinate exactly a power of two, just below it, just above it
    for name, top in (("extent exactly 2^5", 32.0), ("extent just below 2^5", 32.0 * (1 - 2.0 ** -23)), ("extent just above 2^5", 32.0 * (1 + 2.0 ** -22))):
        t = surfaces(g, N, m, 31.0)
        t[:, 0, 0] = top
        t[:, 1, 1] = -top
        out.append((name, surfaces(g, N, n, 31.0), t))
    # far rows at the edge of being left out of the image: rows at exactly 16x the rest's extent (kept: the scale then serves them, the rest loses 4 bits),
    # and just beyond (left out)
    for name, f in (("64 rows at exactly 16x the extent", 16.0), ("64 rows just inside 16x", 15.99), ("64 rows at 17x (left out of the image)", 17.0)):
        t = surfaces(g, N, m, 10.0)
        mx = t.abs().amax(dim=(1, 2))
        t[:, :64] = (mx[:, None, None] * f) * torch.sign(torch.randn((N, 64, 3), generator=g))
        s = surfaces(g, N, n, 10.0)
        s[:, :64] = t[:, :64] * (1 + 1e-6)                                            # some queries out there too
        out.append((name, s, t))
    # queries at the edge of the f16 range (|x| s up to 60000: ~29x the cloud's extent away) and wide dynamic range inside one cloud
    t = surfaces(g, N, m, 5.0)
    s = surfaces(g, N, n, 5.0)
    s[:, ::3] *= 28.0
    out.append(("queries up to 28x the extent away", s, t))
    t = surfaces(g, N, m, 1.0) * torch.logspace(-4, 1.5, m)[None, :, None]
    out.append(("targets over 5.5 decades of scale", surfaces(g, N, n, 1.0) * torch.logspace(-4, 1.5, n)[None, :, None], t))
    # heavy cancellation inside the dot product: x and y nearly orthogonal with large components of opposite sign
    t = surfaces(g, N, m, 30.0)
    t[:, :, 2] = -t[:, :, 0] + 1e-3 * torch.randn((N, m), generator=g)
    s = surfaces(g, N, n, 30.0)
    s[:, :, 2] = s[:, :, 0]
    out.append(("x.y cancels: x = (a,b,a), y = (c,d,-c)", s, t))
    return out


def probe(src, tgt):
    """-> (N,4) [max ratio, its error, its bound, pairs checked] per cloud."""
    lib = _lib.load()
    src, tgt = src.cuda().contiguous(), tgt.cuda().contiguous()
    N, n, _ = src.shape
    m = tgt.shape[1]
    tgt4 = _ops.pack_target(tgt)
    img = _ops.f16_image(tgt4, m)
    out = torch.zeros((N, 4), dtype=torch.float32, device="cuda")
    _lib.check(lib.dicp_knn_f16_probe(_ops._p(src), None, _ops._p(tgt4), _ops._p(img), None, None, N, n, m, tgt4.shape[1], _ops._p(out), _ops._stream()), "dicp_knn_f16_probe")
    torch.cuda.synchronize()
    return out.cpu()


if __name__ == "__main__":
    worst, pairs = 0.0, 0.0
    print("the matrix-core filter's error bound, searched for its worst case: max over all (query, image row) pairs of |filter / s^2 - score()| / E  (E: csrc/knn_f16.hip)")
    for seed in range(3):
        for name, s, t in adversarial_cases(seed=seed):
            r = probe(s, t)
            k = int(r[:, 0].argmax())
            worst, pairs = max(worst, float(r[:, 0].max())), pairs + float(r[:, 3].sum())
            print("seed %d  %-58s max ratio %.4f   (error %.3e against E %.3e; %.3g pairs)" % (seed, name, float(r[k, 0]), float(r[k, 1]), float(r[k, 2]), float(r[:, 3].sum())))
    print("worst ratio over %.3g pairs: %.4f  (the searches need <= 1; the header's accumulation term assumes 4x the measured worst case)" % (pairs, worst))
