"""A/B timing of the two accumulate-backward kernels at the benchmark shape on REAL matches (one sweep-kNN pass
under the identity pose): dicp_accumulate_bwd (row atomics, original order) vs dicp_accumulate_bwd_window
(sorted space, LDS window).  `no_tgt` rows = the same launch without target gradients (streams only)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dicp_amd import _lib, _loop, _ops
from dicp_amd.synthetic import make_pairs

B = int(os.environ.get("B", 256)); n = int(os.environ.get("NPTS", 16384)); rounds = int(os.environ.get("ROUNDS", 7))
mode = os.environ.get("MODE", "pt2pl")
lib = _lib.load()
if os.environ.get("HARD") == "1":       # independently sampled, partially overlapping clouds a metre off (dicp_amd.synthetic.make_independent_pairs)
    from dicp_amd.synthetic import make_independent_pairs
    src, tgt = make_independent_pairs(B, n, n, seed=3, dtype=torch.float32, ragged=False)
else:
    src, tgt = make_pairs(B, n, n, seed=3)
src, tgt = src.cuda(), tgt.cuda()
if mode == "pt2pt":
    tgt = tgt[:, :, :3].contiguous()
c = tgt.shape[2]
cv = c
dt, code = src.dtype, _lib.F32
sw = _ops.SweepIndex(tgt)
m_pad = sw.tgs4.shape[1]
qo = sw.query_order(src, None)
idx = torch.empty((B, n), dtype=torch.int32, device="cuda")
spos = torch.empty((B, n), dtype=torch.int32, device="cuda")
pose = torch.tensor([[1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0]] * B, dtype=dt, device="cuda")
sw.knn(src, pose, qo, out=idx, spos=spos)
w0 = torch.ones((B, n), dtype=dt, device="cuda")
gs = torch.randn((B, 36), dtype=dt, device="cuda")
gb = torch.randn((B, 6), dtype=dt, device="cuda")
P = _loop.LoopConfig(icp_type=mode, differentiable=True, max_iterations=1, tolerance=0.0, trim_dist=5.0, loss_name="huber",
                    loss_metric=1.0, dim=3, const_iter=True, tanh_steepness=10.0, match_ratio_thresh=0.01).params()
p, st = _ops._p, _ops._stream()
src_s = _ops._gather_rows_raw(src, qo)
w_s = _ops._gather_rows_raw(w0.unsqueeze(-1), qo).squeeze(-1).contiguous()
tgt_s = _ops._gather_rows_raw(tgt, sw.tperm)
nb, nw = lib.dicp_accumulate_blocks(n), lib.dicp_window_blocks(code, n, m_pad)
gsrc, gw, gtgt = torch.zeros_like(src), torch.zeros_like(w0), torch.zeros_like(tgt)
part = torch.zeros((B, max(nb, nw), _lib.NBWD_PAD), dtype=dt, device="cuda")
wt = lib.dicp_window_rows(code)
slab = torch.zeros((B, nw, wt, cv), dtype=dt, device="cuda")
gfar = torch.zeros((B, m_pad, cv), dtype=dt, device="cuda")


def atomic(want):
    _lib.check(lib.dicp_accumulate_bwd(code, ctypes.byref(P), p(src), p(tgt), c, p(idx), p(pose), p(w0), None, p(gs), p(gb), None,
                                       B, n, n, p(gsrc), p(gtgt) if want else None, p(gw), p(part), st), "bwd")


def window(want):
    _lib.check(lib.dicp_accumulate_bwd_window(code, ctypes.byref(P), p(src_s), p(tgt_s), c, p(spos), p(spos), p(qo), p(pose), p(w_s), None, p(gs), p(gb), None,
                                              B, n, m_pad, p(gsrc), p(slab) if want else None, p(gfar) if want else None, p(gw), p(part), 0, st),
               "bwd_window")


def reduce_():
    _lib.check(lib.dicp_window_reduce(code, p(slab), p(spos), p(qo), p(sw.tperm), p(gfar), None, B, n, n, m_pad, cv, p(gtgt), c, 0, st), "reduce")


def fresh():
    for t_ in (gsrc, gw, gtgt, slab, gfar):
        t_.zero_()


# correctness of the windowed form against the atomic one (one call each on fresh accumulators)
fresh(); atomic(True); torch.cuda.synchronize()
ref = (gsrc.clone(), gw.clone(), gtgt.clone(), part[:, :nb].sum(dim=1).clone())
fresh(); window(True)
a_src, a_w = torch.zeros_like(src), torch.zeros_like(w0)
_lib.check(lib.dicp_permute_add_rows(code, p(gsrc), p(qo), B, n, n, n, 3, 3, p(a_src), n, 3, st), "permute")
_lib.check(lib.dicp_permute_add_rows(code, p(gw), p(qo), B, n, n, n, 1, 1, p(a_w), n, 1, st), "permute")
reduce_(); torch.cuda.synchronize()
far_rows = int((gfar.abs().sum(dim=2) != 0).sum())
print("window vs atomic: max|d gsrc| %.2e  |d gw| %.2e  |d gtgt| %.2e (max |gtgt| %.2e)  |d partials| rel %.2e   far rows %d (%.3f%%)"
      % ((a_src - ref[0]).abs().max().item(), (a_w - ref[1]).abs().max().item(), (gtgt - ref[2]).abs().max().item(), ref[2].abs().max().item(),
         (part[:, :nw].sum(dim=1) - ref[3]).abs().max().item() / max(1.0, ref[3].abs().max().item()), far_rows, 100.0 * far_rows / (B * n)))
fresh()

cases = {"atomic": lambda: atomic(True), "atomic_no_tgt": lambda: atomic(False), "window": lambda: window(True),
         "window_no_tgt": lambda: window(False), "window_reduce (once per call)": reduce_}
times = {k: [] for k in cases}
for rnd in range(rounds + 1):
    for name, fn in cases.items():
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        torch.cuda.synchronize()
        if rnd:
            times[name].append(a.elapsed_time(b))
sp = torch.gather(spos.long(), 1, qo.long())       # spos is indexed by query: bring it to slot order
slot = torch.arange(n, device="cuda")[None, :]
off = (sp - slot).abs().float()
print("B=%d n=m=%d %s  window blocks/cloud=%d rows/window=%d  |spos - slot|: median %.0f  p99 %.0f  max %.0f"
      % (B, n, mode, nw, wt, off.median().item(), off.flatten().kthvalue(int(0.99 * off.numel())).values.item(), off.max().item()))
for name, ts in times.items():
    ts = sorted(ts)
    print("%-32s median %.3f ms  min %.3f ms" % (name, ts[len(ts) // 2], ts[0]))
