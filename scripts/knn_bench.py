"""A/B timing of the kNN launch configurations at the benchmark shape (interleaved rounds, HIP events)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dicp_amd import _lib, _ops
from dicp_amd.synthetic import make_pairs

B = int(os.environ.get("B", 256)); n = int(os.environ.get("NPTS", 16384)); rounds = int(os.environ.get("ROUNDS", 5))
src, tgt = make_pairs(B, n, n, seed=3)
src, tgt = src.cuda(), tgt.cuda()
if os.environ.get("NEAR") == "1":      # converged-pose queries: every query 0.01-noise away from a target
    src = (tgt[:, torch.randperm(n, device="cuda"), :3] + 0.01 * torch.randn((B, n, 3), device="cuda")).contiguous()
tgt4 = _ops.pack_target(tgt)
idx = torch.empty((B, n), dtype=torch.int32, device="cuda")
variants = {"valu_q4c16": 1 | (5 << 8), "valu_q8c16w4": 1 | (11 << 8), "mfma_nb2g1": 2 | (1 << 8), "mfma_nb4g4": 2 | (5 << 8)}
sel = os.environ.get("VARIANTS")
if sel:
    variants = {k: v for k, v in variants.items() if k in sel.split(",")}
ref = None
times = {k: [] for k in variants}
for rnd in range(rounds + 1):
    for name, v in variants.items():
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        _ops.knn(src, None, tgt4, n, v, out=idx)
        b.record()
        torch.cuda.synchronize()
        if rnd:
            times[name].append(a.elapsed_time(b))
        else:
            if ref is None:
                ref = idx.clone()
            else:
                print("%-12s mismatches vs first variant: %d" % (name, int((idx != ref).sum())))
# exact sorted-sweep variants
sw = _ops.SweepIndex(tgt)
qo = sw.query_order(src, None)
for name, cfg, q in (("sweep_q1", 1, qo), ("sweep_q2", 2, qo), ("sweep_q4", 3, qo), ("sweep_q2_unsorted", 2, None),
                     ("sweep_q1c16", 4, qo), ("sweep_q2c16", 5, qo), ("sweep_q4c16", 6, qo), ("sweep_q3", 7, qo), ("sweep_q3c16", 8, qo), ("scan_w256g8", 16, qo), ("scan_w256g4", 17, qo), ("scan_w512g8", 18, qo), ("scan_w384g8", 19, qo)):
    if sel and name not in sel.split(","):
        continue
    ts = []
    for rnd in range(rounds + 1):
        sw.pair_shards.zero_()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        sw.knn(src, None, q, out=idx, cfg=cfg)
        b.record()
        torch.cuda.synchronize()
        if rnd:
            ts.append(a.elapsed_time(b))
        elif ref is not None:
            print("%-12s mismatches vs first variant: %d   pairs scored: %.2f%% of brute force" % (name, int((idx != ref).sum()), 100.0 * sw.pairs.item() / (float(B) * n * n)))
    times[name] = ts
pairs = float(B) * n * n
print("lib:", _lib.LIB_PATH)
for name, ts in times.items():
    ts = sorted(ts)
    med = ts[len(ts) // 2]
    print("%-12s median %.3f ms  min %.3f ms  -> %.1f TF (8nm)  %.2f Gpairs/s" % (name, med, ts[0], 8 * pairs / med / 1e9, pairs / med / 1e6))
