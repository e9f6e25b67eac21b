"""The per-call index build (key sort + rows) at the BASELINE cloud sizes: dicp_sweep_sort / dicp_sweep_build vs torch.sort of the same keys."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from dicp_amd import _ops
for (B, n, dt) in ((256, 16384, torch.float32), (64, 65536, torch.float32), (256, 65536, torch.float32), (64, 16384, torch.float64)):
    tgt = (torch.rand((B, n, 6), device="cuda", dtype=dt) * 20 - 10)
    def timeit(fn):
        fn(); torch.cuda.synchronize(); ts = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
        return sorted(ts)[2]
    t_idx = timeit(lambda: _ops.SweepIndex(tgt, sorted_rows=True))
    t_sort = timeit(lambda: torch.sort(tgt[:, :, 0].contiguous(), dim=1, stable=True))
    print("B=%4d n=%6d %s: SweepIndex (sort + table + rows) %.3f ms   torch.sort of the keys alone %.3f ms" % (B, n, str(dt)[6:], t_idx, t_sort), flush=True)
