import ctypes, os, sys
sys.path.insert(0, ".")
import numpy as np, torch
os.environ.setdefault("FORMS", "mfma")
exec(open("scripts/f16_sweep_bench.py").read())
lib = _lib.load()
n_w = 32768
buf = (ctypes.c_ulonglong * (8 * n_w))()
lib.dicp_dbg_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
torch.cuda.synchronize()
print("rc", lib.dicp_dbg_stamps(buf, 8 * n_w))
a = np.frombuffer(buf, dtype=np.uint64).reshape(n_w, 8).astype(np.int64)
d = np.diff(a[:, :7], axis=1)
names = ["prologue", "sweep loop", "wait row cache", "merge + margins", "rescoring", "rare paths (ties, scan, pass 2)"]
for i, nm in enumerate(names):
    print("%-34s median %8.0f  mean %8.0f  p90 %8.0f cycles" % (nm, np.median(d[:, i]), d[:, i].mean(), np.percentile(d[:, i], 90)))
print("total per wave: median %.0f" % np.median(a[:, 6] - a[:, 0]))
