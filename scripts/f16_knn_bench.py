"""The matrix-core search at a BASELINE shape, for rocprofv3: B x n x m brute force, VALU and MFMA forms, a few launches each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dicp_amd import _lib, _ops
from dicp_amd.synthetic import make_pairs

B = int(os.environ.get("B", 256)); n = int(os.environ.get("NPTS", 16384)); reps = int(os.environ.get("REPS", 5))
forms = os.environ.get("FORMS", "valu,mfma").split(",")
chunk = min(B, 64)
parts = [make_pairs(chunk, n, n, seed=3 + i) for i in range(B // chunk)]
src = torch.cat([p[0] for p in parts]).cuda(); tgt = torch.cat([p[1][:, :, :3] for p in parts]).cuda()
del parts
frame = _ops.search_frame(tgt)
tgt4 = _ops.pack_target(tgt, frame)
pose = _ops.search_pose(None, frame, B)
img = _ops.f16_image(tgt4, n)
idx = torch.empty((B, n), dtype=torch.int32, device="cuda")
res = {}
for form in forms:
    v = _lib.KNN_MFMA if form == "mfma" else _lib.KNN_VALU
    ts = []
    for r in range(reps + 1):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); _ops.knn(src, pose, tgt4, n, v, out=idx, image=img); b.record(); torch.cuda.synchronize()
        if r:
            ts.append(a.elapsed_time(b))
    res[form] = idx.clone()
    ts.sort()
    pairs = float(B) * n * n
    print("%-5s B=%d n=m=%d: median %.3f ms  min %.3f ms   %.1f Tpairs/s  (8nm flop: %.0f TF)" % (form, B, n, ts[len(ts) // 2], ts[0], pairs / ts[len(ts) // 2] / 1e9, 8 * pairs / ts[len(ts) // 2] / 1e9))
if len(res) == 2:
    print("mismatches:", int((res["valu"] != res["mfma"]).sum()))
again, scan = _ops.f16_counters(img, B, tgt4.shape[1])
print("second filter pass: %.4f %% of the queries per launch; exact scans: %.4f %%" % (100.0 * again / (B * n) / max(1, (reps + 1) * forms.count("mfma")), 100.0 * scan / (B * n) / max(1, (reps + 1) * forms.count("mfma"))))
