import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "tests"))
import traceback
import test_gpu_fuzz as F
bad = 0
for seed in range(12, 150):
    for kind in ("plain", "plane", "dups", "far"):
        try:
            F.test_sweep_path_equals_brute_path(seed, kind)
        except Exception as e:
            bad += 1
            print("FAIL", seed, kind, repr(e)[:200])
for seed in range(8, 60):
    try:
        F.test_every_knn_form_returns_the_same_indices(seed)
    except Exception as e:
        bad += 1
        print("FAIL knn", seed, repr(e)[:200])
print("done, failures:", bad)
