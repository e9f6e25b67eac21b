"""Seeded soak of the match certificates beyond the 24 cases of tests/test_gpu_cert_soak.py (same generator).
usage: python scripts/cert_soak.py [cases] [seed]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_cert_soak import run_case     # noqa: E402
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
bad = 0
for c in range(cases):
    same, gok, what = run_case(c, seed)
    print("case %2d %s: %s, gradients %s" % (c, what, "IDENTICAL" if same else "DIFFERENT", "ok" if gok else "OFF"), flush=True)
    bad += (not same) or (not gok)
print("%d of %d cases failed" % (bad, cases))
sys.exit(1 if bad else 0)
