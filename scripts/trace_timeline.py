"""Timeline of the LAST (timed) ICP call in a rocprofv3 --kernel-trace of bench.py: kernel, duration, gap before it.

    rocprofv3 --kernel-trace -d out -o t --output-format csv -- python3 bench.py --no-cpu-baseline
    python scripts/trace_timeline.py out/t_kernel_trace.csv [--full]
"""
import csv, sys, collections

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    own = "(anonymous namespace)::" in n and "at::native" not in n
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    if own:
        return n.split("<")[0].split("(")[0]
    if n.startswith("Cijk"):
        return "GEMM"
    for key in ("segmented_sort", "fill_reverse_indices", "FillFunctor", "direct_copy", "copyBuffer", "where_kernel", "CompareEq",
                "MulFunctor", "DivFunctor", "sum_functor", "CUDAFunctor_add", "searchsorted", "BatchedCopy", "masked_fill", "gather", "arange",
                "compare_scalar", "clamp", "reciprocal"):
        if key in n:
            return "torch:" + key
    return n.split("<")[0].split("(")[0]


# the timed call = everything after the last big idle gap that precedes the last 10 kNN launches
knn = [i for i, r in enumerate(rows) if "knn_sweep" in r["Kernel_Name"] or "knn_valu_kernel" in r["Kernel_Name"] or "knn_scan" in r["Kernel_Name"]]
first = knn[-13] if len(knn) >= 13 else knn[0]      # 10 timed launches + 3 brute-force launches of the roofline leg after them
# the call starts a few dozen small set-up kernels before its first kNN launch: walk back until the idle gap that
# separates it from the previous (untimed, synchronised) call
lo = first
while lo > 0 and first - lo < 80 and int(rows[lo]["Start_Timestamp"]) - int(rows[lo - 1]["End_Timestamp"]) < 90_000:
    lo -= 1
# (a fast host leaves less than that between two calls: the call's own target sort is its first long kernel, at most a
# couple of fills precede it)
sorts = [i for i in range(lo, first) if "sort_keys_kernel" in rows[i]["Kernel_Name"] or "segmented_sort" in rows[i]["Kernel_Name"]]
if sorts:
    lo = sorts[-1]
    steps = 0
    while (lo > 0 and steps < 3 and ("Fill" in rows[lo - 1]["Kernel_Name"] or "search_frame" in rows[lo - 1]["Kernel_Name"])
           and int(rows[lo]["Start_Timestamp"]) - int(rows[lo - 1]["End_Timestamp"]) < 90_000):
        lo -= 1; steps += 1
# ... and ends at the first idle gap after its backward (the host synchronises there; later legs of bench.py follow)
hi = first
while hi + 1 < len(rows) and int(rows[hi + 1]["Start_Timestamp"]) - int(rows[hi]["End_Timestamp"]) < 200_000:
    hi += 1
seg = rows[lo:hi + 1]
t0 = int(seg[0]["Start_Timestamp"])
agg = collections.OrderedDict()
prev = None
gaps = 0.0
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = max(0.0, (s - prev) / 1e3) if prev else 0.0
    gaps += gap
    nm = short(r["Kernel_Name"])
    a = agg.setdefault(nm, [0, 0.0])
    a[0] += 1; a[1] += (e - s) / 1e3
    if "--full" in sys.argv:
        grid = "x".join(r.get(k, "?") for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z") if r.get(k, "1") != "1") or "1"
        print("%9.1f us  dur %8.1f  gap %7.1f  %-36s threads %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, nm, grid))
    prev = max(prev or 0, e)
span = (prev - t0) / 1e3
print("timed call: %.1f us from first to last kernel, %d kernels, idle gaps %.1f us" % (span, len(seg), gaps))
for nm, (cnt, dur) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-44s x%-3d %9.1f us  %5.1f %%" % (nm[:44], cnt, dur, 100 * dur / span))
