"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, --kernel-trace only) into the
per-launch HBM traffic table bench.py quotes.  Units and corrections per MI355X_MICROARCH.md (HBM):
FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly half of a wide coalesced
streaming read, so it is doubled; WRITE_SIZE is exact for 16-B streaming stores and float atomics.

    python scripts/pmc_summary.py <fetch_dir> <write_dir> <out.json> [B n [commit]]
"""
import collections, csv, glob, json, sys


def load(d, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return agg


import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_sources_sha16      # (what bench.py compares: a traffic figure quoted from this profile says whether the kernels have changed since)

fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {"units": "bytes per launch (mean over launches)", "fetch_correction": "FETCH_SIZE KiB x 1024 x 2 (gfx950 wide-read under-count)",
       "workload": {"B": int(sys.argv[4]) if len(sys.argv) > 4 else 256, "n": int(sys.argv[5]) if len(sys.argv) > 5 else 16384},
       "commit": sys.argv[6] if len(sys.argv) > 6 else "unknown", "kernel_sources_sha16": kernel_sources_sha16(), "kernels": {}}
for name in sorted(set(fetch) | set(write)):
    if "anonymous namespace" not in name:
        continue
    short = name.split("::")[1].split("(")[0] if "::" in name else name
    f = sum(fetch.get(name, [0])) / max(1, len(fetch.get(name, [])))
    w = sum(write.get(name, [0])) / max(1, len(write.get(name, [])))
    out["kernels"][short] = {"launches": len(fetch.get(name, [])), "FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w,
                             "read_bytes_corrected": f * 1024 * 2, "write_bytes": w * 1024, "hbm_bytes": f * 1024 * 2 + w * 1024}
    # launches in which every cloud was at work (the backward's truncated reverse sweep leaves most accumulate_bwd launches (next to) empty:
    # their blocks read one flag and leave): those whose read traffic is at least half the largest launch's
    fl, wl = fetch.get(name, []), write.get(name, [])
    # (the two passes are separate runs of the command, whose warm-up repeats until its calls agree: they need not have the same number of launches,
    #  so each pass picks its own full launches)
    #  so a kernel with (next to) empty launches -- by its reads -- takes the full ones of each pass by that pass's own figures)
    if fl and wl:
        kf = [v for v in fl if v >= 0.5 * max(fl)]
        if len(kf) == len(fl):
            kw = wl                                         # every launch did its work: the mean over all of them
        elif len(fl) == len(wl):
            kw = [wl[i] for i, v in enumerate(fl) if v >= 0.5 * max(fl)]
        else:
            kw = [v for v in wl if v >= 0.5 * max(wl)] if max(wl) > 0 else wl
        ff, wf = sum(kf) / len(kf), sum(kw) / len(kw)
        out["kernels"][short].update({"full_launches": len(kf), "hbm_bytes_full_launches": ff * 1024 * 2 + wf * 1024})
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
