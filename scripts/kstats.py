"""Per-kernel summary of a rocprofv3 --kernel-trace --output-format csv run: calls, mean / total duration.
    python scripts/kstats.py <dir> [top]"""
import collections, csv, glob, re, sys
rows = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        own = re.search(r"\(anonymous namespace\)::(\w+)(<[^(]*>)?\(", name)
        name = (own.group(1) + (own.group(2) or "")) if own else name.split("(")[0][-70:]
        rows[name].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = sum(sum(v) for v in rows.values())
print("%-58s %7s %10s %10s %6s" % ("kernel", "calls", "mean us", "total us", "%"))
for name, v in sorted(rows.items(), key=lambda kv: -sum(kv[1]))[:int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    print("%-58s %7d %10.2f %10.1f %6.1f" % (name[:58], len(v), sum(v) / len(v) / 1e3, sum(v) / 1e3, 100.0 * sum(v) / tot))
