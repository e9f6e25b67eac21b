"""Parse hipcc -Rpass-analysis=kernel-resource-usage remarks from stdin into one line per kernel (demangled): VGPRs, AGPRs, SGPRs, scratch bytes per lane,
occupancy (waves per SIMD), LDS bytes.  argv[1]: regex filter on the demangled name."""
import re
import subprocess
import sys

flt = re.compile(sys.argv[1] if len(sys.argv) > 1 else ".")
cur, rows = None, []
for line in sys.stdin:
    m = re.search(r"remark: .*Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    for key, pat in (("vgpr", r"VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("sgpr", r"SGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                     ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
        m = re.search(r"remark: .*" + pat, line)
        if m and cur is not None and key not in cur:
            cur[key] = int(m.group(1))
names = [r["name"] for r in rows]
dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines() if names else []
print("%-6s %-5s %-5s %-8s %-4s %-7s %s" % ("VGPR", "AGPR", "SGPR", "scratch", "occ", "LDS", "kernel"))
for r, d in zip(rows, dem):
    d = re.sub(r"^void \(anonymous namespace\)::", "", d)
    d = re.sub(r"\(.*$", "", d)
    if flt.search(d):
        print("%-6d %-5d %-5d %-8d %-4d %-7d %s" % (r.get("vgpr", -1), r.get("agpr", -1), r.get("sgpr", -1), r.get("scratch", -1), r.get("occ", -1), r.get("lds", -1), d))
