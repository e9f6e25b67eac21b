# kernel timeline of the last call of scripts/call_profile.py.  usage: bash scripts/call_timeline.sh <out-name> [scene|random] [K] [calls]
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-ctl}; mkdir -p $O
rocprofv3 --kernel-trace -d $O/t -o t --output-format csv -- python3 $R/scripts/call_profile.py ${2:-random} ${3:-10} ${4:-5} > $O/log.txt 2>&1
python3 - $O/t/t_kernel_trace.csv > $O/timeline.txt <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
# the last call: from the last search_frame_kernel (the first kernel of a sweep-path call) to the last pose_grad_out_kernel (the backward's last)
start = max([i for i, r in enumerate(rows) if "search_frame_kernel" in r["Kernel_Name"]] or [0])
ends = [i for i, r in enumerate(rows) if "pose_grad_out_kernel" in r["Kernel_Name"] and i > start]
stop = (max(ends) + 1) if ends else len(rows)
t0 = int(rows[start]["Start_Timestamp"]); prev = None
for r in rows[start:stop]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
    print("%9.1f us  dur %8.1f  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, 0.0 if prev is None else (s - prev) / 1e3, name))
    prev = e
print("call: %.1f us" % ((prev - t0) / 1e3))
PY
rm -rf $O/t; tail -3 $O/log.txt
