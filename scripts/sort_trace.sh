# per-launch durations of the key-sort kernels in scripts/sort_check.py (in launch order).  usage: bash scripts/sort_trace.sh
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d /tmp/st -o s --output-format csv -- python3 $R/scripts/sort_check.py > /tmp/st.log 2>&1
python3 - <<'PY'
import csv
rows = sorted(csv.DictReader(open('/tmp/st/s_kernel_trace.csv')), key=lambda r: int(r["Start_Timestamp"]))
out = []
for r in rows:
    n = r["Kernel_Name"]
    if "sort_big" in n or "sort_keys" in n or "sweep_rows" in n or "sweep_buckets" in n or "search_frame" in n:
        out.append("%s %.1f" % (n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
# the last SweepIndex of each configuration: print the last 10 launches before each change of grid size is hard to see; just print a window per config
print("\n".join(out[-120:]))
PY
