# Hardware counters of chosen kernels of the benchmark command, one rocprofv3 --pmc pass per counter set (kernel-trace only, as the guide prescribes),
# summarised per kernel as min / quartiles / max over its dispatches (the median of accumulate_kernel<float, 1, true> is a certified, cached launch).
# usage: bash scripts/pmc_kernels.sh <out-name> '<kernel-regex>' "<set 1>" "<set 2>" ...      (sets: space-separated counter names)
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-pmc}; mkdir -p $O; RX=$2; shift 2
i=0
for set in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $O/p$i -o p --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-extra-legs > $O/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - "$O" "$RX" <<'PY'
import csv, glob, sys, collections, re
rx = re.compile(sys.argv[2])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p*/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if rx.search(k):
            short = k.split("::")[1].split("(")[0] if "::" in k else k.split("(")[0]
            agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
def q(v, p):
    v = sorted(v); return v[min(len(v) - 1, int(p * len(v)))]
with open(sys.argv[1] + "/summary.txt", "w") as out:
    for k in sorted(agg):
        out.write(k + "\n")
        for c in sorted(agg[k]):
            v = agg[k][c]
            out.write("   %-28s n=%3d min=%14.1f p25=%14.1f median=%14.1f p75=%14.1f max=%14.1f\n" % (c, len(v), min(v), q(v, .25), q(v, .5), q(v, .75), max(v)))
print(open(sys.argv[1] + "/summary.txt").read())
PY
rm -rf $O/p[0-9]*/
