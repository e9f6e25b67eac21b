# average duration of the kernels whose name contains $1, per build: bash scripts/kernel_time.sh <name> <libA.so> <libB.so> ...
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; K=$1; shift
for L in "$@"; do
  O=$R/gpurun_out/kt_$(basename $L .so); rm -rf $O; mkdir -p $O
  DICP_HIP_LIB=$R/$L rocprofv3 --kernel-trace --stats -d $O -o s --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $O/log.txt 2>&1
  python3 - "$O/s_kernel_stats.csv" "$K" "$L" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Name"]:
        print("%-28s %-60s calls %4s  avg %9.1f us" % (sys.argv[3], r["Name"].split("::")[1].split("(")[0][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
