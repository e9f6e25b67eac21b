# SQ / matrix-pipe counters of the matrix-core search (separate --pmc passes, kernel-trace only).  usage: bash scripts/pmc_f16.sh <outdir-name> [B] [NPTS] [script] [POSE_ITERS]
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-pmc_f16}; mkdir -p $O
export B=${2:-256} NPTS=${3:-16384} REPS=3 FORMS=mfma POSE_ITERS=${5:-0}
S=${4:-f16_knn_bench.py}
python3 $R/scripts/$S > $O/timing.txt 2>&1
rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 $R/scripts/$S > $O/kt.log 2>&1 || echo "kernel trace failed"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_IFETCH"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $O/p$i -o p --output-format csv -- python3 $R/scripts/$S > $O/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - "$O" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p*/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "knn_f16" in k:
            short = "knn_f16_sweep_kernel" if "sweep" in k else ("knn_f16_kernel" if "knn_f16_kernel" in k else k.split("(")[0][-40:])
            agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(sys.argv[1] + "/summary.txt", "w") as out:
    out.write(open(sys.argv[1] + "/timing.txt").read())
    for f in glob.glob(sys.argv[1] + "/kt/**/*_kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "knn" in r["Name"]:
                out.write("kernel-trace stats: %s calls %s avg %.1f us\n" % (r["Name"].split("(")[0][-60:], r["Calls"], float(r["AverageNs"]) / 1e3))
    for k in sorted(agg):
        out.write(k + "\n")
        for c in sorted(agg[k]):
            v = agg[k][c]
            out.write("   %-32s n=%3d mean=%16.1f min=%16.1f max=%16.1f\n" % (c, len(v), sum(v) / len(v), min(v), max(v)))
print(open(sys.argv[1] + "/summary.txt").read())
PY
